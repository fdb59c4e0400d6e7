#!/bin/bash
# Per-kernel time of the bench step (rocprofv3 --kernel-trace --stats), top rows printed.
# Usage on the GPU box: bash tools/kstats.sh <tag> [bench args]
set -e
TAG=${1:-k}; shift || true
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/kstats_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- \
    python3 "$R/bench.py" --no-cpu-baseline --no-kernel-timing ${KSTEPS:---steps 10 --warmup 3} "$@" > "$OUT/bench.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:16]:
    print("%-60s calls %5s  avg %9.2f us  total %9.3f ms  %s%%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
grep -o '"ms_per_step": [0-9.]*' "$OUT/bench.log"
