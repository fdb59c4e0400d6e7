#!/bin/bash
# HBM-side bytes of the log-mel kernel: one rocprofv3 run per counter (FETCH_SIZE, WRITE_SIZE), median per launch.
# Usage on the GPU box: bash tools/traffic_logmel.sh <tag>   -> gpurun_out/<tag>_pmc_logmel_<counter>/ and a line on stdout
set -e
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
cd /tmp; export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf "$OUT/${TAG}_pmc_logmel_$C"
    timeout -s KILL 120 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/${TAG}_pmc_logmel_$C" -- \
        python3 "$R/tools/logmel_one.py" 5 > "$OUT/${TAG}_pmc_logmel_$C.log" 2>&1
done
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, sys
out, tag = sys.argv[1], sys.argv[2]
vals = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("%s/%s_pmc_logmel_%s/**/*counter_collection.csv" % (out, tag, c), recursive=True)[0]
    v = sorted(float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "logmel_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c)
    vals[c] = v[len(v) // 2]
fetch, write = vals["FETCH_SIZE"] * 1024 * 2, vals["WRITE_SIZE"] * 1024     # KiB; FETCH_SIZE counts half of a wide streaming read on gfx950
algo = 32 * (480000 * 4 + 2998 * 80 * 4)
print("logmel_kernel 32x480000: FETCH_SIZE %.1f KiB raw -> %.2f MB, WRITE_SIZE %.1f KiB -> %.2f MB, traffic %.2f MB = %.3f x the algorithmic %.2f MB"
      % (vals["FETCH_SIZE"], fetch / 1e6, vals["WRITE_SIZE"], write / 1e6, (fetch + write) / 1e6, (fetch + write) / algo, algo / 1e6))
PY
