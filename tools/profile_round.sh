#!/bin/bash
# Collects the profiler evidence kept under profiles/: kernel-trace statistics of the bench command and
# the HBM-side byte counters of the two roofline kernels (one counter per profiler run).
# Usage on the GPU box: bash tools/profile_round.sh <tag>   -> gpurun_out/<tag>_*
set -e
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
mkdir -p "$OUT"
cd /tmp
export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_stats_graph" -- \
    python3 "$R/bench.py" --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/${TAG}_stats_graph.log" 2>&1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_stats_eager" -- \
    python3 "$R/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-graph > "$OUT/${TAG}_stats_eager.log" 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
    timeout -s KILL 120 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/${TAG}_pmc_gemm_$C" -- \
        python3 "$R/tools/gemm_one.py" 48128 4096 1024 1 5 > "$OUT/${TAG}_pmc_gemm_$C.log" 2>&1
    timeout -s KILL 120 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/${TAG}_pmc_logmel_$C" -- \
        python3 "$R/tools/logmel_one.py" 5 > "$OUT/${TAG}_pmc_logmel_$C.log" 2>&1
done
tail -1 "$OUT/${TAG}_stats_graph.log"
