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
# BASELINE configs[1] (one 30 s utterance) and configs[4] (7B geometry, fp8 projections, batch 8): kernel statistics and bench lines
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_stats_b1" -- \
    python3 "$R/bench.py" --batch 1 --steps 30 --warmup 5 --no-cpu-baseline > "$OUT/${TAG}_stats_b1.log" 2>&1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_stats_g7b_fp8" -- \
    python3 "$R/bench.py" --model G7B --batch 8 --fp8 --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/${TAG}_stats_g7b_fp8.log" 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
    timeout -s KILL 120 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/${TAG}_pmc_gemm8_$C" -- \
        python3 "$R/tools/gemm8_one.py" > "$OUT/${TAG}_pmc_gemm8_$C.log" 2>&1
done
tail -1 "$OUT/${TAG}_stats_graph.log"
