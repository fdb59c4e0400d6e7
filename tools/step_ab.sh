#!/bin/bash
# Step-level A/B on one box: bench.py (steps only) alternating between library variants, three rounds.
# Usage (on the GPU box): bash tools/step_ab.sh lib1.so lib2.so ...
R=$(cd "$(dirname "$0")/.." && pwd)
for round in 1 2 3; do
    for lib in "$@"; do
        ms=$(OASR_HIP_LIBRARY="$R/$lib" timeout -k 10 120 python3 "$R/bench.py" --no-cpu-baseline --no-kernel-timing --steps 30 --warmup 5 2>/dev/null |
             python3 -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])") || exit 1
        echo "round $round $(basename $lib) $ms"
    done
done
