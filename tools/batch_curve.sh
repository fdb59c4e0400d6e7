#!/bin/bash
# Step time of the default model over batch sizes (30 s utterances); one line per batch in gpurun_out/curve_<tag>.txt
tag=${1:-cur}; shift
out=gpurun_out/curve_$tag.txt; mkdir -p gpurun_out; : > $out
for b in ${@:-1 2 3 4 5 6 8 10 12 16 20 24 32}; do
  timeout -k 10 120 python bench.py --batch $b --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing > gpurun_out/curve_one.json 2>gpurun_out/curve_one.err || exit 1
  python - $b >> $out <<'PY'
import json,sys
r=json.loads(open("gpurun_out/curve_one.json").read().strip().splitlines()[-1])
print("batch %s: %.3f ms/step  value %.0f" % (sys.argv[1], r["ms_per_step"], r["value"]))
PY
done
cat $out
