#!/bin/bash
# usage: scratch/ab.sh "<ENV=VAL ...>" "<ENV=VAL ...>" ...   -- bench.py once per environment setting, twice round robin (box drift)
mkdir -p gpurun_out
for rep in 1 2; do
  i=0
  for cfg in "$@"; do
    i=$((i+1))
    env $cfg timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline $AB_ARGS > gpurun_out/ab_${i}_$rep.json 2> gpurun_out/ab_${i}_$rep.err || { echo "FAILED: $cfg"; tail -3 gpurun_out/ab_${i}_$rep.err; exit 1; }
    python - "$cfg" gpurun_out/ab_${i}_$rep.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
print(f"{sys.argv[1]:40s} it/s {d['value']:.2f}  ms/step {d['ms_per_step']:.3f}  kernels {d['config']['kernel_ms']}")
PY
  done
done
