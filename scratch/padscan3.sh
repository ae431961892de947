#!/bin/bash
OUT=$PWD/gpurun_out/padscan3; mkdir -p $OUT
for rep in 1 2 3 4; do
  for p in 0 320 384 640 768 1088; do
    CP_TUNE_FIELD_PAD=$p timeout -k 10 300 python3 bench.py --steps 20 --warmup 3 --repeats 3 --no-cpu-baseline > $OUT/b.json 2> $OUT/b.err || { echo "pad $p FAILED"; continue; }
    python3 -c "import json; d=json.load(open('$OUT/b.json')); print('tv4096 rep $rep pad $p kernel', round(d['config']['kernel_ms']['dual'],4))"
  done
done
