#!/bin/bash
OUT=$PWD/gpurun_out/parts; mkdir -p $OUT
for rep in 1 2; do
for cols in 4096 2048 1024 512; do
for v in 0 2 4 8 16; do
  CP_TUNE_PARTS8=$v timeout -k 10 300 python3 bench.py --cols $cols --steps 20 --warmup 3 --repeats 3 --no-cpu-baseline > $OUT/b.json 2> $OUT/b.err || { echo "$v FAILED"; continue; }
  python3 -c "import json; d=json.load(open('$OUT/b.json')); print('cols $cols parts $v rep $rep kernel ms', round(d['config']['kernel_ms']['dual'],4))"
done; done; done
