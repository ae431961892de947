#!/bin/bash
OUT=$PWD/gpurun_out/pitch; mkdir -p $OUT
for rep in 1 2; do
for m in 4096 4160 4224 4352 4032; do
 for cols in 4096 512; do
  timeout -k 10 300 python3 bench.py --size $m --cols $cols --steps 20 --warmup 3 --repeats 3 --no-cpu-baseline > $OUT/b.json 2> $OUT/b.err || { echo "m $m FAILED"; tail -2 $OUT/b.err; continue; }
  python3 -c "import json; d=json.load(open('$OUT/b.json')); k=d['config']['kernel_ms']['dual']; print('m $m cols $cols rep $rep kernel ms', round(k,4), 'ns per pixel-image', round(k*1e6/($m*$cols*16),4), d['roofline']['kernel'])"
 done
done
done
