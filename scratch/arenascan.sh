#!/bin/bash
# placement experiment: X, P, g, b of a handle from one allocation (CP_TUNE_ARENA=<KB between buffers>) against separate allocations
OUT=$PWD/gpurun_out/arenascan; mkdir -p $OUT
for rep in 1 2 3 4; do
  for a in none 0 3 1027 65; do
    if [ $a = none ]; then unset CP_TUNE_ARENA; else export CP_TUNE_ARENA=$a; fi
    timeout -k 10 300 python3 bench.py --steps 20 --warmup 3 --repeats 3 --no-cpu-baseline "$@" > $OUT/b.json 2> $OUT/b.err || { echo "arena $a FAILED"; tail -5 $OUT/b.err; exit 1; }
    python3 -c "import json; d=json.load(open('$OUT/b.json')); print('rep $rep arena $a kernel', round(d['config']['kernel_ms']['dual'],4))"
  done
done
