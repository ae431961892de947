#!/bin/bash
# do the virtual addresses of the big buffers differ between the fast and the slow processes?
OUT=$PWD/gpurun_out/addrscan; mkdir -p $OUT
export CP_DEBUG_ADDR=1
for rep in 1 2 3 4 5 6 7 8; do
  timeout -k 10 300 python3 bench.py --steps 20 --warmup 3 --repeats 2 --no-cpu-baseline "$@" > $OUT/b.json 2> $OUT/b.err || { echo "FAILED"; tail -5 $OUT/b.err; exit 1; }
  python3 -c "import json; d=json.load(open('$OUT/b.json')); print('rep $rep kernel', round(d['config']['kernel_ms']['dual'],4))"
  grep "cpengine addr" $OUT/b.err | tail -1
done
