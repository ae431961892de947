#!/bin/bash
# is the fast / slow mode of a process a clock state? time the single-workgroup eigensolve kernel (pure latency / ALU) beside the field kernel
OUT=$PWD/gpurun_out/modescan; mkdir -p $OUT
export CP_TUNE_TIME_ALL=1
for rep in 1 2 3 4 5 6 7 8; do
  timeout -k 10 300 python3 bench.py --steps 20 --warmup 3 --repeats 3 --no-cpu-baseline "$@" > $OUT/b.json 2> $OUT/b.err || { echo "FAILED"; tail -5 $OUT/b.err; exit 1; }
  python3 -c "import json; d=json.load(open('$OUT/b.json')); print('rep $rep', {k: round(x,5) for k,x in d['config']['kernel_ms'].items()}, 'copy', d['roofline']['measured_copy_GBps'])"
done
