#!/bin/bash
# placement search of cp_create: probe times, the set that was kept, the kernel time of the run that follows; fresh processes, the
# search switched on and off alternately
OUT=$PWD/gpurun_out/placescan; mkdir -p $OUT
for rep in 1 2 3 4 5 6 7 8; do
 for tries in ${TRIES_LIST:-3 1}; do
  export CP_PLACEMENT_TRIES=$tries
  t0=$(date +%s.%N)
  timeout -k 10 300 python3 bench.py --steps 20 --warmup 3 --repeats 3 --no-cpu-baseline "$@" > $OUT/b.json 2> $OUT/b.err || { echo "FAILED"; tail -5 $OUT/b.err; exit 1; }
  t1=$(date +%s.%N)
  python3 -c "import json; d=json.load(open('$OUT/b.json')); print('rep $rep tries $tries kernel', {k: round(x,4) for k,x in d['config']['kernel_ms'].items() if k != 'iterate'}, 'placement', [round(x,3) for x in d['placement']['probe_ms_per_iteration']], d['placement']['chosen'], 'wall s', round($t1-$t0,1))"
 done
done
