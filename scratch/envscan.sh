#!/bin/bash
# usage: scratch/envscan.sh VAR "v1 v2 ..." [bench args]: bench.py once per value of an environment knob ("-" = unset), two rounds
OUT=$PWD/gpurun_out/envscan; mkdir -p $OUT
VAR=$1; VALS=$2; shift 2
for rep in 1 2; do
  for v in $VALS; do
    if [ "$v" = "-" ]; then unset $VAR; else export $VAR=$v; fi
    timeout -k 10 300 python3 bench.py --steps 20 --warmup 3 --repeats 3 --no-cpu-baseline "$@" > $OUT/b.json 2> $OUT/b.err || { echo "$VAR=$v FAILED"; tail -5 $OUT/b.err; exit 1; }
    python3 -c "import json; d=json.load(open('$OUT/b.json')); print('rep $rep $VAR=$v ms/step', round(d['ms_per_step'],4), {k: round(x,4) for k,x in d['config']['kernel_ms'].items()})"
  done
done
