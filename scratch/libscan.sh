#!/bin/bash
# usage: scratch/libscan.sh "<lib tags>" [bench args]: bench.py with scratch/exp/libcp_<tag>.so (tag "base" = the shipped library), kernel times
OUT=$PWD/gpurun_out/libscan; mkdir -p $OUT
TAGS=$1; shift
for rep in 1 2; do
  for t in $TAGS; do
    if [ $t = base ]; then export CP_ENGINE_LIB=$PWD/ma_amd/libcpengine.so; else export CP_ENGINE_LIB=$PWD/scratch/exp/libcp_$t.so; fi
    timeout -k 10 300 python3 bench.py --steps 20 --warmup 3 --repeats 3 --no-cpu-baseline "$@" > $OUT/b.json 2> $OUT/b.err || { echo "$t FAILED"; tail -5 $OUT/b.err; exit 1; }
    python3 -c "import json; d=json.load(open('$OUT/b.json')); print('rep $rep $t ms/step', round(d['ms_per_step'],4), {k: round(v,4) for k,v in d['config']['kernel_ms'].items()})"
  done
done
