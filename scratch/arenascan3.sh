#!/bin/bash
# EXPERIMENT (scratch/exp/libcp_arena.so): single-allocation layouts against separate allocations for the curvature model and other shapes
OUT=$PWD/gpurun_out/arenascan3; mkdir -p $OUT
export CP_ENGINE_LIB=$PWD/scratch/exp/libcp_arena.so
for rep in 1 2 3 4; do
  for a in none 012345:2560:0 012345:1024:0 012345:1536:0; do
    if [ $a = none ]; then unset CP_TUNE_ARENA; else export CP_TUNE_ARENA=$a; fi
    timeout -k 10 300 python3 bench.py --steps 20 --warmup 3 --repeats 2 --no-cpu-baseline "$@" > $OUT/b.json 2> $OUT/b.err || { echo "arena $a FAILED"; tail -5 $OUT/b.err; exit 1; }
    python3 -c "import json; d=json.load(open('$OUT/b.json')); print('rep $rep arena $a ms/step', round(d['ms_per_step'],4), {k: round(v,4) for k,v in d['config']['kernel_ms'].items() if k != 'iterate'})"
  done
done
