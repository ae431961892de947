#!/bin/bash
# EXPERIMENT (library scratch/exp/libcp_arena.so, not the shipped one): layouts of the six big buffers inside one allocation
OUT=$PWD/gpurun_out/arenascan2; mkdir -p $OUT
export CP_ENGINE_LIB=$PWD/scratch/exp/libcp_arena.so
for rep in 1 2; do
  for a in 012345:128:0 012345:256:0 012345:384:0 012345:512:0 012345:640:0 012345:768:0 012345:896:0 012345:1152:0 012345:1280:0 012345:1536:0 012345:1792:0 012345:2560:0; do
    if [ $a = none ]; then unset CP_TUNE_ARENA; else export CP_TUNE_ARENA=$a; fi
    timeout -k 10 300 python3 bench.py --steps 20 --warmup 3 --repeats 2 --no-cpu-baseline "$@" > $OUT/b.json 2> $OUT/b.err || { echo "arena $a FAILED"; tail -5 $OUT/b.err; exit 1; }
    python3 -c "import json; d=json.load(open('$OUT/b.json')); print('rep $rep arena $a kernel', round(d['config']['kernel_ms']['dual'],4))"
  done
done
