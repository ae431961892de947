#!/bin/bash
# EXPERIMENT: field pads beyond 64 KB (elements), buffers in one arena with 1 GB alignment (deterministic placement), scratch/exp/libcp_arena.so
OUT=$PWD/gpurun_out/padscan4; mkdir -p $OUT
export CP_ENGINE_LIB=$PWD/scratch/exp/libcp_arena.so
export CP_TUNE_ARENA=${ARENA:-012345:1024:0}
for rep in 1 2 3; do
  for p in 384 8576 16768 24960 65920 131456 262528 393600 524672; do
    export CP_TUNE_FIELD_PAD=$p
    timeout -k 10 300 python3 bench.py --steps 20 --warmup 3 --repeats 2 --no-cpu-baseline "$@" > $OUT/b.json 2> $OUT/b.err || { echo "pad $p FAILED"; tail -5 $OUT/b.err; exit 1; }
    python3 -c "import json; d=json.load(open('$OUT/b.json')); print('rep $rep pad $p kernel', round(d['config']['kernel_ms']['dual'],4))"
  done
done
