#!/bin/bash
# 4-wave kernel (default below 3585 rows) against the 8-wave hand-over kernel forced, over problem sizes
OUT=$PWD/gpurun_out/wavescan; mkdir -p $OUT
for cfg in "2048 16" "1536 16" "1024 16" "1024 10" "768 16" "512 16" "512 10" "384 16" "256 16" "256 10" "1024 8" "1024 6" "2048 8"; do
  set -- $cfg
  for w in - 8; do
    if [ $w = - ]; then unset CP_TUNE_FUSED_WAVES; else export CP_TUNE_FUSED_WAVES=$w; fi
    timeout -k 10 300 python3 bench.py --size $1 --k $2 --steps 40 --warmup 5 --repeats 3 --no-cpu-baseline > $OUT/b.json 2> $OUT/b.err || { echo "FAILED $cfg $w"; tail -5 $OUT/b.err; exit 1; }
    python3 -c "import json; d=json.load(open('$OUT/b.json')); print('size $1 k $2 waves $w ms/step', round(d['ms_per_step'],4), 'kernel', round(d['config']['kernel_ms']['dual'],4), d['roofline']['kernel'][-22:])"
  done
done
