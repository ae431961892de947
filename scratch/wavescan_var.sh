#!/bin/bash
# variance-TV: 4-wave kernel against the 8-wave hand-over kernel forced, over problem sizes
for sz in 2048 1536 1024 768 512; do
  for w in - 8; do
    if [ $w = - ]; then unset CP_TUNE_VAR_WAVES; else export CP_TUNE_VAR_WAVES=$w; fi
    echo -n "size $sz waves $w: "; BENCH_SIZE=$sz timeout -k 10 300 python3 scratch/bench_other.py var | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_it'],4), round(d['kernel_ms']['dual'],4))" || exit 1
  done
done
