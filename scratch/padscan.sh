#!/bin/bash
# one fresh process per field pad (same allocation sequence -> comparable placement); prints kernel ms of k_mf_fused_tv
for cols in 512 4096; do
  if [ $cols = 512 ]; then pads="0 32 64 96 128 160 192 224 256 288 320 384 448 512 640 768 1024 1536 2048"; else pads="0 64 128 192 256 320 384 512 768 1024 2048 4096 6144 8192"; fi
  for p in $pads; do
    echo -n "cols $cols "; FIELD_PADS=$p timeout -k 10 300 python3 scratch/bimodal.py $cols 2>/dev/null | grep "^handle" || exit 1
  done
done
