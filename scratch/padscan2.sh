#!/bin/bash
# field pad candidates x shapes, one bench.py process each
OUT=$PWD/gpurun_out/padscan2; mkdir -p $OUT
run() {  # label, args...
  lab=$1; shift
  for p in 0 320 384 640 768; do
    CP_TUNE_FIELD_PAD=$p timeout -k 10 300 python3 bench.py --steps 20 --warmup 3 --repeats 3 --no-cpu-baseline "$@" > $OUT/b.json 2> $OUT/b.err || { echo "$lab pad $p FAILED"; tail -3 $OUT/b.err; continue; }
    python3 -c "import json; d=json.load(open('$OUT/b.json')); print('$lab pad $p ms_per_step', round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['config']['kernel_ms'].items()})"
  done
}
run tv4096
run tv4096x512 --cols 512
run tv4096x1024 --cols 1024
run tv4096x2048 --cols 2048
run tv2048 --size 2048
run tv1024k10 --size 1024 --k 10 --steps 100
run tv4096k8 --k 8
run tvfp32 --fp32
run curv4096 --reg curvature
run curv4096x512 --reg curvature --cols 512
run batch --workload batch
