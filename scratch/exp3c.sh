#!/bin/bash
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/exp3c; mkdir -p $OUT
run() {
  tag=$1; export CP_ENGINE_LIB=$2; shift 2
  timeout -k 10 240 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline "$@" > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err || return 1
  python3 -c "import json; d=json.load(open('$OUT/bench_$tag.json')); print('$tag', round(d['ms_per_step'],3), round(d['config']['kernel_ms']['dual'],4), [round(x,3) for x in d['samples_ms_per_step']])"
}
for rep in 1 2; do
for shape in "--cols 4096" "--cols 512" "--cols 1024" "--size 2048"; do
  t=$(echo $shape | tr -d ' -')
  run early_${t}_$rep $PWD/ma_amd/libcpengine.so $shape || exit 1
  run late_${t}_$rep $PWD/scratch/exp/libcp_late.so $shape || exit 1
  run noshf_${t}_$rep $PWD/scratch/exp/libcp_noshf.so $shape || exit 1
done; done
