#!/bin/bash
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/exp3b; mkdir -p $OUT
run() {
  tag=$1; export CP_ENGINE_LIB=$2; shift 2
  timeout -k 10 240 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline "$@" > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err || return 1
  python3 -c "import json; d=json.load(open('$OUT/bench_$tag.json')); print('$tag', round(d['ms_per_step'],3), d['config']['kernel_ms'], d['handover_fallbacks'])"
}
run bnd $PWD/ma_amd/libcpengine.so || exit 1
run noshf $PWD/scratch/exp/libcp_noshf.so || exit 1
run bnd_slab $PWD/ma_amd/libcpengine.so --cols 512 || exit 1
run noshf_slab $PWD/scratch/exp/libcp_noshf.so --cols 512 || exit 1
run bnd2 $PWD/ma_amd/libcpengine.so || exit 1
run noshf2 $PWD/scratch/exp/libcp_noshf.so || exit 1
