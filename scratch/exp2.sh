#!/bin/bash
# round 3, experiment 2 (GPU box): parity of the lane-shift (i-1, j) kernels and the bounded hand-over, then timing / FETCH of
# k_mf_fused_tv (with / without the shift, with / without the LDS fences) and k_mf_fused_curv with 4 / 8 waves
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/exp2
mkdir -p $OUT
[ -n "$SKIP_PYTEST" ] || timeout -k 10 900 python3 -u -m pytest -x -q --timeout 300 --timeout-method thread tests/test_gpu_mfnn.py tests/test_gpu_fp32.py tests/test_gpu_var.py tests/test_zz_no_alloc.py > $OUT/pytest.txt 2>&1 || { tail -30 $OUT/pytest.txt; exit 1; }
[ -n "$SKIP_PYTEST" ] || tail -3 $OUT/pytest.txt
run() {  # tag, lib, extra env, bench args
  tag=$1; export CP_ENGINE_LIB=$2; shift 2
  timeout -k 10 240 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline "$@" > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err || return 1
  python3 -c "import json; d=json.load(open('$OUT/bench_$tag.json')); print('$tag', round(d['ms_per_step'],3), d['config']['kernel_ms'])"
  timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/v$tag/pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline "$@" > $OUT/pmc_$tag.log 2>&1 || return 1
  python3 scratch/prof_summary.py $OUT/v$tag | grep -v efused | tee $OUT/summary_$tag.txt
}
run tv $PWD/ma_amd/libcpengine.so || exit 1
run tv_noshf $PWD/scratch/exp/libcp_noshf.so || exit 1
run tv_nofence $PWD/scratch/exp/libcp_nofence.so || exit 1
run curv4 $PWD/ma_amd/libcpengine.so --reg curvature || exit 1
export CP_TUNE_CURV_WAVES=8
run curv8 $PWD/ma_amd/libcpengine.so --reg curvature || exit 1
unset CP_TUNE_CURV_WAVES
run curv4_noshf $PWD/scratch/exp/libcp_noshf.so --reg curvature || exit 1
