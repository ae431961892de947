#!/bin/bash
# round 3, experiment 3: the boundary array (SHF=1) against the element-shifted loads (SHF=0) on one box: parity tests, then time + FETCH
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/exp3; mkdir -p $OUT
timeout -k 10 900 python3 -u -m pytest -x -q --timeout 300 --timeout-method thread tests/test_gpu_mfnn.py tests/test_gpu_fp32.py tests/test_gpu_slab.py tests/test_gpu_golden.py tests/test_gpu_drivers.py tests/test_gpu_fuzz.py > $OUT/pytest.txt 2>&1 || { tail -40 $OUT/pytest.txt; exit 1; }
tail -3 $OUT/pytest.txt
run() {
  tag=$1; export CP_ENGINE_LIB=$2; shift 2
  timeout -k 10 240 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline "$@" > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err || return 1
  python3 -c "import json; d=json.load(open('$OUT/bench_$tag.json')); print('$tag', round(d['ms_per_step'],3), d['config']['kernel_ms'], d['handover_fallbacks'])"
  timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/v$tag/pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline "$@" > $OUT/pmc_$tag.log 2>&1 || return 1
  timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/v$tag/pmc_write -- python3 bench.py --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline "$@" > $OUT/pmcw_$tag.log 2>&1 || return 1
  python3 scratch/prof_summary.py $OUT/v$tag | grep -v efused | tee $OUT/summary_$tag.txt
}
run bnd $PWD/ma_amd/libcpengine.so || exit 1
run noshf $PWD/scratch/exp/libcp_noshf.so || exit 1
run bnd_slab $PWD/ma_amd/libcpengine.so --cols 512 || exit 1
run noshf_slab $PWD/scratch/exp/libcp_noshf.so --cols 512 || exit 1
run bnd_fp32 $PWD/ma_amd/libcpengine.so --fp32 || exit 1
run noshf_fp32 $PWD/scratch/exp/libcp_noshf.so --fp32 || exit 1
run bnd_1024 $PWD/ma_amd/libcpengine.so --size 1024 --k 10 --steps 100 || exit 1
run noshf_1024 $PWD/scratch/exp/libcp_noshf.so --size 1024 --k 10 --steps 100 || exit 1
