#!/bin/bash
# usage: [ENV=...] scratch/pmc_quick.sh <tag> [bench args]: FETCH_SIZE and WRITE_SIZE/TCC hit-miss passes of bench.py, medians per kernel
TAG=$1; shift
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmcq_$TAG
mkdir -p $OUT
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $OUT/bench_fetch.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $OUT/bench_write.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_sq -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $OUT/bench_sq.log 2>&1 || exit 1
python3 scratch/prof_summary.py $OUT | grep -A3 "== pmc" | grep -v "k_mf_efused\|^--"
