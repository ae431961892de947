#!/bin/bash
# usage: scratch/prof_fetch.sh <tag>  -- FETCH_SIZE / WRITE_SIZE+TCC passes only (env knobs are inherited)
set -e
TAG=$1
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_fetch.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_write.log 2>&1
python3 scratch/prof_summary.py $OUT > $OUT/summary.txt 2>&1 || true
cat $OUT/summary.txt
