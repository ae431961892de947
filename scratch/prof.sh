#!/bin/bash
# usage: scratch/prof.sh <tag> [bench.py arguments ...]   (runs on the GPU box from the repo root): kernel trace + separate PMC passes of bench.py
set -e
TAG=$1
shift
EXTRA="$@"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
python3 -c "import bench; print(bench.source_hash())" > $OUT/source_hash.txt   # the sources these measurements belong to
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 10 --warmup 2 --repeats 1 --no-cpu-baseline $EXTRA > $OUT/bench_trace.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_MOPS_F64 --output-format csv -d $OUT/pmc_sq -- python3 bench.py --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline $EXTRA > $OUT/bench_sq.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline $EXTRA > $OUT/bench_fetch.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline $EXTRA > $OUT/bench_write.log 2>&1
python3 scratch/prof_summary.py $OUT > $OUT/summary.txt 2>&1 || true
cat $OUT/summary.txt
find $OUT -name "*kernel_trace.csv" -size +20M -delete || true
