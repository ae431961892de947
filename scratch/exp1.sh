#!/bin/bash
# round 3, experiment 1 (GPU box, repo root): copy-kernel shapes + where k_mf_fused_tv's excess reads come from (CP_EXP variants: results wrong, timing / FETCH only)
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/exp1
mkdir -p $OUT
timeout -k 10 120 scratch/exp/bw_copy > $OUT/copy.txt 2>&1 || exit 1
cat $OUT/copy.txt
for e in base 1 2 4 7; do
  if [ $e = base ]; then export CP_ENGINE_LIB=$PWD/ma_amd/libcpengine.so; else export CP_ENGINE_LIB=$PWD/scratch/exp/libcp_exp$e.so; fi
  timeout -k 10 240 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_$e.json 2> $OUT/bench_$e.err || exit 1
  python3 -c "import json,sys; d=json.load(open('$OUT/bench_$e.json')); print('$e', d['ms_per_step'], d['config']['kernel_ms'])"
  timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/v$e/pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_$e.log 2>&1 || exit 1
  python3 scratch/prof_summary.py $OUT/v$e | tee $OUT/summary_$e.txt
done
