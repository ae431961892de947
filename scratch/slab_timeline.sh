#!/bin/bash
# usage: scratch/slab_timeline.sh <tag>: kernel timeline of the multi-GPU code path (one RCCL rank) on the slab of one of 8 ranks
export TMPDIR=/tmp CP_FORCE_SLAB_PATH=1 MASTER_ADDR=127.0.0.1
for reg in tv curvature; do
  OUT=$PWD/gpurun_out/slabtl_$1_$reg
  mkdir -p $OUT
  timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 bench.py --cols 512 --reg $reg --steps 40 --warmup 5 --repeats 1 --no-cpu-baseline > $OUT/bench.log 2>&1 || { echo "failed $reg"; tail -5 $OUT/bench.log; exit 1; }
  f=$(ls $OUT/*/*kernel_trace.csv | head -1)
  echo "== $reg"; grep '^{' $OUT/bench.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms/step', d['ms_per_step'], d['config']['kernel_ms'])"
  python3 scratch/trace_gaps.py $f
  rm -f $f
done
