#!/bin/bash
# usage: scratch/libscan_other.sh "<lib tags>" <bench_other workloads>: A/B of libraries on the secondary workloads (tag "base" = shipped)
TAGS=$1; shift
for rep in 1 2 3; do
  for t in $TAGS; do
    if [ $t = base ]; then export CP_ENGINE_LIB=$PWD/ma_amd/libcpengine.so; else export CP_ENGINE_LIB=$PWD/scratch/exp/libcp_$t.so; fi
    echo -n "rep $rep $t "; timeout -k 10 300 python3 scratch/bench_other.py "$@" | python3 -c "import sys,json; [print(d['workload'], round(d['ms_per_it'],4), {k: round(v,4) for k,v in d['kernel_ms'].items() if v}) for d in map(json.loads, sys.stdin)]" || exit 1
  done
done
