#!/bin/bash
# usage: scratch/pmc_other.sh <tag> <bench_other workload>: FETCH_SIZE / WRITE_SIZE passes of scratch/bench_other.py, medians per kernel
TAG=$1; W=$2
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmco_$TAG
mkdir -p $OUT
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 scratch/bench_other.py $W > $OUT/fetch.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_write -- python3 scratch/bench_other.py $W > $OUT/write.log 2>&1 || exit 1
python3 - $OUT <<'PY'
import sys, glob, csv, collections
out = sys.argv[1]
for sub in ("pmc_fetch", "pmc_write"):
    for f in glob.glob(out + f"/{sub}/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, d in agg.items():
            if not any(t in k for t in ("k_var_lazy", "k_curv_y2", "k_mf_fused", "k_pw_march")):
                continue
            print(sub, k, " ".join(f"{c}={sorted(v)[len(v)//2]:.4g}" for c, v in d.items()))
PY
