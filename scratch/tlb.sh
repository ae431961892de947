#!/bin/bash
# address-translation counters of the field kernels (is the process-to-process spread a TLB effect?)
set -e
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/tlb; mkdir -p $OUT
for rep in 1 2 3 4 5 6; do
timeout -s KILL 300 rocprofv3 --pmc ${CTRS:-TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum} --output-format csv -d $OUT/pmc_tlb$rep -- python3 bench.py --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline "$@" > $OUT/bench_tlb$rep.json 2> $OUT/bench_tlb$rep.log
grep "^{" $OUT/bench_tlb$rep.json | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('kernel ms', d['config']['kernel_ms'])"
python3 - $OUT/pmc_tlb$rep <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    if "fused" in k or "y2" in k:
        print(k, {c: sorted(v)[len(v) // 2] for c, v in d.items()}, "n", len(next(iter(d.values()))))
PY
done
