#!/bin/bash
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/full_$1; mkdir -p $OUT
timeout -k 10 1500 python3 -u -m pytest tests -m gpu -x -q --timeout 600 --timeout-method thread --durations=8 > $OUT/pytest.txt 2>&1 || { tail -40 $OUT/pytest.txt; exit 1; }
tail -14 $OUT/pytest.txt
for a in "" "--cols 512" "--reg curvature" "--reg curvature --cols 512"; do
timeout -k 10 600 python3 bench.py --no-cpu-baseline $a > $OUT/bench.json 2> $OUT/bench.err || { tail -20 $OUT/bench.err; exit 1; }
python3 -c "import json; d=json.load(open('$OUT/bench.json')); print('$a', round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['config']['kernel_ms'].items()}, round(d['roofline']['frac'],3))"
done
