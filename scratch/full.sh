#!/bin/bash
# full GPU test suite, then the default bench line, then the profile passes (scratch/prof.sh <tag>)
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/full_$1; mkdir -p $OUT
timeout -k 10 1500 python3 -u -m pytest tests -m gpu -x -q --timeout 600 --timeout-method thread --durations=15 > $OUT/pytest.txt 2>&1 || { tail -40 $OUT/pytest.txt; exit 1; }
tail -22 $OUT/pytest.txt
timeout -k 10 600 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err || { tail -20 $OUT/bench.err; exit 1; }
cat $OUT/bench.json
[ -n "$NO_PROF" ] || bash scratch/prof.sh $1
