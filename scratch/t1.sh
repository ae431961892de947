#!/bin/bash
OUT=$PWD/gpurun_out/t1; mkdir -p $OUT
timeout -k 10 900 python3 -u -m pytest -x -q --timeout 300 --timeout-method thread "$@" > $OUT/pytest.txt 2>&1; rc=$?
tail -${TAILN:-30} $OUT/pytest.txt | cut -c1-250
exit $rc
