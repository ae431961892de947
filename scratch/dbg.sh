#!/bin/bash
OUT=$PWD/gpurun_out/dbg; mkdir -p $OUT
CP_ENGINE_LIB=$PWD/scratch/exp/libcp_noshf.so timeout -k 10 300 python3 -u -m pytest -q --timeout 120 --timeout-method thread tests/test_gpu_mfnn.py -k "solve_matches_oracle and curvature" > $OUT/noshf.txt 2>&1; tail -3 $OUT/noshf.txt
timeout -k 10 300 python3 -u -m pytest -q --timeout 120 --timeout-method thread tests/test_gpu_mfnn.py -k "solve_matches_oracle and curvature" > $OUT/shf.txt 2>&1; tail -3 $OUT/shf.txt
