#!/bin/bash
# smoke + fast-mode parity of the current build, then A/B of the default build against every libdeme_v_*.so
out=gpurun_out/r4b; mkdir -p $out; rm -f $out/*
python __graft_entry__.py smoke > $out/smoke.log 2>&1; echo "smoke rc $?" >> $out/smoke.log; tail -3 $out/smoke.log
timeout 900 python -m pytest tests/test_fast_mode.py tests/test_fast_mode_features.py -q -m gpu > $out/fast_tests.log 2>&1; tail -6 $out/fast_tests.log
ROUNDS=${ROUNDS:-2} bash tools/gpu_ab.sh > $out/ab.log 2>&1; cat $out/ab.log
