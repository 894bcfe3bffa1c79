#!/bin/bash
# round 5, call h: 24-byte tile sums / crossing records: parity tests, then A/B against the 32-byte build
out=gpurun_out/r5h; mkdir -p $out; rm -f $out/*
python __graft_entry__.py smoke > $out/smoke.log 2>&1; tail -2 $out/smoke.log
timeout 1500 python -m pytest tests/test_fast_mode.py tests/test_fast_mode_features.py tests/test_engine_order.py tests/test_force_hook.py tests/test_full_size.py -x -q -m gpu > $out/tests.log 2>&1; tail -4 $out/tests.log
ROUNDS=3 timeout 900 bash tools/gpu_ab.sh > $out/ab.log 2>&1; cat $out/ab.log
