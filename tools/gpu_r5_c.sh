#!/bin/bash
# round 5, call c: new GPU tests (ADVICE fixes, library-side decomposition), then A/B of the split-loop / prologue-priority variants
out=gpurun_out/r5c; mkdir -p $out; rm -f $out/*
timeout 900 python -m pytest tests/test_multi.py tests/test_engine_order.py -x -q -m gpu -k "multi or exact_mode_on or renewed_order_leaves or clump_major" -s > $out/new_tests.log 2>&1; tail -15 $out/new_tests.log
ROUNDS=2 timeout 1200 bash tools/gpu_ab.sh > $out/ab.log 2>&1; cat $out/ab.log
