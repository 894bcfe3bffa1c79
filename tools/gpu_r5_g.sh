#!/bin/bash
# round 5, call g: tests touched since the last full run (slab order, shell slabs, NOZERO default, fused auto in the shell)
out=gpurun_out/r5g; mkdir -p $out; rm -f $out/*
python __graft_entry__.py smoke > $out/smoke.log 2>&1; tail -2 $out/smoke.log
timeout 1500 python -m pytest tests/test_multi.py tests/test_host_shell.py tests/test_fast_mode.py tests/test_fast_mode_features.py tests/test_engine_order.py -x -q -m gpu -s > $out/tests.log 2>&1; tail -6 $out/tests.log
