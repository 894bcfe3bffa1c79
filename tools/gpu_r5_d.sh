#!/bin/bash
# round 5, call d: the host-shell slab tests + the fixed engine-order test, then A/B of the tile-kernel variants
out=gpurun_out/r5d; mkdir -p $out; rm -f $out/*
timeout 900 python -m pytest tests/test_host_shell.py tests/test_engine_order.py -x -q -m gpu -k "slabs or renewed_order_leaves" -s > $out/new_tests.log 2>&1; tail -8 $out/new_tests.log
ROUNDS=2 timeout 1500 bash tools/gpu_ab.sh > $out/ab.log 2>&1; cat $out/ab.log
