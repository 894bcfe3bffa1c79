#!/bin/bash
# round 5, call f: the 8-slab shell test again (DEME_SLAB_HALO), then A/B of the L2-prefetch variants of the tile pass
out=gpurun_out/r5f; mkdir -p $out; rm -f $out/*
timeout 600 python -m pytest tests/test_host_shell.py -x -q -m gpu -k "slabs" -s > $out/new_tests.log 2>&1; tail -4 $out/new_tests.log
ROUNDS=2 timeout 1500 bash tools/gpu_ab.sh > $out/ab.log 2>&1; cat $out/ab.log
