#!/bin/bash
# round 4, first call: smoke + the fast-mode tests with the kernel-name asserts, then the knock-in / knock-out variants of
# k_tile_forces (deme_tile.h DEME_TILE_KI) against the default build, then the full-size tile-pass tests
out=gpurun_out/r4a; mkdir -p $out; rm -f $out/*
python __graft_entry__.py smoke > $out/smoke.log 2>&1; echo "smoke rc $?" >> $out/smoke.log; tail -3 $out/smoke.log
timeout 900 python -m pytest tests/test_fast_mode.py -x -q -m gpu > $out/fast_tests.log 2>&1; tail -4 $out/fast_tests.log
ROUNDS=2 bash tools/gpu_ab.sh > $out/ab.log 2>&1; cat $out/ab.log
timeout 1500 python -m pytest tests/test_full_size.py -x -q -m gpu -k "one_launch or fast_mode_matches" -s > $out/full_tests.log 2>&1; tail -8 $out/full_tests.log
