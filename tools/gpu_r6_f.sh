#!/bin/bash
# round 6, call f: the whole GPU suite at HEAD (plain kernel default), then the persistent kernel on the tests that name the tile pass,
# and the multi tests through the per-device worker threads
out=gpurun_out/r6f; mkdir -p $out; rm -f $out/*
timeout 3000 python -m pytest tests -x -q -m gpu > $out/pytest_gpu.log 2>&1; tail -5 $out/pytest_gpu.log
DEME_TILE_PERSIST=1 timeout 1500 python -m pytest tests/test_full_size.py tests/test_fast_mode.py tests/test_engine_order.py tests/test_mesh.py -x -q -m gpu > $out/pytest_persist.log 2>&1; tail -4 $out/pytest_persist.log
DEME_MULTI_FORCE_WORKERS=1 timeout 1500 python -m pytest tests/test_multi.py tests/test_host_shell.py -x -q -m gpu > $out/pytest_workers.log 2>&1; tail -4 $out/pytest_workers.log
cp gpurun_out/measured_errors.txt $out/ 2>/dev/null
