#!/bin/bash
# the whole GPU suite at HEAD (plain kernel default), the slab / decomposition tests with the persistent kernel, the multi
# tests through the per-device worker threads
out=gpurun_out/suite; mkdir -p $out; rm -f $out/*
timeout 3000 python -m pytest tests -x -q -m gpu > $out/pytest_gpu.log 2>&1; grep -E "passed|failed" $out/pytest_gpu.log | tail -2
DEME_TILE_PERSIST=1 timeout 2400 python -m pytest tests/test_full_size.py tests/test_fast_mode.py tests/test_engine_order.py tests/test_mesh.py tests/test_multi.py tests/test_config2_slabs.py -x -q -m gpu > $out/pytest_persist.log 2>&1; grep -E "passed|failed" $out/pytest_persist.log | tail -2
DEME_MULTI_FORCE_WORKERS=1 timeout 1500 python -m pytest tests/test_multi.py tests/test_host_shell.py -x -q -m gpu > $out/pytest_workers.log 2>&1; grep -E "passed|failed" $out/pytest_workers.log | tail -2
cp gpurun_out/measured_errors.txt $out/ 2>/dev/null
