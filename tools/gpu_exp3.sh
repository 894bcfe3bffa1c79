#!/bin/bash
mkdir -p gpurun_out/exp3
python -m pytest tests/test_fast_mode.py -m gpu -q -s > gpurun_out/exp3/pytest_fast.log 2>&1; echo "rc $?" >> gpurun_out/exp3/pytest_fast.log
grep -E "fast vs|passed|failed|Error|assert" gpurun_out/exp3/pytest_fast.log | head -30
timeout 900 bash tools/prof.sh fast prof1 trace sqA sqB fetch write tcp ea ta
