#!/bin/bash
# round 5, call k: the whole GPU suite on the current tree, then the profile set r05b (24-byte records in)
out=gpurun_out/r5k; mkdir -p $out; rm -f $out/* gpurun_out/measured_errors.txt
timeout 1500 python -m pytest tests -x -q -m gpu --timeout 600 > $out/gpu_suite.log 2>&1; tail -4 $out/gpu_suite.log
bash tools/gpu_r5_prof.sh r05b > $out/prof.log 2>&1; tail -4 $out/prof.log
