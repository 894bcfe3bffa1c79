#!/bin/bash
mkdir -p gpurun_out/r3c
bash tools/prof.sh t3 r3c trace > gpurun_out/r3c/log3.txt 2>&1
cat gpurun_out/r3c/t3_kernels.txt | cut -c1-150
