#!/bin/bash
mkdir -p gpurun_out/r3c
bash tools/prof.sh t1 r3c trace sqA sqB lds > gpurun_out/r3c/log.txt 2>&1
grep -E "k_tile_forces|k_integrate|^kernel" gpurun_out/r3c/log.txt | cut -c1-250
