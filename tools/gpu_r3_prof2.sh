#!/bin/bash
mkdir -p gpurun_out/r3c
bash tools/prof.sh t2 r3c fetch write ea tcp > gpurun_out/r3c/log2.txt 2>&1
grep -E "k_tile_forces|k_integrate|^kernel|copyBuffer" gpurun_out/r3c/log2.txt | cut -c1-250
