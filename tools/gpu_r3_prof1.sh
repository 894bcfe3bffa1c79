#!/bin/bash
mkdir -p gpurun_out/r3c
rocprofv3 -L 2>/dev/null | grep -i -o "SQ_[A-Z_]*LDS[A-Z_]*" | sort -u > gpurun_out/r3c/lds_counters.txt
bash tools/prof.sh t1 r3c trace sqA sqB lds tcp > gpurun_out/r3c/log.txt 2>&1
tail -60 gpurun_out/r3c/log.txt | cut -c1-250
