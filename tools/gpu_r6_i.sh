#!/bin/bash
out=gpurun_out/r6i; mkdir -p $out; rm -f $out/*
bash profiles/run_profile.sh r6i_slabs8_125k --slabs 8 --clumps 125000 > $out/log.txt 2>&1
cp gpurun_out/r6i_slabs8_125k_kernels.txt gpurun_out/r6i_slabs8_125k_bench.json $out/ 2>/dev/null
cat $out/r6i_slabs8_125k_kernels.txt | cut -c1-160 | head -40
