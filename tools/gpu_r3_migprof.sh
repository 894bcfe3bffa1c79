#!/bin/bash
out=$PWD/gpurun_out/r3m; mkdir -p $out; ROOT=$PWD
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/prof_mig
rocprofv3 --kernel-trace --hip-trace --stats --output-format csv -d /tmp/prof_mig -o p -- python $ROOT/bench.py --no-cpu-baseline --slabs 2 --clumps 400000 --steps 300 --warmup 10 --migrate-every 100 --drift 0.5 > $out/mig_bench.json 2> $out/mig.err
ls /tmp/prof_mig/*/ | head
f=$(find /tmp/prof_mig -name 'p_hip_api_stats.csv' | head -1); head -25 $f | cut -c1-160 > $out/hip_api_stats.txt; cat $out/hip_api_stats.txt
f=$(find /tmp/prof_mig -name 'p_kernel_stats.csv' | head -1); head -30 $f | cut -c1-160 > $out/kernel_stats.txt; cat $out/kernel_stats.txt | grep -i "mig\|name" | head -30
tail -c 700 $out/mig_bench.json
