#!/bin/bash
mkdir -p gpurun_out/r02c
bash tools/prof.sh r02c r02c trace sqA sqB fetch write tcp ea > gpurun_out/r02c/log.txt 2>&1
tail -40 gpurun_out/r02c/log.txt | cut -c1-200
python bench.py --steps 20 --warmup 5 > gpurun_out/r02c/bench_driver.json 2> gpurun_out/r02c/bench_driver.err
tail -c 1500 gpurun_out/r02c/bench_driver.json
