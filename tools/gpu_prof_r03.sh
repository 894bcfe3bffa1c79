#!/bin/bash
# round 3: kernel trace + PMC passes of the default bench (owner-tile force pass); summaries -> gpurun_out/r03/ (copied to profiles/r03/)
TAG=${1:-r03a}
mkdir -p gpurun_out/r03
bash tools/prof.sh $TAG r03 trace sqA sqB lds fetch write tcp ea > gpurun_out/r03/${TAG}_log.txt 2>&1
tail -60 gpurun_out/r03/${TAG}_log.txt | cut -c1-220
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03/${TAG}_bench_driver.json 2> /dev/null
tail -c 600 gpurun_out/r03/${TAG}_bench_driver.json
