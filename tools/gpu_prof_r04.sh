#!/bin/bash
# round 4: kernel trace + PMC passes of the default bench; summaries -> gpurun_out/r04/ (copied to profiles/r04/)
TAG=${1:-r04a}
mkdir -p gpurun_out/r04
bash tools/prof.sh $TAG r04 trace sqA sqB lds fetch write tcp ea > gpurun_out/r04/${TAG}_log.txt 2>&1
tail -40 gpurun_out/r04/${TAG}_log.txt | cut -c1-200
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r04/${TAG}_bench_driver.json 2> /dev/null
tail -c 400 gpurun_out/r04/${TAG}_bench_driver.json
