#!/bin/bash
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_tl
R=$GRAFT_REPO_ROOT
python $R/bench.py --steps 80 --warmup 10 --no-cpu-baseline --state-cache /tmp/bed_tl.npz > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_tl -o p -- python $R/bench.py --steps 80 --warmup 10 --no-cpu-baseline --state-cache /tmp/bed_tl.npz > /tmp/tl.json 2>/tmp/tl.err
mkdir -p $R/gpurun_out/r04
python $R/profiles/timeline.py $(find /tmp/prof_tl -name 'p_kernel_trace.csv' | head -1) $R/gpurun_out/r04/${1:-r04b}_detection_timeline.txt
