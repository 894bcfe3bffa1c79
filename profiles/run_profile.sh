#!/bin/bash
# usage (on the GPU box, via gpurun): bash profiles/run_profile.sh <tag> [bench args...]
# Writes gpurun_out/<tag>_kernels.txt (condensed kernel trace of the timed region) and gpurun_out/<tag>_bench.json
set -e
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o p -- python $ROOT/bench.py "$@" --no-cpu-baseline > /tmp/prof_$TAG.log 2>&1 || (tail -20 /tmp/prof_$TAG.log; exit 1)
grep '^{"metric"' /tmp/prof_$TAG.log > $ROOT/gpurun_out/${TAG}_bench.json
STEPS=$(python3 -c "import json;print(json.load(open('$ROOT/gpurun_out/${TAG}_bench.json'))['steps'])")
python3 $ROOT/profiles/summarize.py $(find /tmp/prof_$TAG -name '*kernel_trace.csv' | head -1) $ROOT/gpurun_out/${TAG}_kernels.txt $STEPS
