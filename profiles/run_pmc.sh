#!/bin/bash
# usage (GPU box): bash profiles/run_pmc.sh <tag> "<COUNTER1 COUNTER2 ...>" [bench args...]
# One rocprofv3 --pmc pass (counters only, with --kernel-trace) -> gpurun_out/<tag>_pmc.txt
set -e
TAG=$1; CTRS=$2; shift; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_$TAG
rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d /tmp/pmc_$TAG -o p -- python $ROOT/bench.py "$@" --no-cpu-baseline > /tmp/pmc_$TAG.log 2>&1 || (tail -20 /tmp/pmc_$TAG.log; exit 1)
STEPS=$(grep '^{"metric"' /tmp/pmc_$TAG.log | python3 -c "import json,sys;print(json.loads(sys.stdin.read())['steps'])")
python3 $ROOT/profiles/summarize_pmc.py $(find /tmp/pmc_$TAG -name '*counter_collection.csv' | head -1) $ROOT/gpurun_out/${TAG}_pmc.txt $STEPS
