#!/bin/bash
# usage: bash profiles/run_pmc_lib.sh <tag> <lib.so> "<counters>" [bench args]  -- PMC pass with another build of the library
TAG=$1; LIB=$2; CTRS=$3; shift; shift; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_$TAG
DEME_HIP_LIB=$ROOT/dem-engine_amd/csrc/$LIB rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d /tmp/pmc_$TAG -o p -- python $ROOT/bench.py "$@" --no-cpu-baseline > /tmp/pmc_$TAG.log 2>&1 || (tail -5 /tmp/pmc_$TAG.log; exit 1)
python3 $ROOT/profiles/summarize_pmc.py $(find /tmp/pmc_$TAG -name '*counter_collection.csv' | head -1) $ROOT/gpurun_out/${TAG}_pmc.txt 40 | grep -E "^kernel|k_calc_forces"
