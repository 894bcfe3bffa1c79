#!/bin/bash
out=$PWD/gpurun_out/r3m; mkdir -p $out; ROOT=$PWD
export TMPDIR=/tmp; cd /tmp
python $ROOT/bench.py --no-cpu-baseline --state-cache /tmp/bed.npz > /dev/null 2>&1
for mode in single slabs; do
  rm -rf /tmp/prof_m
  if [ $mode = single ]; then ARGS="--state-cache /tmp/bed.npz --steps 200"; else ARGS="--slabs 2 --clumps 200000 --steps 200 --presettle 400"; fi
  rocprofv3 --hip-trace --stats --output-format csv -d /tmp/prof_m -o p -- python $ROOT/bench.py --no-cpu-baseline $ARGS > /dev/null 2> $out/m_$mode.err
  f=$(find /tmp/prof_m -name 'p_hip_api_stats.csv' | head -1)
  echo "== $mode"; grep -E "hipMalloc|hipFree|hipLaunchKernel\"|hipStreamSynchronize|hipHostMalloc" $f | cut -c1-120
done
