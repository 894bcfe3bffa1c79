#!/bin/bash
# A/B of two library builds on one box: alternate runs of the default bench
out=gpurun_out/r3h; mkdir -p $out
python bench.py --no-cpu-baseline --state-cache /tmp/bed.npz > /dev/null 2>&1
for r in 1 2 3; do
  for v in new old; do
    if [ $v = old ]; then export DEME_HIP_LIB=$PWD/tools/probes/libdeme_old.so; else unset DEME_HIP_LIB; fi
    python bench.py --no-cpu-baseline --state-cache /tmp/bed.npz > $out/ab_${v}_$r.json 2>/dev/null
    python - <<PY
import json
d=json.loads(open('$out/ab_${v}_$r.json').read().strip().split('\n')[-1]); k=d['kernels_ms']
print(f"$v $r: step {d['ms_per_step']:.4f} force {k['calc_forces']:.4f} integ {k['integrate']:.4f} det {k['detect_update']:.3f}")
PY
  done
done
