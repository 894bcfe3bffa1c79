#!/bin/bash
# A/B of library variants (dem-engine_amd/csrc/libdeme_v_<name>.so) against the default build on one box
out=gpurun_out/r3j; mkdir -p $out
python bench.py --no-cpu-baseline --state-cache /tmp/bed.npz > /dev/null 2>&1
for r in 1 2; do
  for v in base $VARIANTS; do
    if [ $v = base ]; then unset DEME_HIP_LIB; else export DEME_HIP_LIB=$PWD/dem-engine_amd/csrc/libdeme_v_$v.so; fi
    python bench.py --no-cpu-baseline --state-cache /tmp/bed.npz > $out/ab_${v}_$r.json 2>/dev/null
    python - <<PY
import json
d=json.loads(open('$out/ab_${v}_$r.json').read().strip().split('\n')[-1]); k=d['kernels_ms']
print(f"$v $r: step {d['ms_per_step']:.4f} force {k['calc_forces']:.4f} integ {k['integrate']:.4f} det {k['detect_update']:.3f}")
PY
  done
done
