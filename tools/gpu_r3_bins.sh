#!/bin/bash
# detection time against the bin size (the state cache holds the settled bed; the bin size is a parameter of the detection only)
out=gpurun_out/r3h; mkdir -p $out
python bench.py --no-cpu-baseline --state-cache /tmp/bed.npz > /dev/null 2>&1
for bm in ${BMS:-3.0 3.5 4.0 4.5 5.0 6.0}; do
  python bench.py --no-cpu-baseline --bin-multiple $bm --state-cache /tmp/bed.npz > $out/bins_$bm.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open('$out/bins_$bm.json').read().strip().split('\n')[-1]); k=d['kernels_ms']
print(f"bin multiple $bm: step {d['ms_per_step']:.4f} force {k['calc_forces']:.4f} integ {k['integrate']:.4f} det {k['detect_update']:.3f} touches {d['config']['bin_sphere_touches']}")
PY
done
