#!/bin/bash
out=gpurun_out/r3u; mkdir -p $out
python bench.py --no-cpu-baseline --state-cache /tmp/bed.npz > /dev/null 2>&1
for r in 1 2 3; do for v in 1 0; do
  DEME_SPIN_SYNC=$v python bench.py --no-cpu-baseline --state-cache /tmp/bed.npz --steps 20 --warmup 5 > $out/drv_$v_$r.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open('$out/drv_$v_$r.json').read().strip().split('\n')[-1]); k=d['kernels_ms']
print(f"spin=$v run $r: step {d['ms_per_step']:.4f} force {k['calc_forces']:.4f} integ {k['integrate']:.4f} det {k['detect_update']:.3f}")
PY
done; done
