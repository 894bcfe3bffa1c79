#!/bin/bash
out=gpurun_out/r3d; mkdir -p $out; rm -f $out/*.json
python bench.py --no-cpu-baseline --state-cache /tmp/bed.npz > $out/pad0.json 2>/dev/null
for pad in 14000 40000 100000; do
  DEME_TILE_LDS_PAD=$pad python bench.py --no-cpu-baseline --state-cache /tmp/bed.npz > $out/pad$pad.json 2>/dev/null
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3d/*.json')):
    try:
        d=json.loads(open(f).read().strip().split('\n')[-1]); k=d['kernels_ms']
        print(f"{f:36s} step {d['ms_per_step']:.4f} force {k['calc_forces']:.4f} integ {k['integrate']:.4f} det {k['detect_update']:.3f}")
    except Exception as e: print(f,'ERR',e)
PY
