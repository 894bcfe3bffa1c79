#!/bin/bash
mkdir -p gpurun_out/xcd2
for order in lattice morton; do
for g in 0 8 32 128; do
  DEME_XCD_GROUP=$g python bench.py --no-cpu-baseline --order $order --state-cache /tmp/bed_$order.npz > gpurun_out/xcd2/${order}_$g.json 2>/dev/null
done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/xcd2/*.json')):
    try:
        d=json.loads(open(f).read().strip().split('\n')[-1])
        k=d['kernels_ms']
        print(f"{f:40s} step {d['ms_per_step']:.4f} force {k['calc_forces']:.4f} integ {k['integrate']:.4f} det {k['detect_update']:.3f} nc {d['config']['contacts_this_rank']}")
    except Exception as e:
        print(f, 'ERR', e)
PY
