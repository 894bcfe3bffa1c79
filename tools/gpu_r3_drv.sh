#!/bin/bash
out=gpurun_out/r3q; mkdir -p $out; rm -f $out/*.json
python bench.py --no-cpu-baseline --state-cache /tmp/bed.npz > /dev/null 2>&1
for r in 1 2; do
python bench.py --no-cpu-baseline --state-cache /tmp/bed.npz --steps 20 --warmup 5 > $out/drv_lock_$r.json 2>/dev/null
for D in 10 15; do python bench.py --no-cpu-baseline --state-cache /tmp/bed.npz --steps 20 --warmup 5 --async-detection $D > $out/drv_async${D}_$r.json 2>/dev/null; done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3q/*.json')):
    try:
        d=json.loads(open(f).read().strip().split('\n')[-1]); k=d['kernels_ms']
        print(f"{f.split('/')[-1]:22s} step {d['ms_per_step']:.4f} value {d['value']:.3e} force {k['calc_forces']:.4f} integ {k['integrate']:.4f} det {k['detect_update']:.3f} n_det {k['detections_in_timed_region']} async {k.get('async_detection')}")
    except Exception as e: print(f,'ERR',e)
PY
