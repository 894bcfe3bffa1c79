#!/bin/bash
out=gpurun_out/r4e; mkdir -p $out; rm -f $out/*
for D in 0 8 10 15 20; do
  python bench.py --no-cpu-baseline --state-cache /tmp/bed.npz --async-detection $D > $out/def_D$D.json 2>/dev/null
  python bench.py --no-cpu-baseline --state-cache /tmp/bed.npz --steps 20 --warmup 5 --async-detection $D > $out/drv_D$D.json 2>/dev/null
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4e/*.json')):
    try:
        d=json.loads(open(f).read().strip().split('\n')[-1]); k=d['kernels_ms']
        print(f"{f:36s} step {d['ms_per_step']:.4f} force {k['calc_forces']:.4f} integ {k['integrate']:.4f} det {k['detect_update']:.3f} ndet {k['detections_in_timed_region']} async {k.get('async_detection')}")
    except Exception as e: print(f,'ERR',e)
PY
