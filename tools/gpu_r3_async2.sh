#!/bin/bash
out=gpurun_out/r3p; mkdir -p $out
timeout 1500 python -m pytest tests/test_async_detection.py -x -q -m gpu > $out/tests.log 2>&1; echo "rc $?" >> $out/tests.log; tail -3 $out/tests.log
python bench.py --no-cpu-baseline --state-cache /tmp/bed.npz > $out/lock.json 2>/dev/null
for D in 5 10 15 20; do python bench.py --no-cpu-baseline --state-cache /tmp/bed.npz --async-detection $D > $out/async$D.json 2>/dev/null; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3p/*.json')):
    try:
        d=json.loads(open(f).read().strip().split('\n')[-1]); k=d['kernels_ms']
        print(f"{f.split('/')[-1]:16s} step {d['ms_per_step']:.4f} force {k['calc_forces']:.4f} integ {k['integrate']:.4f} det {k['detect_update']:.3f} async {k.get('async_detection')}")
    except Exception as e: print(f,'ERR',e)
PY
