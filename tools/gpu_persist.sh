#!/bin/bash
mkdir -p gpurun_out/persist
for n in 0 1024 1280 1536 2048 2560 4096; do
  DEME_FORCE_PERSIST=$n python bench.py --no-cpu-baseline --state-cache /tmp/bedp.npz > gpurun_out/persist/p_$n.json 2>/dev/null
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/persist/*.json'), key=lambda f:int(f.split('_')[-1][:-5])):
    try:
        d=json.loads(open(f).read().strip().split('\n')[-1]); k=d['kernels_ms']
        print(f"{f:40s} step {d['ms_per_step']:.4f} force {k['calc_forces']:.4f} integ {k['integrate']:.4f}")
    except Exception as e:
        print(f,'ERR',e)
PY
python -m pytest tests/test_fast_mode.py -m gpu -q 2>&1 | tail -1
DEME_FORCE_PERSIST=1280 python -m pytest tests/test_fast_mode.py tests/test_decomp.py -m gpu -q 2>&1 | tail -1
