#!/bin/bash
mkdir -p gpurun_out/int1
timeout 1200 python -m pytest tests -m gpu -x -q -k "not million" > gpurun_out/int1/pytest.txt 2>&1
grep -E "passed|failed|error" gpurun_out/int1/pytest.txt | tail -3
bash tools/gpu_variants.sh
for o in lattice morton; do python bench.py --no-cpu-baseline --order $o > gpurun_out/int1/order_$o.json 2>/dev/null; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/int1/order_*.json')):
    d=json.loads(open(f).read().strip().split('\n')[-1]); k=d['kernels_ms']
    print(f"{f:40s} step {d['ms_per_step']:.4f} force {k['calc_forces']:.4f} integ {k['integrate']:.4f} det {k['detect_update']:.3f}")
PY
