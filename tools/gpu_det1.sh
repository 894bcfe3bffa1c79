#!/bin/bash
# detection work, iteration 1: parity suite, then bench + bin-multiple sweep + kernel trace
mkdir -p gpurun_out/det1
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/det1/pytest.txt 2>&1
tail -5 gpurun_out/det1/pytest.txt
ST="--state-cache /tmp/bed_default.npz"
for BM in 4.0 3.0 3.5 5.0; do
  python bench.py --no-cpu-baseline --steps 80 --warmup 10 --bin-multiple $BM > gpurun_out/det1/bench_bm$BM.json 2>/dev/null
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/det1/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().split('\n')[-1]); k=d['kernels_ms']
        print(f"{f:40s} step {d['ms_per_step']:.4f} force {k['calc_forces']:.4f} integ {k['integrate']:.4f} det {k['detect_update']:.3f} x{k['detect_updates']} inc {d['config']['bin_sphere_touches']} nc {d['config']['contacts_this_rank']}")
    except Exception as e:
        print(f,'ERR',e)
PY
bash tools/prof.sh det1 det1 trace
