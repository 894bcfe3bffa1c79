#!/bin/bash
out=gpurun_out/r3o; mkdir -p $out
timeout 1500 python -m pytest tests/test_fast_mode.py tests/test_mesh.py tests/test_async_detection.py tests/test_decomp.py -x -q -m gpu -s > $out/tests.log 2>&1; echo "rc $?" >> $out/tests.log; grep -E "mesh scene|passed|failed|rc " $out/tests.log | tail -5
python bench.py --no-cpu-baseline --clumps 2000000 --mesh-triangles 50000 > $out/mesh_fixed.json 2>$out/mesh_fixed.err
DEME_TILE=0 python bench.py --no-cpu-baseline --clumps 2000000 --mesh-triangles 50000 > $out/mesh_fixed_notile.json 2>/dev/null
python bench.py --no-cpu-baseline --clumps 2000000 --mesh-triangles 50000 --mesh-update-every 40 > $out/mesh_deform.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3o/*.json')):
    try:
        d=json.loads(open(f).read().strip().split('\n')[-1]); k=d['kernels_ms']
        print(f"{f.split('/')[-1]:28s} step {d['ms_per_step']:.4f} value {d['value']:.3e} force {k['calc_forces']:.4f} integ {k['integrate']:.4f} det {k['detect_update']:.3f} kern {d['roofline'].get('kernel')} contacts {d['config']['contacts_this_rank']}")
    except Exception as e: print(f,'ERR',e)
PY
tail -2 $out/mesh_fixed.err
