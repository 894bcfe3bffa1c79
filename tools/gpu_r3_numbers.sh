#!/bin/bash
# the numbers DESIGN.md section 5 quotes for round 3 (one box, back to back)
out=gpurun_out/r3n; mkdir -p $out; rm -f $out/*
python bench.py --state-cache /tmp/bed.npz > $out/default.json 2>$out/default.err
python bench.py --no-cpu-baseline --state-cache /tmp/bed.npz --steps 4000 > $out/long.json 2>/dev/null
python bench.py --no-cpu-baseline --state-cache /tmp/bed.npz --steps 20 --warmup 5 > $out/driver.json 2>/dev/null
python bench.py --no-cpu-baseline --state-cache /tmp/bed.npz --async-detection 10 > $out/async10.json 2>/dev/null
DEME_ARITH=exact python bench.py --no-cpu-baseline --state-cache /tmp/bed.npz > $out/exact.json 2>/dev/null
DEME_TILE=0 python bench.py --no-cpu-baseline --state-cache /tmp/bed.npz > $out/blockkernel.json 2>/dev/null
python bench.py --no-cpu-baseline --config5 > $out/config5.json 2>/dev/null
python bench.py --no-cpu-baseline --clumps 2000000 --mesh-triangles 50000 --mesh-update-every 40 > $out/mesh.json 2>/dev/null
python bench.py --no-cpu-baseline --clumps 2000000 --mesh-triangles 50000 --async-detection 10 > $out/mesh_fixed_async.json 2>/dev/null
python bench.py --no-cpu-baseline --clumps 2000000 --mesh-triangles 50000 > $out/mesh_fixed.json 2>/dev/null
python bench.py --no-cpu-baseline --clumps 10000000 --steps 100 > $out/tenmillion.json 2>/dev/null
python bench.py --no-cpu-baseline --order random > $out/random.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3n/*.json')):
    try:
        d=json.loads(open(f).read().strip().split('\n')[-1]); k=d['kernels_ms']
        cb=d.get('cpu_baseline')
        print(f"{f.split('/')[-1]:24s} step {d['ms_per_step']:.4f} value {d['value']:.3e} force {k['calc_forces']:.4f} integ {k['integrate']:.4f} det {k['detect_update']:.3f} frac {d['roofline']['frac']:.3f} kern {d['roofline'].get('kernel')} contacts {d['config']['contacts_this_rank']} cpu {cb and (cb.get('value'), cb.get('cores'))}")
    except Exception as e: print(f,'ERR',e)
PY
