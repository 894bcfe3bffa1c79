#!/bin/bash
mkdir -p gpurun_out/final
python bench.py --steps 20 --warmup 5 > gpurun_out/final/bench_driver.json 2> gpurun_out/final/bench_driver.err
python bench.py --no-cpu-baseline > gpurun_out/final/bench_200.json 2>/dev/null
DEME_ARITH=exact python bench.py --no-cpu-baseline > gpurun_out/final/bench_200_exact.json 2>/dev/null
python bench.py --no-cpu-baseline --clumps 2000000 --mesh-triangles 50000 --mesh-update-every 40 > gpurun_out/final/config3_2M_deformable.json 2>gpurun_out/final/config3.err
python bench.py --no-cpu-baseline --config5 > gpurun_out/final/config5.json 2>/dev/null
python bench.py --no-cpu-baseline --steps 4000 > gpurun_out/final/bench_4000.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/final/*.json')):
    try:
        d=json.loads(open(f).read().strip().split('\n')[-1]); k=d['kernels_ms']
        print(f"{f:48s} step {d['ms_per_step']:.4f} value {d['value']:.3e} force {k['calc_forces']:.4f} integ {k['integrate']:.4f} det {k['detect_update']:.3f} x{k['detect_updates']} frac {d['roofline']['frac']:.3f} nc {d['config']['contacts_this_rank']}")
        if d.get('cpu_baseline'): print('   cpu', json.dumps(d['cpu_baseline'])[:600])
    except Exception as e:
        print(f,'ERR',e)
PY
tail -3 gpurun_out/final/config3.err
