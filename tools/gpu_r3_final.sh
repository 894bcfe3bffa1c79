#!/bin/bash
# round 3, end-of-round set: whole GPU suite, smoke, the default bench, the driver's shape, the migration flavour
out=gpurun_out/r3g; mkdir -p $out; rm -f $out/*
python __graft_entry__.py smoke > $out/smoke.log 2>&1; echo "smoke rc $?" >> $out/smoke.log
timeout 2400 python -m pytest tests -x -q -m gpu > $out/gpu_tests.log 2>&1; echo "pytest rc $?" >> $out/gpu_tests.log
python bench.py --state-cache /tmp/bed.npz > $out/bench_default.json 2>$out/bench_default.err
python bench.py --no-cpu-baseline --steps 20 --warmup 5 --state-cache /tmp/bed.npz > $out/bench_driver.json 2>/dev/null
python bench.py --no-cpu-baseline --async-detection 10 --state-cache /tmp/bed.npz > $out/bench_async.json 2>/dev/null
python bench.py --no-cpu-baseline --slabs 2 --steps 400 --migrate-every 100 --drift 0.5 > $out/bench_migrate.json 2>$out/bench_migrate.err
python bench.py --no-cpu-baseline --slabs 2 --steps 400 > $out/bench_slabs2.json 2>/dev/null
tail -4 $out/smoke.log; tail -6 $out/gpu_tests.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3g/*.json')):
    try:
        d=json.loads(open(f).read().strip().split('\n')[-1]); k=d['kernels_ms']
        print(f"{f:40s} step {d['ms_per_step']:.4f} force {k['calc_forces']:.4f} integ {k['integrate']:.4f} det {k['detect_update']:.3f} frac {d['roofline']['frac']:.3f} mig {d.get('migration')} cpu {d.get('cpu_baseline') and d['cpu_baseline'].get('value')}")
    except Exception as e: print(f,'ERR',e)
PY
tail -3 $out/bench_migrate.err
