#!/bin/bash
out=gpurun_out/r4d; mkdir -p $out; rm -f $out/*
timeout 1500 python -m pytest tests/test_force_hook.py tests/test_engine_order.py -x -q -m gpu > $out/hook.log 2>&1; tail -4 $out/hook.log
python bench.py --no-cpu-baseline --config5 > $out/bench_c5_default.json 2>$out/bench_c5.err; tail -2 $out/bench_c5.err
python bench.py --no-cpu-baseline --config5 --tile-policy 0 > $out/bench_c5_tile.json 2>/dev/null
python bench.py --no-cpu-baseline --custom-model > $out/bench_cm_tile.json 2>$out/bench_cm.err; tail -2 $out/bench_cm.err
python bench.py --no-cpu-baseline --custom-model --tile-policy 100000 > $out/bench_cm_general.json 2>/dev/null
python bench.py --no-cpu-baseline > $out/bench_default.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4d/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().split('\n')[-1]); k=d['kernels_ms']
        print(f"{f:40s} step {d['ms_per_step']:.4f} force {k['calc_forces']:.4f} integ {k['integrate']:.4f} det {k['detect_update']:.3f} nc {d['config']['contacts_this_rank']} kernel {d['roofline'].get('kernel')} frac {d['roofline'].get('frac'):.3f}")
    except Exception as e: print(f,'ERR',e)
PY
