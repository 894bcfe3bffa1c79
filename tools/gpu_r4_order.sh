#!/bin/bash
# the engine-side order: its own tests, the fast-mode suites (now translated at the boundary), the exact-mode parity core, bench by order
out=gpurun_out/r4c; mkdir -p $out; rm -f $out/*
python __graft_entry__.py smoke > $out/smoke.log 2>&1; echo "smoke rc $?" >> $out/smoke.log; tail -3 $out/smoke.log
timeout 1200 python -m pytest tests/test_engine_order.py -x -q -m gpu -s > $out/order_tests.log 2>&1; tail -12 $out/order_tests.log
timeout 1500 python -m pytest tests/test_fast_mode.py tests/test_fast_mode_features.py tests/test_gpu_parity.py tests/test_inspect.py tests/test_prescription.py tests/test_io.py -q -m gpu > $out/suite.log 2>&1; tail -8 $out/suite.log
for o in morton lattice random; do
  python bench.py --no-cpu-baseline --order $o > $out/bench_$o.json 2>$out/bench_$o.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4c/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().split('\n')[-1]); k=d['kernels_ms']
        print(f"{f:36s} step {d['ms_per_step']:.4f} force {k['calc_forces']:.4f} integ {k['integrate']:.4f} det {k['detect_update']:.3f} kernel {d['roofline'].get('kernel')}")
    except Exception as e: print(f,'ERR',e)
PY
