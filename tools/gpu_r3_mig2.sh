#!/bin/bash
out=gpurun_out/r3m; mkdir -p $out
timeout 1500 python -m pytest tests/test_config2_slabs.py -x -q -m gpu -k "migration or drifting" > $out/mig_tests.log 2>&1; echo "rc $?" >> $out/mig_tests.log; tail -3 $out/mig_tests.log
python bench.py --no-cpu-baseline --slabs 2 --steps 400 --migrate-every 100 --drift 0.5 > $out/bench_migrate.json 2>$out/bench_migrate.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3m/bench_migrate.json').read().strip().split('\n')[-1]); k=d['kernels_ms']
print(f"step {d['ms_per_step']:.4f} force {k['calc_forces']:.4f} integ {k['integrate']:.4f} det {k['detect_update']:.3f} mig {d.get('migration')}")
PY
