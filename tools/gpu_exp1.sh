#!/bin/bash
# round-2 first GPU session: test suite, driver-shaped bench, XCD-group experiment
mkdir -p gpurun_out/exp1
python -m pytest tests -m gpu -x -q > gpurun_out/exp1/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/exp1/pytest.log
python bench.py --steps 20 --warmup 5 > gpurun_out/exp1/bench_driver.json 2> gpurun_out/exp1/bench_driver.err
python bench.py --no-cpu-baseline > gpurun_out/exp1/bench_200.json 2>> gpurun_out/exp1/bench_driver.err
for order in lattice morton; do
for g in 0 4 16 64 256; do
  DEME_XCD_GROUP=$g python bench.py --no-cpu-baseline --order $order --steps 200 --warmup 20 > gpurun_out/exp1/xcd_${order}_$g.json 2>/dev/null
done
done
tail -3 gpurun_out/exp1/pytest.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/exp1/*.json')):
    try:
        d=json.loads(open(f).read().strip().split('\n')[-1])
        print(f, round(d['ms_per_step'],4), d['kernels_ms'], d['config']['contacts_this_rank'])
    except Exception as e:
        print(f, 'ERR', e)
PY
