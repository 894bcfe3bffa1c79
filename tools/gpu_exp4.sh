#!/bin/bash
mkdir -p gpurun_out/exp4
python -m pytest tests/test_fast_mode.py -m gpu -q -s > gpurun_out/exp4/pytest_fast.log 2>&1; echo "rc $?" >> gpurun_out/exp4/pytest_fast.log
grep -E "fast vs|passed|failed|Error|^E  |max \||wildcard [0-9]" gpurun_out/exp4/pytest_fast.log | head -40
python bench.py --no-cpu-baseline > gpurun_out/exp4/fast.json 2>gpurun_out/exp4/fast.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/exp4/*.json')):
    try:
        d=json.loads(open(f).read().strip().split('\n')[-1])
        print(f, round(d['ms_per_step'],4), d['kernels_ms'], d['config']['contacts_this_rank'], round(d['roofline']['frac'],3))
    except Exception as e:
        print(f, 'ERR', e)
PY
tail -3 gpurun_out/exp4/fast.err
