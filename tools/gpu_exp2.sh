#!/bin/bash
mkdir -p gpurun_out/exp2
python -m pytest tests/test_fast_mode.py -m gpu -x -q -s > gpurun_out/exp2/pytest_fast.log 2>&1; echo "rc $?" >> gpurun_out/exp2/pytest_fast.log
tail -25 gpurun_out/exp2/pytest_fast.log
python bench.py --no-cpu-baseline > gpurun_out/exp2/fast.json 2>gpurun_out/exp2/fast.err
DEME_ARITH=exact python bench.py --no-cpu-baseline > gpurun_out/exp2/exact.json 2>/dev/null
DEME_HIP_LIB=$PWD/dem-engine_amd/csrc/libdeme_hip_noslp.so python bench.py --no-cpu-baseline > gpurun_out/exp2/fast_noslp.json 2>/dev/null
DEME_ARITH=exact DEME_HIP_LIB=$PWD/dem-engine_amd/csrc/libdeme_hip_noslp.so python bench.py --no-cpu-baseline > gpurun_out/exp2/exact_noslp.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/exp2/*.json')):
    try:
        d=json.loads(open(f).read().strip().split('\n')[-1])
        print(f, round(d['ms_per_step'],4), d['kernels_ms'], d['config']['contacts_this_rank'], round(d['roofline']['frac'],3))
    except Exception as e:
        print(f, 'ERR', e)
PY
tail -5 gpurun_out/exp2/fast.err
