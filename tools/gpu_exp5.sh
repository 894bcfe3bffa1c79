#!/bin/bash
mkdir -p gpurun_out/exp5
python -m pytest tests -m gpu -x -q > gpurun_out/exp5/pytest.log 2>&1; tail -3 gpurun_out/exp5/pytest.log
python bench.py --no-cpu-baseline > gpurun_out/exp5/fast.json 2>/dev/null
python - <<'PY'
import json
d=json.loads(open('gpurun_out/exp5/fast.json').read().strip().split('\n')[-1]); print(round(d['ms_per_step'],4), d['kernels_ms'])
PY
timeout 600 bash tools/prof.sh det prof2 trace
