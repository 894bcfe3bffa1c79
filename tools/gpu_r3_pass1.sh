#!/bin/bash
out=gpurun_out/r3l; mkdir -p $out
timeout 1500 python -m pytest tests/test_config2_slabs.py tests/test_async_detection.py -x -q -m gpu -k "not ten_million and not million" > $out/tests.log 2>&1; echo "rc $?" >> $out/tests.log; tail -3 $out/tests.log
for r in 1 2; do for v in 1 0; do
  DEME_PASS1_BESIDE=$v python bench.py --no-cpu-baseline --clumps 2000000 --slabs 2 --steps 200 > $out/two_$v_$r.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open('$out/two_$v_$r.json').read().strip().split('\n')[-1]); k=d['kernels_ms']
print(f"beside=$v run $r: step {d['ms_per_step']:.4f} force {k['calc_forces']:.4f} integ {k['integrate']:.4f} det {k['detect_update']:.3f}")
PY
done; done
