#!/bin/bash
out=gpurun_out/r3i; mkdir -p $out
timeout 2400 python -m pytest tests/test_config2_slabs.py -x -q -m gpu -s -k "evaluated_once or (million and True)" > $out/once_tests.log 2>&1; echo "rc $?" >> $out/once_tests.log
grep -E "one evaluation|configs\[2\]|passed|failed|rc " $out/once_tests.log | tail -12
for cc in both once; do
  python bench.py --no-cpu-baseline --slabs 2 --steps 400 --cross-contacts $cc > $out/slabs2_$cc.json 2>$out/slabs2_$cc.err
  python bench.py --no-cpu-baseline --slabs 2 --steps 400 --cross-contacts $cc --async-detection 10 > $out/slabs2_${cc}_async.json 2>/dev/null
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3i/slabs2_*.json')):
    try:
        d=json.loads(open(f).read().strip().split('\n')[-1]); k=d['kernels_ms']
        print(f"{f:44s} step {d['ms_per_step']:.4f} force {k['calc_forces']:.4f} integ {k['integrate']:.4f} det {k['detect_update']:.3f} contacts {d['config']['contacts_this_rank']}")
    except Exception as e: print(f,'ERR',e)
PY
tail -3 $out/slabs2_once.err
