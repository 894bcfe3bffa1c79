#!/bin/bash
# round 5, call e: the whole GPU suite on the current tree (prints of the full-size legs kept), fused-step small-bed A/B, tile variants
out=gpurun_out/r5e; mkdir -p $out; rm -f $out/*
timeout 1500 python -m pytest tests -x -q -m gpu -s > $out/gpu_suite.log 2>&1; tail -5 $out/gpu_suite.log
for n in 10000 50000 200000; do for f in 0 1; do
  DEME_FUSED=$f python bench.py --clumps $n --no-cpu-baseline --steps 400 --warmup 40 > $out/fused_${n}_$f.json 2>/dev/null
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5e/fused_*.json')):
    try:
        d=json.loads(open(f).read().strip().split('\n')[-1]); k=d['kernels_ms']
        print(f"{f:44s} step {d['ms_per_step']:.4f} force {k['calc_forces']:.4f} integ {k['integrate']:.4f} kernel {d['roofline']['kernel']}")
    except Exception as e: print(f,'ERR',e)
PY
ROUNDS=2 timeout 900 bash tools/gpu_ab.sh > $out/ab.log 2>&1; cat $out/ab.log
