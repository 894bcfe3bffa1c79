#!/bin/bash
mkdir -p gpurun_out/det2
cd tools/sortbench
for v in 8_512_12 10_512_12 11_512_12 11_256_16 11_1024_8 11_512_16; do
  for b in 23 22 21; do ./sb_$v 10237635 $b; done
  ./sb_$v 4340672 20
  ./sb_$v 8170190 22
done > ../../gpurun_out/det2/sortbench.txt 2>&1
cd ../..
cat gpurun_out/det2/sortbench.txt
for BM in 4.0 5.0 6.0 8.0; do
  python bench.py --no-cpu-baseline --steps 80 --warmup 10 --bin-multiple $BM > gpurun_out/det2/bench_bm$BM.json 2>/dev/null
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/det2/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().split('\n')[-1]); k=d['kernels_ms']
        print(f"{f:40s} step {d['ms_per_step']:.4f} force {k['calc_forces']:.4f} integ {k['integrate']:.4f} det {k['detect_update']:.3f} x{k['detect_updates']} inc {d['config']['bin_sphere_touches']} nc {d['config']['contacts_this_rank']}")
    except Exception as e:
        print(f,'ERR',e)
PY
BENCH_ARGS="--steps 80 --warmup 10 --no-cpu-baseline --bin-multiple 4.0" bash tools/prof.sh bm4 det2 trace | grep -v "^#" | head -12
BENCH_ARGS="--steps 80 --warmup 10 --no-cpu-baseline --bin-multiple 6.0" bash tools/prof.sh bm6 det2 trace | grep -v "^#" | head -12
