#!/bin/bash
# A/B: the default build against every libdeme_v_*.so, interleaved rounds (box-to-box spread is ~5 %, run-to-run ~1 %)
mkdir -p gpurun_out/ab; rm -f gpurun_out/ab/*.json
R=${ROUNDS:-2}
for r in $(seq 1 $R); do
  python bench.py --no-cpu-baseline --state-cache /tmp/bed.npz > gpurun_out/ab/cur_$r.json 2>/dev/null
  for f in dem-engine_amd/csrc/libdeme_v_*.so; do
    n=$(basename $f .so); n=${n#libdeme_v_}
    DEME_HIP_LIB=$PWD/$f python bench.py --no-cpu-baseline --state-cache /tmp/bed.npz > gpurun_out/ab/${n}_$r.json 2>gpurun_out/ab/${n}_$r.err
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/ab/*.json')):
    try:
        d=json.loads(open(f).read().strip().split('\n')[-1]); k=d['kernels_ms']
        print(f"{f:36s} step {d['ms_per_step']:.4f} force {k['calc_forces']:.4f} integ {k['integrate']:.4f} det {k['detect_update']:.3f}")
    except Exception as e: print(f,'ERR',e)
PY
