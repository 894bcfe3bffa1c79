#!/bin/bash
# bench every libdeme_v_*.so variant next to the default build (fast mode)
mkdir -p gpurun_out/var
python bench.py --no-cpu-baseline --state-cache /tmp/bed.npz > gpurun_out/var/default.json 2>/dev/null
for f in dem-engine_amd/csrc/libdeme_v_*.so; do
  n=$(basename $f .so); n=${n#libdeme_v_}
  DEME_HIP_LIB=$PWD/$f python bench.py --no-cpu-baseline --state-cache /tmp/bed.npz > gpurun_out/var/$n.json 2>/dev/null
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/var/*.json')):
    try:
        d=json.loads(open(f).read().strip().split('\n')[-1])
        k=d['kernels_ms']
        print(f"{f:40s} step {d['ms_per_step']:.4f} force {k['calc_forces']:.4f} integ {k['integrate']:.4f} det {k['detect_update']:.3f} nc {d['config']['contacts_this_rank']}")
    except Exception as e:
        print(f, 'ERR', e)
PY
