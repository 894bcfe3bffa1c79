#!/bin/bash
# round 3: the owner-tile force pass against the round-2 kernels (DEME_TILE=0), fast-mode parity first
out=gpurun_out/r3b; mkdir -p $out; rm -f $out/*.json
python __graft_entry__.py smoke > $out/smoke.log 2>&1; echo "smoke rc $?" >> $out/smoke.log
timeout 1200 python -m pytest tests/test_fast_mode.py tests/test_fast_mode_features.py -x -q -m gpu > $out/fast_tests.log 2>&1
tail -5 $out/smoke.log; tail -15 $out/fast_tests.log
for r in 1 2; do
  DEME_TILE=1 python bench.py --no-cpu-baseline --state-cache /tmp/bed.npz > $out/tile_$r.json 2>$out/tile_$r.err
  for g in ${XCDS:-4 16}; do
    DEME_XCD_GROUP=$g DEME_TILE=1 python bench.py --no-cpu-baseline --state-cache /tmp/bed.npz > $out/tile_xcd${g}_$r.json 2>/dev/null
  done
  DEME_TILE=0 python bench.py --no-cpu-baseline --state-cache /tmp/bed.npz > $out/old_$r.json 2>$out/old_$r.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3b/*.json')):
    try:
        d=json.loads(open(f).read().strip().split('\n')[-1]); k=d['kernels_ms']
        print(f"{f:36s} step {d['ms_per_step']:.4f} force {k['calc_forces']:.4f} integ {k['integrate']:.4f} det {k['detect_update']:.3f}")
    except Exception as e: print(f,'ERR',e)
PY
tail -3 $out/tile_1.err
