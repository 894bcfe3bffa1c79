#!/bin/bash
# round 6, call b: the persistent tile kernel (DEME_TILE_PERSIST=1, the default) against the one-workgroup-per-tile kernel (=0)
out=gpurun_out/r6b; mkdir -p $out; rm -f $out/*
timeout 600 python __graft_entry__.py smoke > $out/smoke.log 2>&1; tail -3 $out/smoke.log
for r in 1 2 3; do
  for pe in 0 1; do
    DEME_TILE_PERSIST=$pe timeout 600 python bench.py --no-cpu-baseline --state-cache /tmp/bed.npz > $out/persist${pe}_$r.json 2>$out/persist${pe}_$r.err
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r6b/persist*.json')):
    try:
        d=json.loads(open(f).read().strip().split('\n')[-1]); k=d['kernels_ms']
        print(f"{f:40s} step {d['ms_per_step']:.4f} force {k['calc_forces']:.4f} integ {k['integrate']:.4f} det {k['detect_update']:.3f}")
    except Exception as e: print(f,'ERR',e)
PY
DEME_HIP_LIB=$PWD/dem-engine_amd/csrc/libdeme_s_stamps.so DEME_TILE_STAMPS_FILE=$PWD/$out/stamps_p.bin:150 timeout 600 python bench.py --no-cpu-baseline --state-cache /tmp/bed.npz > $out/stamps_bench.json 2>$out/stamps.err
timeout 900 python tools/persist_compare.py 200000 60 > $out/persist_compare.log 2>&1; tail -8 $out/persist_compare.log
