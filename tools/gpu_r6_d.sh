#!/bin/bash
# round 6, call d: store knock-outs of the persistent kernel (bytes elasticity) and contiguous partitions; PERSIST=0 rows are the plain kernel of the same library
out=gpurun_out/r6d; mkdir -p $out; rm -f $out/*
run() { # name lib persist
  DEME_TILE_PERSIST=$3 DEME_HIP_LIB=$2 timeout 600 python bench.py --no-cpu-baseline --state-cache /tmp/bed.npz > $out/$1.json 2>$out/$1.err
}
L=$PWD/dem-engine_amd/csrc
for r in 1 2; do
  run cur_p1_$r $L/libdeme_hip.so 1
  run cur_p0_$r $L/libdeme_hip.so 0
  for v in ko16 ko32 ko64 ko112 contig; do
    run ${v}_p1_$r $L/libdeme_v_$v.so 1
  done
  run ko112_p0_$r $L/libdeme_v_ko112.so 0
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r6d/*.json')):
    try:
        d=json.loads(open(f).read().strip().split('\n')[-1]); k=d['kernels_ms']
        print(f"{f:40s} step {d['ms_per_step']:.4f} force {k['calc_forces']:.4f} integ {k['integrate']:.4f} det {k['detect_update']:.3f}")
    except Exception as e: print(f,'ERR',e)
PY
