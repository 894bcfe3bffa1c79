#!/bin/bash
# round 6, call a: knock-ins (one more load level, one more barrier per round), stream depth 1, per-tile phase stamps
out=gpurun_out/r6a; mkdir -p $out; rm -f $out/*
timeout 600 python __graft_entry__.py smoke > $out/smoke.log 2>&1; tail -2 $out/smoke.log
ROUNDS=2 timeout 1500 bash tools/gpu_ab.sh > $out/ab.log 2>&1; cat $out/ab.log
DEME_HIP_LIB=$PWD/dem-engine_amd/csrc/libdeme_s_stamps.so DEME_TILE_STAMPS_FILE=$PWD/$out/stamps.bin:150 timeout 600 python bench.py --no-cpu-baseline --state-cache /tmp/bed.npz > $out/stamps_bench.json 2>$out/stamps.err
ls -la $out
