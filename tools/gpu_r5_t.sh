#!/bin/bash
out=gpurun_out/r5t; mkdir -p $out; rm -f $out/*
for v in t320 t384; do DEME_HIP_LIB=$PWD/dem-engine_amd/csrc/libdeme_v_$v.so timeout 300 python __graft_entry__.py smoke > $out/smoke_$v.log 2>&1; tail -2 $out/smoke_$v.log; done
ROUNDS=2 timeout 900 bash tools/gpu_ab.sh > $out/ab.log 2>&1; cat $out/ab.log
