#!/bin/bash
# fused one-kernel step: default library (HMAX 192) vs HMAX 256 variant vs DEME_FUSED=0
mkdir -p gpurun_out/r04h
P='import sys,json; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); k=d["kernels_ms"]; r=d["roofline"]; print(sys.argv[1], d["ms_per_step"], k["calc_forces"], k["integrate"], k["detect_update"], r["kernel"], round(r["frac"],3), r["tile"])'
for cfg in "1 " "0 " "1 $PWD/dem-engine_amd/csrc/libdeme_v_h256.so" "0 $PWD/dem-engine_amd/csrc/libdeme_v_h256.so"; do
  set -- $cfg
  for ord in morton random; do
    DEME_FUSED=$1 DEME_HIP_LIB=$2 python bench.py --no-cpu-baseline --order $ord 2>/dev/null | python -c "$P" "fused=$1 lib=$(basename ${2:-default}) $ord"
  done
done
