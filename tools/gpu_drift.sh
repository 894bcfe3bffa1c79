#!/bin/bash
cd tests/clients
for s in 1 3; do
  echo "---- slabs $s"
  DEME_ARITH=exact DEME_SLABS_PER_DEVICE=$s timeout 300 ./demo_drift 1500 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl" | grep "SLABS\|SUM\|DRIFT\|rror\|what\|POS 0 \|POS 37 \|POS 3700\|drift\]"
done
