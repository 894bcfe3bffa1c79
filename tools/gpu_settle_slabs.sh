#!/bin/bash
mkdir -p /tmp/s1 /tmp/s2
cd dem-engine_amd/host
DEME_ARITH=exact timeout 300 ./demo_settle 10 3000 /tmp/s1 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl" | grep -v "^t=" | cut -c1-250
echo ---- two slabs
DEME_ARITH=exact DEME_SLABS_PER_DEVICE=2 timeout 300 ./demo_settle 10 3000 /tmp/s2 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl" | grep -v "^t=" | cut -c1-250
