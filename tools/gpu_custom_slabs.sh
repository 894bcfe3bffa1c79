#!/bin/bash
# demo_custom single-domain and in two slabs: what the decomposed shell answers
mkdir -p /tmp/c1 /tmp/c2
echo '__device__ inline float demo_charge_force(float qq) { return (float)(2.5e-3 * qq); }' > /tmp/c1/demo_helpers.h
cp /tmp/c1/demo_helpers.h /tmp/c2/
cd dem-engine_amd/host
DEME_KERNEL_INCLUDE_PATH=/tmp/c1 timeout 300 ./demo_custom /tmp/c1 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl" | grep "CHECK\|DEMO\|rror\|what\|terminate" | cut -c1-300
echo ---- two slabs
DEME_SLABS_PER_DEVICE=2 DEME_KERNEL_INCLUDE_PATH=/tmp/c2 timeout 300 ./demo_custom /tmp/c2 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl" | grep "CHECK\|DEMO\|rror\|what\|terminate" | cut -c1-300
