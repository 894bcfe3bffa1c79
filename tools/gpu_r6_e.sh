#!/bin/bash
# round 6, call e: skip flags checked behind the first loads (cur) against in front of them (early), plain kernel
out=gpurun_out/r6e; mkdir -p $out; rm -f $out/*
ROUNDS=3 timeout 1500 bash tools/gpu_ab.sh > $out/ab.log 2>&1; cat $out/ab.log
