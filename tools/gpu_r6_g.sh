#!/bin/bash
out=gpurun_out/r6g; mkdir -p $out; rm -f $out/*
ROUNDS=3 timeout 1500 bash tools/gpu_ab.sh > $out/ab.log 2>&1; cat $out/ab.log
