#!/bin/bash
out=gpurun_out/r5j; mkdir -p $out; rm -f $out/*
timeout 900 python -m pytest --timeout 300 tests/test_multi.py tests/test_host_shell.py -x -q -m gpu -s > $out/tests.log 2>&1; tail -6 $out/tests.log; grep -n "Error\|assert " $out/tests.log | head
