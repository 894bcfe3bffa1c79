#!/bin/bash
# round 5: kernel traces of the flavours (configs[3] at its size, configs[4]) on the final tree -> gpurun_out/r05/
out=gpurun_out/r05; mkdir -p $out
BENCH_ARGS="--steps 80 --warmup 10 --no-cpu-baseline --clumps 2000000 --mesh-triangles 50000 --state-cache /tmp/deme_bed_mesh.npz" bash tools/prof.sh r05b_mesh r05 trace > $out/r05b_mesh_log.txt 2>&1
BENCH_ARGS="--steps 80 --warmup 10 --no-cpu-baseline --config5" bash tools/prof.sh r05b_config5 r05 trace > $out/r05b_config5_log.txt 2>&1
BENCH_ARGS="--steps 80 --warmup 10 --no-cpu-baseline --config5 --tile-policy 0" bash tools/prof.sh r05b_config5_tilepass r05 trace > $out/r05b_config5_tilepass_log.txt 2>&1
for t in mesh config5 config5_tilepass; do echo "== $t"; head -8 $out/r05b_${t}_kernels.txt | cut -c1-150; done
