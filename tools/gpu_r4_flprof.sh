#!/bin/bash
# kernel traces of the flavours (configs[3] mesh variant, configs[4] run-time compiled model) on their own
TAG=${1:-r04k}; out=gpurun_out/r04; mkdir -p $out
BENCH_ARGS="--steps 80 --warmup 10 --no-cpu-baseline --clumps 2000000 --mesh-triangles 50000 --state-cache /tmp/deme_bed_mesh.npz" bash tools/prof.sh ${TAG}_mesh r04 trace > $out/${TAG}_mesh_log.txt 2>&1
BENCH_ARGS="--steps 80 --warmup 10 --no-cpu-baseline --config5" bash tools/prof.sh ${TAG}_config5 r04 trace > $out/${TAG}_config5_log.txt 2>&1
BENCH_ARGS="--steps 80 --warmup 10 --no-cpu-baseline --config5 --tile-policy 0" bash tools/prof.sh ${TAG}_config5_tilepass r04 trace > $out/${TAG}_config5_tilepass_log.txt 2>&1
head -12 $out/${TAG}_mesh_kernels.txt $out/${TAG}_config5_kernels.txt $out/${TAG}_config5_tilepass_kernels.txt | cut -c1-150
