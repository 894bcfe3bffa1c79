#!/bin/bash
# PMC passes of the flavours the review asked to keep: configs[3] (mesh variant of the tile pass) and configs[4] (run-time compiled model)
TAG=${1:-r04r}; out=gpurun_out/r04; mkdir -p $out
BENCH_ARGS="--steps 80 --warmup 10 --no-cpu-baseline --clumps 2000000 --mesh-triangles 50000 --state-cache /tmp/deme_bed_mesh.npz" bash tools/prof.sh ${TAG}_mesh r04 sqA lds fetch write > $out/${TAG}_mesh_pmc_log.txt 2>&1
BENCH_ARGS="--steps 80 --warmup 10 --no-cpu-baseline --config5" bash tools/prof.sh ${TAG}_config5 r04 sqA lds fetch write > $out/${TAG}_config5_pmc_log.txt 2>&1
grep -E "^kernel|k_tile_forces|k_calc_forces|deme_custom|k_integrate" $out/${TAG}_mesh_sqA_pmc.txt $out/${TAG}_config5_sqA_pmc.txt | cut -c1-220
