#!/bin/bash
# round 4, closing set: kernel trace + PMC passes + detection timeline of the default bench, the bench by hand-over order, the
# flavours (configs[3], configs[4], user model on clumps, 1e7 clumps), the driver's shape, the default line with its CPU baseline.
# Summaries -> gpurun_out/r04/ (copied to profiles/r04/).
TAG=${1:-r04g}
out=gpurun_out/r04; mkdir -p $out
bash tools/prof.sh $TAG r04 trace sqA sqB lds fetch write tcp ea > $out/${TAG}_log.txt 2>&1
bash tools/gpu_r4_tl.sh ${TAG} > /dev/null 2>&1
line() { python - "$@" <<'PY'
import json,sys
for f in sys.argv[1:]:
    try:
        d=json.loads(open(f).read().strip().split('\n')[-1]); k=d['kernels_ms']; r=d['roofline']
        print(f"{f.split('/')[-1]:34s} step {d['ms_per_step']:.4f} ms  value {d['value']:.3e}  force {k['calc_forces']:.4f}  integ {k['integrate']:.4f}  det {k['detect_update']:.3f}  contacts {d['config']['contacts_this_rank']}  kernel {r.get('kernel')}  frac {r.get('frac'):.3f}  tile {r.get('tile')}")
    except Exception as e: print(f,'ERR',e)
PY
}
for o in lattice morton random; do python bench.py --no-cpu-baseline --order $o > $out/${TAG}_order_$o.json 2>/dev/null; done
line $out/${TAG}_order_*.json > $out/order_bench.txt; cat $out/order_bench.txt
python bench.py --no-cpu-baseline --config5 > $out/${TAG}_fl_config5.json 2>/dev/null
python bench.py --no-cpu-baseline --config5 --tile-policy 0 > $out/${TAG}_fl_config5_tilepass.json 2>/dev/null
python bench.py --no-cpu-baseline --custom-model > $out/${TAG}_fl_custom_clumps.json 2>/dev/null
python bench.py --no-cpu-baseline --custom-model --tile-policy 100000 > $out/${TAG}_fl_custom_clumps_general.json 2>/dev/null
python bench.py --no-cpu-baseline --clumps 2000000 --mesh-triangles 50000 > $out/${TAG}_fl_mesh_fixed.json 2>/dev/null
python bench.py --no-cpu-baseline --clumps 2000000 --mesh-triangles 50000 --mesh-update-every 40 > $out/${TAG}_fl_mesh_deform.json 2>/dev/null
python bench.py --no-cpu-baseline --clumps 10000000 --presettle 24000 > $out/${TAG}_fl_1e7.json 2>/dev/null
python bench.py --no-cpu-baseline --async-detection 20 > $out/${TAG}_fl_async20.json 2>/dev/null
DEME_ARITH=exact python bench.py --no-cpu-baseline > $out/${TAG}_fl_exact.json 2>/dev/null
line $out/${TAG}_fl_*.json > $out/flavours.txt; cat $out/flavours.txt
# kernel traces of the flavours the review asked to keep: configs[3] (mesh variant of the tile pass) and configs[4] (run-time compiled model)
BENCH_ARGS="--steps 80 --warmup 10 --no-cpu-baseline --clumps 2000000 --mesh-triangles 50000 --state-cache /tmp/deme_bed_mesh.npz" bash tools/prof.sh ${TAG}_mesh r04 trace > $out/${TAG}_mesh_log.txt 2>&1
BENCH_ARGS="--steps 80 --warmup 10 --no-cpu-baseline --config5" bash tools/prof.sh ${TAG}_config5 r04 trace > $out/${TAG}_config5_log.txt 2>&1
BENCH_ARGS="--steps 80 --warmup 10 --no-cpu-baseline --config5 --tile-policy 0" bash tools/prof.sh ${TAG}_config5_tilepass r04 trace > $out/${TAG}_config5_tilepass_log.txt 2>&1
python bench.py --steps 20 --warmup 5 > $out/final_bench_driver_shape.json 2>/dev/null
python bench.py > $out/final_bench_default.json 2>/dev/null
line $out/final_bench_*.json
