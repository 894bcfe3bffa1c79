#!/bin/bash
# the flavours of the bench on the current tree -> gpurun_out/<round dir>/flavours.txt     usage: gpu_flavours.sh [tag] [round dir]
TAG=${1:-r06b}; RD=${2:-r06}
out=gpurun_out/$RD; mkdir -p $out
line() { python - "$@" <<'PY'
import json,sys
for f in sys.argv[1:]:
    try:
        d=json.loads(open(f).read().strip().split('\n')[-1]); k=d['kernels_ms']; r=d['roofline']
        print(f"{f.split('/')[-1]:38s} step {d['ms_per_step']:.4f} ms  value {d['value']:.3e}  force {k['calc_forces']:.4f}  integ {k['integrate']:.4f}  det {k['detect_update']:.3f}  contacts {d['config']['contacts_this_rank']}  kernel {r.get('kernel')}  frac {r.get('frac'):.3f}")
    except Exception as e: print(f,'ERR',e)
PY
}
for o in lattice morton random; do python bench.py --no-cpu-baseline --order $o > $out/${TAG}_order_$o.json 2>/dev/null; done
python bench.py --no-cpu-baseline --config5 > $out/${TAG}_fl_config5.json 2>/dev/null
python bench.py --no-cpu-baseline --config5 --tile-policy 0 > $out/${TAG}_fl_config5_tilepass.json 2>/dev/null
python bench.py --no-cpu-baseline --custom-model > $out/${TAG}_fl_custom_clumps.json 2>/dev/null
python bench.py --no-cpu-baseline --clumps 2000000 --mesh-triangles 50000 > $out/${TAG}_fl_mesh_fixed.json 2>/dev/null
python bench.py --no-cpu-baseline --clumps 2000000 --mesh-triangles 50000 --mesh-update-every 40 > $out/${TAG}_fl_mesh_deform.json 2>/dev/null
python bench.py --no-cpu-baseline --clumps 10000000 --presettle 24000 > $out/${TAG}_fl_1e7.json 2>/dev/null
python bench.py --no-cpu-baseline --async-detection 20 > $out/${TAG}_fl_async20.json 2>/dev/null
python bench.py --no-cpu-baseline --steps 20 --warmup 5 --async-detection 20 > $out/${TAG}_fl_async20_driver_shape.json 2>/dev/null
python bench.py --no-cpu-baseline --slabs 2 > $out/${TAG}_fl_slabs2.json 2>/dev/null
python bench.py --no-cpu-baseline --slabs 8 > $out/${TAG}_fl_slabs8.json 2>/dev/null
DEME_ARITH=exact python bench.py --no-cpu-baseline > $out/${TAG}_fl_exact.json 2>/dev/null
line $out/${TAG}_order_*.json $out/${TAG}_fl_*.json > $out/flavours.txt; cat $out/flavours.txt
