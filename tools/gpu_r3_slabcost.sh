#!/bin/bash
# what the slab loop costs per step on one GPU: a 2e6-clump bed as one context and as two slabs of 1e6 (RCCL to self)
out=gpurun_out/r3i; mkdir -p $out
python bench.py --no-cpu-baseline --clumps 2000000 --steps 200 > $out/one_2e6.json 2>/dev/null
python bench.py --no-cpu-baseline --clumps 2000000 --slabs 2 --steps 200 > $out/two_1e6.json 2>$out/two_1e6.err
python bench.py --no-cpu-baseline --clumps 2000000 --slabs 2 --steps 200 --no-overlap > $out/two_1e6_noov.json 2>/dev/null
python - <<'PY'
import json,glob
for f in ['one_2e6','two_1e6','two_1e6_noov']:
    try:
        d=json.loads(open(f'gpurun_out/r3i/{f}.json').read().strip().split('\n')[-1]); k=d['kernels_ms']
        print(f"{f:16s} step {d['ms_per_step']:.4f} force {k['calc_forces']:.4f} integ {k['integrate']:.4f} det {k['detect_update']:.3f} contacts {d['config']['contacts_this_rank']} host {d.get('host_enqueue_us_per_step')} {d['config'].get('parallelism')}")
    except Exception as e: print(f,'ERR',e)
PY
tail -2 $out/two_1e6.err
