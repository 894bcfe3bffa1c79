#!/bin/bash
out=gpurun_out/r6h; mkdir -p $out; rm -f $out/*
for n in 20000 250000 2000000; do timeout 600 python tools/host_enqueue_probe.py $n >> $out/host_enqueue.txt 2>&1; done
cat $out/host_enqueue.txt | grep slab
timeout 900 python bench.py --no-cpu-baseline --slabs 8 --clumps 125000 > $out/slabs8_125k.json 2>$out/slabs8_125k.err
timeout 900 python bench.py --no-cpu-baseline --slabs 8 --clumps 125000 --cross-contacts once > $out/slabs8_125k_once.json 2>$out/slabs8_125k_once.err
python - <<'PY'
import json
for f in ['slabs8_125k','slabs8_125k_once']:
    try:
        d=json.loads(open('gpurun_out/r6h/'+f+'.json').read().strip().split('\n')[-1]); print(f, 'ms/step', d['ms_per_step'], d.get('halo_loop'), d['kernels_ms'])
    except Exception as e: print(f,'ERR',e)
PY
