#!/bin/bash
# round 6, call c: how many listed contacts carry a history at all; persistent vs plain kernel bit for bit on the settled bench bed
out=gpurun_out/r6c; mkdir -p $out; rm -f $out/*
timeout 600 python bench.py --no-cpu-baseline --state-cache /tmp/bed.npz > $out/bench.json 2>$out/bench.err
python - <<'PY' > gpurun_out/r6c/history_fill.txt 2>&1
import numpy as np
z=np.load('/tmp/bed.npz'); wc=z['wc']; t=z['ctype']
nz=(wc!=0).any(1)
print('contacts',len(wc),'with any nonzero wildcard',nz.sum(),'fraction %.4f'%nz.mean())
print('delta_tan nonzero %.4f, delta_time nonzero %.4f'%((wc[:,:3]!=0).any(1).mean(),(wc[:,3]!=0).mean()))
for c in np.unique(t): print('type',c,'count',(t==c).sum(),'nonzero fraction %.4f'%nz[t==c].mean())
# run structure: how often does a 4-contact (64-byte) group hold no history at all
g=nz[:len(nz)//4*4].reshape(-1,4)
print('64-byte groups of 4 contacts with no history at all: %.4f'%(~g.any(1)).mean())
PY
cat $out/history_fill.txt
timeout 900 python tools/persist_compare.py 1000000 100 > $out/persist_compare.log 2>&1; tail -8 $out/persist_compare.log
