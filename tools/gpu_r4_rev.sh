#!/bin/bash
# single evaluation of cross-cut contacts: reverse exchange beside the integration (DEME_REV_SPLIT=1, default) vs before it, vs double evaluation
python -m pytest tests/test_config2_slabs.py -q -m gpu -x -k "cross or once" 2>&1 | grep -E "passed|failed" | tail -2
P='import sys,json; d=json.loads(sys.stdin.read()); k=d["kernels_ms"]; print(sys.argv[1], d["ms_per_step"], k["calc_forces"], k["integrate"], d["config"].get("cross_cut_contacts"), d["config"].get("contacts_this_rank"))'
for sl in 2 8; do
  python bench.py --no-cpu-baseline --slabs $sl --clumps 500000 --cross-contacts both 2>/dev/null | grep '^{' | tail -1 | python -c "$P" "slabs=$sl both"
  for sp in 0 1; do DEME_REV_SPLIT=$sp python bench.py --no-cpu-baseline --slabs $sl --clumps 500000 --cross-contacts once 2>/dev/null | grep '^{' | tail -1 | python -c "$P" "slabs=$sl once split=$sp"; done
done
