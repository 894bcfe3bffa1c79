import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
pkg = entry.load_package()
from tests.test_decomp import GKEYS
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
b = pkg.model.packed_bed(n, seed=3)
p, sc = b.Initialize()
m = pkg.abi.Multi(devices=(0,))
t0 = time.time(); m.build(p, sc, slabs_per_device=4, axis=-1, halo=0.03, arith="fast"); t1 = time.time()
m.step(41); m.sync()
t2 = time.time()
st = m.download_state(); ga, gb, gt = m.contacts(); W = np.stack([m.wildcard(w) for w in range(4)], 1)
t3 = time.time()
m.reset(); t4 = time.time()
arr = dict(b.arrays)
for k in GKEYS: arr[k] = np.asarray(st[k]).copy()
sc2 = pkg.abi.make_scene_struct(arr, b.counts)
m.build(p, sc2, slabs_per_device=4, axis=-1, halo=0.03, arith="fast"); t5 = time.time()
m.seed_contacts(ga, gb, gt, W); t6 = time.time()
m.step(1); m.sync(); t7 = time.time()
print(f"{n} clumps, 4 slabs: first build {t1-t0:.2f} s; replan: gather state + {len(ga)} contacts {t3-t2:.2f} s, reset (groups reopened) {t4-t3:.2f} s, "
      f"plan + build {t5-t4:.2f} s, seed {t6-t5:.2f} s, first step (detection) {t7-t6:.3f} s; total {t7-t2:.2f} s")
