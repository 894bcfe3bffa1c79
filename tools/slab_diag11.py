import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
pkg = entry.load_package()
import importlib
sys.path.insert(0, "tests")
from test_multi import _positions
b = pkg.model.packed_bed(20_000, seed=6, cd_freq=0, spacing_mult=2.5, init_vz=-0.2, aspect=(2.0, 1.0, 0.5))
p, sc = b.Initialize()
nc = int(sc.nOwnerClumps)
b.arrays["vX"][:nc] = 2.0
sc = pkg.abi.make_scene_struct(b.arrays, b.counts)
orc = entry.load_oracle(); orc.build()
print("oracle module:", orc)
m = pkg.abi.Multi(devices=(0,)); m.build(p, sc, slabs_per_device=4, axis=0, halo=0.035, arith="exact"); m.set_migration(100)
one = pkg.abi.Multi(devices=(0,)); one.build(p, sc, slabs_per_device=1, axis=0, halo=0.035, arith="exact")
sim = orc.make_sim(pkg, p, sc) if orc else None
if orc: orc.set_num_threads(16)
for k in range(15):
    m.step(100); one.step(100); m.sync(); one.sync()
    xm = _positions(pkg, p, m.download_state(), nc); x1 = _positions(pkg, p, one.download_state(), nc)
    line = f"step {100*(k+1)}: slabs vs one slab |dx| {np.abs(xm-x1).max():.3e}"
    if sim:
        sim.step(100); xo = _positions(pkg, p, sim.download_state(), nc)
        line += f"; slabs vs oracle {np.abs(xm-xo).max():.3e}; one slab vs oracle {np.abs(x1-xo).max():.3e}"
    print(line, "contacts", m.counts()[0].nContacts if hasattr(m.counts()[0], "nContacts") else "")
