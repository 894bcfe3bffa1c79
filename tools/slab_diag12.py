import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import __graft_entry__ as entry
pkg = entry.load_package()
from tests.test_decomp import GKEYS
b = pkg.model.packed_bed(200_000, seed=3)
p, sc = b.Initialize()
nc = int(sc.nOwnerClumps)
c0 = pkg.Context(0); c0.set_arith_mode("fast"); c0.set_params(p); c0.upload_scene(sc); c0.step(3000)   # settle a little
st = c0.download_state()
arr = dict(b.arrays)
for k in GKEYS: arr[k] = np.asarray(st[k]).copy()
for k in ("vX", "vY"):
    arr[k] = arr[k].copy(); arr[k][:nc] += 3.0
sc2 = pkg.abi.make_scene_struct(arr, b.counts)
def pos(s): return pkg.model.decode_positions(s["voxelID"], s["locX"], s["locY"], s["locZ"], p.nvXp2, p.nvYp2, p.voxelSize, p.l)[:nc]
runs = {}
for tag, mig in (("no migration", 0), ("migration every 50", 50), ("migration every 50 + rebalance", -50)):
    m = pkg.abi.Multi(devices=(0,)); m.build(p, sc2, slabs_per_device=4, axis=-1, halo=0.03, arith="fast")
    m.set_migration(abs(mig))
    if mig < 0: m.set_rebalance(2)
    runs[tag] = m
one = pkg.Context(0); one.set_arith_mode("fast"); one.set_params(p); one.upload_scene(sc2)
two = pkg.Context(0); two.set_arith_mode("fast"); two.set_reorder(False) if hasattr(two, "set_reorder") else None; two.set_params(p); two.upload_scene(sc2)
for k in range(6):
    one.step(50); two.step(50)
    x1 = pos(one.download_state()); x2 = pos(two.download_state())
    line = f"step {50*(k+1)}: one ctx vs one ctx in the caller's order {np.abs(x1-x2).max():.2e}"
    for tag, m in runs.items():
        m.step(50); m.sync()
        line += f" | {tag}: {np.abs(pos(m.download_state())-x1).max():.2e} (moved {m.counts()[1]})"
    print(line)
