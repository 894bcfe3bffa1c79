import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
pkg = entry.load_package()
b = pkg.model.packed_bed(20_000, seed=6, cd_freq=0, spacing_mult=2.5, init_vz=-0.2, aspect=(2.0, 1.0, 0.5))
p, sc = b.Initialize()
nc = int(sc.nOwnerClumps)
b.arrays["vX"][:nc] = 2.0
sc = pkg.abi.make_scene_struct(b.arrays, b.counts)
m = pkg.abi.Multi(devices=(0,))
m.build(p, sc, slabs_per_device=4, axis=0, halo=0.035, arith="exact")
m.set_migration(100)
for k in range(15):
    m.step(100); m.sync()
    try:
        st = m.download_state()
        print("after", 100 * (k + 1), "steps: own", [m.slab_counts(s)[0][0] for s in range(4)], "ranges", [tuple(round(v, 4) for v in m.slab_counts(s)[1]) for s in range(4)], "migrated", m.counts()[1])
    except Exception as e:
        print("after", 100 * (k + 1), "steps: download failed:", e); break
try:
    moved, edges = m.rebalance(); print("rebalance moved", moved, "edges", np.round(edges, 4))
    print("own", [m.slab_counts(s)[0] for s in range(4)])
    st = m.download_state(); print("download ok")
    m.step(50); m.sync(); st = m.download_state(); print("download after 50 more ok")
except Exception as e:
    print("failed:", e)
