import os, sys
import numpy as np
os.environ["DEME_MIG_CHECK"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
pkg = entry.load_package()
b = pkg.model.packed_bed(20_000, seed=6, cd_freq=0, spacing_mult=2.5, init_vz=-0.2, aspect=(2.0, 1.0, 0.5))
p, sc = b.Initialize()
nc = int(sc.nOwnerClumps)
b.arrays["vX"][:nc] = 2.0
sc = pkg.abi.make_scene_struct(b.arrays, b.counts)
for dl in (False, True):
    m = pkg.abi.Multi(devices=(0,))
    m.build(p, sc, slabs_per_device=4, axis=0, halo=0.035, arith="exact", caller_order=True)
    m.set_migration(100)
    try:
        for k in range(12):
            m.step(100); m.sync()
            if dl:
                m.download_state()
        print("download", dl, ": fine")
    except Exception as e:
        print("download", dl, "after", 100 * (k + 1), ":", str(e)[-400:])
    m.close()
