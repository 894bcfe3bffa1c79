import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
pkg, orc = entry.load_package(), entry.load_oracle()
orc.build(); orc.set_num_threads(8)
b = pkg.model.packed_bed(1600, seed=4, cd_freq=7, spacing_mult=2.5, init_vz=-0.4, aspect=(2.0, 1.0, 0.5))
b.SetExpandSafetyAdder(0.5)
p, sc = b.Initialize()
m = pkg.abi.Multi(devices=(0,)); m.build(p, sc, slabs_per_device=2, axis=0, halo=0.03, arith="exact")
sim = orc.make_sim(pkg, p, sc)
m.step(60); sim.step(60); m.sync()
ga, gb, gt = m.contacts(); oa, ob, ot, _ = sim.contacts()
print("lists equal", np.array_equal(ga, oa) and np.array_equal(gb, ob))
so = np.asarray(b.arrays["ownerClumpBody"])
for w in range(3):
    gw, ow = m.wildcard(w), sim.wildcard(w)
    d = np.abs(gw - ow); s = np.abs(gw + ow)
    bad = np.nonzero(d > 1e-7)[0]
    print("w", w, "rows off", len(bad), "of", len(d), "; of those with gw ~ -ow:", int((s[bad] < 1e-7).sum()))
    for i in bad[:8]:
        print("   row", i, "A", ga[i], "B", gb[i], "type", gt[i], "ownerA", so[ga[i]], "ownerB", so[gb[i]] if gt[i] == 1 else -1, "multi", gw[i], "oracle", ow[i])
