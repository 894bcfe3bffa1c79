import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
pkg, orc = entry.load_package(), entry.load_oracle()
orc.build(); orc.set_num_threads(8)
b = pkg.model.packed_bed(1600, seed=4, cd_freq=7, spacing_mult=2.5, init_vz=-0.4, aspect=(2.0, 1.0, 0.5))
b.SetExpandSafetyAdder(0.5)
p, sc = b.Initialize()
nc = int(sc.nOwnerClumps)
for ns in (1, 2):
    m = pkg.abi.Multi(devices=(0,)); m.build(p, sc, slabs_per_device=ns, axis=0, halo=0.03, arith="exact")
    sim = orc.make_sim(pkg, p, sc)
    acc = np.zeros((nc, 3), np.float32); acc[:, 0] = np.linspace(-2.0, 2.0, nc, dtype=np.float32)
    m.add_owner_acc(0, acc); sim.add_owner_acc(0, acc)
    m.step(2); sim.step(2); m.sync()
    g, o = m.download_state(), sim.download_state()
    d = np.abs(g["vX"][:nc] - o["vX"][:nc])
    print(ns, "slabs: max |dvX|", d.max(), "at", d.argmax(), "multi", g["vX"][d.argmax()], "oracle", o["vX"][d.argmax()], "acc", acc[d.argmax(), 0], "n off", int((d > 1e-7).sum()))
    m.close()
