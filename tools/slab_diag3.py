"""diagnostic: the demo_bed scene (tests/test_host_shell.py: _bed_scene) in 5 slabs through deme_multi against the oracle, by step count"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
pkg, orc = entry.load_package(), entry.load_oracle()
orc.build(); orc.set_num_threads(8)
from tests.test_host_shell import _bed_inputs, _bed_scene
n = 1100
xyz, q, kind = _bed_inputs(n)
ns = int(sys.argv[1]) if len(sys.argv) > 1 else 5
b = _bed_scene(pkg, xyz, q, kind, None)
p, sc = b.Initialize()
sim = orc.make_sim(pkg, p, sc)
c = np.zeros((15, 4), np.float32); c[0] = (0.01, 0, 0, 0)
sim.set_prescription(10, has=0b111, flags=0b111, coef=c)
m = pkg.abi.Multi(devices=(0,)); m.build(p, sc, slabs_per_device=ns, axis=-1, halo=0.0, arith="exact")
for s in range(ns):
    cx = m.slab_ctx(s)
    cx.compile_prescriptions(*b.prescription_cases())
def pos(st):
    return pkg.model.decode_positions(st["voxelID"], st["locX"], st["locY"], st["locZ"], p.nvXp2, p.nvYp2, p.voxelSize, p.l)
done = 0
for upto in (500, 1000, 2000, 3000, 4000, 5000, 6000, 7000):
    m.step(upto - done); sim.step(upto - done); m.sync(); done = upto
    g, o = m.download_state(), sim.download_state()
    d = np.abs(pos(g) - pos(o)).max(1)
    off = np.nonzero(d > 1e-7)[0]
    cnt, _ = m.counts()
    print(f"step {upto}: |dx| {d.max():.3e}, {len(off)} owners off {off[:10].tolist()} (owners {len(d)}, clumps {n}); contacts multi {int(cnt.nContacts)} oracle {int(sim.counts().nContacts)}; plate x multi {pos(g)[n][0]:.6f} oracle {pos(o)[n][0]:.6f}")
