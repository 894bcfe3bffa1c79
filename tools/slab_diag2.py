"""diagnostic: thin slabs (thinner than two halos) through deme_multi and through decomp.decompose + HaloGroup, against the oracle"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
pkg, orc = entry.load_package(), entry.load_oracle()
orc.build(); orc.set_num_threads(8)
from tests.test_decomp import GKEYS, gather_positions
def bed():
    b = pkg.model.packed_bed(1600, seed=4, cd_freq=7, spacing_mult=2.5, init_vz=-0.4, aspect=(2.0, 1.0, 0.5))
    b.SetExpandSafetyAdder(0.5)
    p, sc = b.Initialize()
    return b, p, sc
b, p, sc = bed()
nc = int(sc.nOwnerClumps)
x = pkg.model.decode_positions(b.arrays["voxelID"], b.arrays["locX"], b.arrays["locY"], b.arrays["locZ"], p.nvXp2, p.nvYp2, p.voxelSize, p.l)[:, 0] + float(p.LBFX)
print("bed x range", x[:nc].min(), x[:nc].max())
sim = orc.make_sim(pkg, p, sc); sim.step(60); o = sim.download_state()
Xo = pkg.model.decode_positions(o["voxelID"], o["locX"], o["locY"], o["locZ"], p.nvXp2, p.nvYp2, p.voxelSize, p.l)[:nc]
halo = float(sys.argv[1]) if len(sys.argv) > 1 else 0.03
for ns in (3, 4, 5, 6, 8):
    try:
        m = pkg.abi.Multi(devices=(0,)); m.build(p, sc, slabs_per_device=ns, axis=0, halo=halo, arith="exact")
        m.step(60); m.sync(); g = m.download_state()
        Xg = pkg.model.decode_positions(g["voxelID"], g["locX"], g["locY"], g["locZ"], p.nvXp2, p.nvYp2, p.voxelSize, p.l)[:nc]
        print(f"multi   {ns} slabs halo {halo}: |dx| {np.abs(Xg - Xo).max():.3e}"); m.close()
    except Exception as e:
        print(f"multi   {ns} slabs: {e}")
    try:
        parts = pkg.decomp.decompose(b.arrays, b.counts, x, ns, halo=halo)
        ctxs = []
        for pt in parts:
            c = pkg.Context(0); c.set_arith_mode("exact"); c.set_params(p); c.upload_scene(pt["scene"]); ctxs.append(c)
        grp = pkg.abi.HaloGroup(rank=0, world=1, device=0)
        for i, (c, pt) in enumerate(zip(ctxs, parts)):
            grp.attach(c, pt, left=ctxs[i - 1] if i > 0 else None, right=ctxs[i + 1] if i + 1 < len(ctxs) else None)
        grp.step(60); grp.sync()
        X, V = gather_positions(pkg, parts, ctxs, p, nc)
        print(f"python  {ns} slabs halo {halo}: |dx| {np.abs(X - Xo).max():.3e}  widths {np.round(np.diff(parts[0]['all_edges'])[1:-1], 4).tolist()}")
        grp.close(); [c.close() for c in ctxs]
    except Exception as e:
        print(f"python  {ns} slabs: {e}")
