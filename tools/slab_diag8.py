import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
pkg = entry.load_package()
b = pkg.model.packed_bed(20_000, seed=6, cd_freq=0, spacing_mult=2.5, init_vz=-0.2, aspect=(2.0, 1.0, 0.5))
p, sc = b.Initialize()
nc = int(sc.nOwnerClumps)
b.arrays["vX"][:nc] = 2.0
x = pkg.model.decode_positions(b.arrays["voxelID"], b.arrays["locX"], b.arrays["locY"], b.arrays["locZ"], p.nvXp2, p.nvYp2, p.voxelSize, p.l)[:, 0] + float(p.LBFX)
for mode in ("download", "both"):
    parts = pkg.decomp.decompose(b.arrays, b.counts, x, 4, halo=0.035)
    ctxs = []
    for pt in parts:
        c = pkg.Context(0); c.set_arith_mode("exact"); c.set_params(p); c.upload_scene(pt["scene"]); ctxs.append(c)
    g = pkg.abi.HaloGroup(rank=0, world=1, device=0)
    for i, (c, pt) in enumerate(zip(ctxs, parts)):
        g.attach(c, pt, left=ctxs[i - 1] if i > 0 else None, right=ctxs[i + 1] if i + 1 < len(ctxs) else None)
    for c, pt in zip(ctxs, parts):
        g.set_slab(c, pt, 0.035)
    for k in range(12):
        g.step(100)
        if mode == "download":
            g.sync()
        g.migrate()
        if mode in ("download", "both"):
            for c in ctxs:
                c.download_state()
        bad = []
        for i, c in enumerate(ctxs):
            og = g.slab_ids(c)[0]
            if (og > nc).any():
                bad.append((i, int(np.nonzero(og > nc)[0][0]), int(og[og > nc][0]), g.slab_counts(c)))
        if bad:
            print(mode, ": garbage ids after", 100 * (k + 1), "steps:", bad); break
    else:
        print(mode, ": 12 migrations fine")
    g.close() if hasattr(g, "close") else None
for every in (300, 600):
    m = pkg.abi.Multi(devices=(0,))
    m.build(p, sc, slabs_per_device=4, axis=0, halo=0.035, arith="exact", caller_order=True)
    m.set_migration(100)
    for k in range(1200 // every):
        m.step(every); m.sync()
        try:
            m.download_state()
        except Exception as e:
            print("multi, download every", every, ": fails after", every * (k + 1), "steps:", str(e)[-130:]); break
    else:
        print("multi, download every", every, ": fine")
    m.close()
