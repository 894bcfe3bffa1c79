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
for snap, so in ((True, False), (True, True)):
    plan, parts = pkg.decomp.decompose_lib(p, sc, 4, 0.035, axis=0, snap=snap, spatial_order=so)
    ctxs = []
    for pt in parts:
        c = pkg.Context(0); c.set_arith_mode("exact"); c.set_params(p); c.upload_scene(pt["scene"]); ctxs.append(c)
    g = pkg.abi.HaloGroup(rank=0, world=1, device=0)
    for i, (c, pt) in enumerate(zip(ctxs, parts)):
        g.attach(c, pt, left=ctxs[i - 1] if i > 0 else None, right=ctxs[i + 1] if i + 1 < len(ctxs) else None)
    for c, pt in zip(ctxs, parts):
        g.set_slab(c, pt, 0.035)
    print("edges", plan.edges)
    for k in range(12):
        g.step(100)
        g.migrate()
        bad = []
        for i, c in enumerate(ctxs):
            og = g.slab_ids(c)[0]
            if (og > nc).any():
                bad.append((i, int(np.nonzero(og > nc)[0][0]), int(og[og > nc][0]), g.slab_counts(c)))
        if bad:
            print("snap", snap, "spatial", so, ": garbage ids after", 100 * (k + 1), "steps:", bad); break
    else:
        print("snap", snap, "spatial", so, ": 12 migrations fine")
