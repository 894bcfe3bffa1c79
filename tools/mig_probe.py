import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("DEME_ARITH", "exact")
import __graft_entry__ as entry
from tests.test_decomp import GKEYS, _sheared_bed, _state_x, gather_positions
from tests.test_config2_slabs import _make, _group
pkg = entry.load_package(); orc = entry.load_oracle(); orc.build(); orc.set_num_threads(32)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
b, p, sc, x = _sheared_bed(pkg, n, 6)
nc = int(sc.nOwnerClumps); halo = 0.035
sim = orc.make_sim(pkg, p, sc)
sim.step(201)
so = sim.download_state()
Xo = pkg.model.decode_positions(so["voxelID"], so["locX"], so["locY"], so["locZ"], p.nvXp2, p.nvYp2, p.voxelSize, p.l)[:nc]
def err(parts, ctxs):
    X, V = gather_positions(pkg, parts, ctxs, p, nc)
    return np.abs(X - Xo).max()
for mode in ("none", "numpy", "library"):
    parts = pkg.decomp.decompose(b.arrays, b.counts, x, 3, halo=halo)
    ctxs = [_make(pkg, p, pt["scene"]) for pt in parts]
    g = _group(pkg, ctxs, parts)
    if mode == "library":
        for c, pt in zip(ctxs, parts): g.set_slab(c, pt, halo)
    moved = 0
    for it in range(4):
        g.step(50); g.sync()
        if mode == "numpy":
            states = [c.download_state() for c in ctxs]; cnts = [c.contacts() for c in ctxs]
            Ws = [np.stack([c.wildcard(w) for w in range(4)], 1) for c in ctxs]
            parts2, seeds = pkg.decomp.migrate_neighbours_in_process(parts, states, cnts, Ws, parts[0]["all_edges"], halo, _state_x(pkg, p))
            moved += sum(len(np.setdiff1d(a["global_ids"], b_["global_ids"])) for a, b_ in zip(parts2, parts))
            g.close()
            parts = parts2
            ctxs = [_make(pkg, p, pt["scene"]) for pt in parts]
            for c, sd in zip(ctxs, seeds): c.seed_contacts(*sd)
            g = _group(pkg, ctxs, parts)
        elif mode == "library":
            moved += g.migrate()
    g.step(1); g.sync()
    if mode == "library":
        Xs = np.zeros((nc, 3))
        for c in ctxs:
            n_own = g.slab_counts(c)[0]; og = g.slab_ids(c)[0]; st = c.download_state()
            X = pkg.model.decode_positions(st["voxelID"], st["locX"], st["locY"], st["locZ"], p.nvXp2, p.nvYp2, p.voxelSize, p.l)
            Xs[og[:n_own].astype(np.int64)] = X[:n_own]
        e = np.abs(Xs - Xo).max()
    else:
        e = err(parts, ctxs)
    print(mode, "moved", moved, "max |dx| vs oracle", e, flush=True)
    g.close()
