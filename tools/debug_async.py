import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("dem-engine_amd")
K, D, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
b = pkg.model.packed_bed(n, seed=12, cd_freq=K, spacing_mult=3.0, init_vz=-1.0, aspect=(1.0, 1.0, 0.5))
b.SetExpandSafetyAdder(float(sys.argv[5]))
p, sc = b.Initialize()
lock, asyn = pkg.Context(0), pkg.Context(0)
for c in (lock, asyn):
    c.set_arith_mode("exact"); c.set_params(p); c.upload_scene(sc)
def pairs(c):
    a, bb, t, _ = c.contacts()
    return dict(zip(zip(a.tolist(), bb.tolist(), t.tolist()), c.wildcard(3).tolist()))
S0 = int(sys.argv[4])
lock.step(S0), asyn.step(S0)
asyn.set_async_detection(D)
for chunk in range(60):
    lock.step(K), asyn.step(K)
    sl, sa = lock.download_state(), asyn.download_state()
    bad = [k for k in ("voxelID", "locX", "locZ", "vX", "vZ", "omgBarX") if not np.array_equal(sl[k], sa[k])]
    pl, pa = pairs(lock), pairs(asyn)
    tl, ta = {q for q in pl if pl[q] > 0}, {q for q in pa if pa[q] > 0}
    if chunk % 10 == 0 or bad:
        print(chunk, "steps", S0 + K * (chunk + 1), "diff", bad, "n", len(pl), len(pa), "touching", len(tl), len(ta), "vmax", float(np.abs(sl["vZ"]).max()))
    if bad:
        break
