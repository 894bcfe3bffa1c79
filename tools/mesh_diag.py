import os, sys, copy
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
pkg = entry.load_package()
import bench
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300_000
b = bench.build_bed(pkg, N, 2024, 40, order="morton")
lo, hi = b.user_box_min, b.user_box_max
v, f = pkg.model.plate_mesh(122, 122, float(hi[0] - lo[0]) * 0.98, float(hi[1] - lo[1]) * 0.98, z=0.0, wavy=0.002)
m = b.AddMeshObject(v, f, 0)
m.SetInitPos(((lo[0] + hi[0]) / 2, (lo[1] + hi[1]) / 2, 0.021))
p, sc = b.Initialize()
keys = ("voxelID", "locX", "locY", "locZ", "oriQw", "oriQx", "oriQy", "oriQz", "vX", "vY", "vZ", "omgBarX", "omgBarY", "omgBarZ")
fast = pkg.Context(0); fast.set_arith_mode("fast"); fast.set_params(p); fast.upload_scene(sc); fast.step(12000)
st = fast.download_state()
exact = pkg.Context(0); exact.set_arith_mode("exact"); exact.set_params(p); exact.upload_scene(sc)
exact.upload_state({k: st[k] for k in keys}); fast.upload_state({k: st[k] for k in keys})
for c in (fast, exact):
    c.compute_margins(0), c.detect(), c.migrate(), c.calc_forces()
print(fast.force_kernel(), exact.force_kernel())
g, o = fast.download_state(), exact.download_state()
n = int(sc.nOwnerClumps)
G = np.stack([g[k][:n] for k in ("aX", "aY", "aZ")], 1).astype(np.float64); O = np.stack([o[k][:n] for k in ("aX", "aY", "aZ")], 1).astype(np.float64)
d = np.abs(G - O).max(1)
print("max |O|", np.abs(O).max(), "max diff", d.max(), "owners off by > 1e-4 of max:", int((d > 1e-4 * np.abs(O).max()).sum()))
a, bb, t, _ = fast.contacts()
so = np.asarray(b.arrays["ownerClumpBody"])
sm_owner = np.zeros(n, bool); sm_owner[so[a[t == 2]]] = True
bad = np.argsort(-d)[:10]
for i in bad:
    print("owner", i, "diff", d[i], "fast", G[i], "exact", O[i], "has sphere-mesh contact", bool(sm_owner[i]))
print("of the owners off, with a sphere-mesh contact:", int(sm_owner[d > 1e-4 * np.abs(O).max()].sum()))
