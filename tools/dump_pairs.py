#!/usr/bin/env python3
"""Profiling aid: settle the bench bed and dump (A owner, B owner) of every contact, for offline tile statistics."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import __graft_entry__ as entry  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
out = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/pairs.npz"
pkg = entry.load_package()
b = bench.build_bed(pkg, n, 2024, 40, order="morton")
p, sc = b.Initialize()
ctx = pkg.Context(0)
ctx.set_params(p)
ctx.upload_scene(sc)
done, last = 0, -1
t0 = time.time()
while done < 30000:
    ctx.step(1000)
    done += 1000
    nc = int(ctx.counts().nContacts)
    if done >= 4000 and last > 0 and abs(nc - last) < 0.002 * nc:
        break
    last = nc
print("settled after", done, "steps", nc, "contacts", time.time() - t0, "s", flush=True)
a, bb, t, _ = ctx.contacts()
own = np.asarray(b.arrays["ownerClumpBody"], np.uint32)
oa = own[a].astype(np.uint32)
ss = (t == 1)
ob = np.full(a.shape, 0xFFFFFFFF, np.uint32)
ob[ss] = own[bb[ss]]
np.savez_compressed(out, oa=oa, ob=ob, n_owners=int(sc.nOwners), n_clumps=int(sc.nOwnerClumps))
print("saved", out, oa.shape, flush=True)
