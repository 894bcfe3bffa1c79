#!/usr/bin/env python3
"""The persistent tile kernel against the one-workgroup-per-tile kernel, bit for bit: the same bed stepped N times (detections
included) in two child processes (DEME_TILE_PERSIST=0 / 1, read once per process), every owner-state array and contact wildcard
compared.   usage: persist_compare.py [clumps] [steps]"""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def child(out, n, steps):
    """steps the bench bed (bench.py --state-cache /tmp/bed.npz, 1e6 clumps, settled) when that file is there, else a lattice that starts overlapping"""
    import __graft_entry__ as entry
    import bench
    pkg = entry.load_package()
    cache = "/tmp/bed.npz"
    if os.path.exists(cache) and n >= 1000000:
        b = bench.build_bed(pkg, 1000000, 2024, 10, order="morton")
        p, sc = b.Initialize()
        ctx = pkg.Context(0)
        ctx.set_arith_mode("fast")
        ctx.set_params(p), ctx.upload_scene(sc)
        z = np.load(cache)
        ctx.upload_state({k: z[k] for k in z.files if k not in ("idA", "idB", "ctype", "wc", "presettle")})
        ctx.seed_contacts(z["idA"], z["idB"], z["ctype"], z["wc"] if z["wc"].size else None)
    else:
        b = pkg.model.packed_bed(n, seed=7, cd_freq=10, spacing_mult=2.4, init_vz=-0.5, order="morton")
        p, sc = b.Initialize()
        ctx = pkg.Context(0)
        ctx.set_arith_mode("fast")
        ctx.set_params(p), ctx.upload_scene(sc)
    ctx.step(steps)
    st = ctx.download_state()
    a, bb, t, _ = ctx.contacts()
    W = np.stack([ctx.wildcard(w) for w in range(4)], 1)
    np.savez(out, a=a, b=bb, t=t, W=W, kernel=np.array(ctx.force_kernel()[0]), **st)

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]))
        sys.exit(0)
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    outs = []
    for pe in (0, 1):
        out = f"/tmp/persist_cmp_{pe}.npz"
        env = dict(os.environ, DEME_TILE_PERSIST=str(pe))
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "--child", out, str(n), str(steps)], env=env)
        outs.append(np.load(out))
    A, B = outs
    bad = 0
    for k in A.files:
        if k == "kernel":
            continue
        same = A[k].shape == B[k].shape and np.array_equal(A[k], B[k])
        if not same:
            bad += 1
            d = np.abs(A[k].astype(np.float64) - B[k].astype(np.float64)).max() if A[k].shape == B[k].shape else float("nan")
            nd = int((A[k] != B[k]).sum()) if A[k].shape == B[k].shape else -1
            print(f"  {k}: DIFFERS (max |d| {d:.3e}, {nd} entries)")
    print(f"persist_compare: {n} clumps, {steps} steps, {len(A['a'])} contacts, kernel {A['kernel']}: " + ("bit-identical" if bad == 0 else f"{bad} arrays differ"))
    sys.exit(1 if bad else 0)
