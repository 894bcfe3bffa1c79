"""host time of the library's slab loop per step (deme_halo_group_host_time) for ONE slab of the headline bed: what a rank of a multi-GPU
run spends enqueuing, beside the GPU time of the step"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
pkg = entry.load_package()
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
b = bench.build_bed(pkg, n, 2024, 40, order="morton")
p, sc = b.Initialize()
x = pkg.model.decode_positions(b.arrays["voxelID"], b.arrays["locX"], b.arrays["locY"], b.arrays["locZ"], p.nvXp2, p.nvYp2, p.voxelSize, p.l)[:, 0] + float(p.LBFX)
for ns in (1, 2):
    parts = pkg.decomp.decompose(b.arrays, b.counts, x, ns, halo=0.03)
    ctxs = []
    for pt in parts:
        c = pkg.Context(0); c.set_arith_mode("fast"); c.set_params(p); c.upload_scene(pt["scene"]); ctxs.append(c)
    g = pkg.abi.HaloGroup(rank=0, world=1, device=0)
    for i, (c, pt) in enumerate(zip(ctxs, parts)):
        g.attach(c, pt, left=ctxs[i - 1] if i > 0 else None, right=ctxs[i + 1] if i + 1 < len(ctxs) else None)
    g.step(3000); g.sync()
    g.host_time(reset=True)
    t0 = time.perf_counter(); g.step(400); t1 = time.perf_counter(); g.sync(); t2 = time.perf_counter()
    us = g.host_time()
    print(f"{ns} slab(s) of {n // ns} clumps: wall per step {1e6 * (t2 - t0) / 400:.1f} us; host returned after {1e6 * (t1 - t0) / 400:.1f} us per step; "
          f"library host timers per step (interior / pack / RCCL / tail): {[round(u / 400, 1) for u in us]}")
    g.close(); [c.close() for c in ctxs]
