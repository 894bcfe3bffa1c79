"""How well does a contact detection overlap with the stepping kernels on one GPU?  Two contexts holding the same settled bed:
A steps (force + integration, no detection), B runs detections; alone and concurrently (two host threads, two streams).
usage: python tools/overlap_probe.py [clumps]"""
import importlib
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("dem-engine_amd")
import bench  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
b = bench.build_bed(pkg, n, 2024, 100000, order="morton", bin_multiple=5.0)
p, sc = b.Initialize()
ctxs = []
for _ in range(2):
    c = pkg.Context(0)
    c.set_params(p), c.upload_scene(sc)
    ctxs.append(c)
A, B = ctxs
p.cdUpdateFreq = 40
A.set_params(p)
A.step(30000), A.sync()
print("contacts", int(A.counts().nContacts))
st = A.download_state()
B.upload_state({k: st[k] for k in st if not k.startswith(("a", "alpha"))})
p.cdUpdateFreq = 100000  # A: one detection at the start, then only force + integration
A.set_params(p)
A.step(1), A.sync()
B.compute_margins(40), B.detect(), B.sync()


def run_a(steps, out):
    t = time.perf_counter()
    A.step(steps), A.sync()
    out["a"] = (time.perf_counter() - t) / steps * 1e3


def run_b(reps, out):
    t = time.perf_counter()
    for _ in range(reps):
        B.compute_margins(40), B.detect()
    B.sync()
    out["b"] = (time.perf_counter() - t) / reps * 1e3


o = {}
run_a(400, o), run_b(20, o)
print(f"alone:      step {o['a']:.4f} ms, detection {o['b']:.3f} ms")
for steps, reps in ((800, 60), (800, 100)):
    o = {}
    ta, tb = threading.Thread(target=run_a, args=(steps, o)), threading.Thread(target=run_b, args=(reps, o))
    t0 = time.perf_counter()
    ta.start(), tb.start(), ta.join(), tb.join()
    wall = (time.perf_counter() - t0) * 1e3
    print(f"concurrent: step {o['a']:.4f} ms over {steps} steps, detection {o['b']:.3f} ms x {reps}; wall {wall:.1f} ms "
          f"(serial would be {steps * 0.0 + 0:.0f})")
