"""The engine's own numbering, host part (csrc/deme_order.inc: recursive bisection into tiles of 128 clumps): runs without a GPU
through deme_order_probe.  The order must be a permutation, a function of the scene alone, and compact whatever order the
caller's was; an order that is compact already is recognised as such (the engine then keeps the caller's numbering)."""
import ctypes as C

import numpy as np


def _probe(pkg, p, arrays, n):
    lib = C.CDLL(pkg.library_path())
    vox = np.ascontiguousarray(arrays["voxelID"][:n], np.uint64)
    lx, ly, lz = (np.ascontiguousarray(arrays[k][:n], np.uint16) for k in ("locX", "locY", "locZ"))
    order, sp = np.zeros(n, np.uint32), (C.c_double * 2)()
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    assert lib.deme_order_probe(C.byref(p), C.c_size_t(n), vp(vox), vp(lx), vp(ly), vp(lz), vp(order), sp) == 0
    return order, float(sp[0]), float(sp[1])


def test_engine_order_is_a_compact_permutation_for_every_input_order(pkg):
    spreads = {}
    for order in ("lattice", "morton", "random"):
        b = pkg.model.packed_bed(30000, seed=5, cd_freq=0, spacing_mult=3.0, aspect=(1.0, 1.0, 0.1), order=order)
        p, sc = b.Initialize()
        n = int(sc.nOwnerClumps)
        perm, given, best = _probe(pkg, p, b.arrays, n)
        assert np.array_equal(np.sort(perm), np.arange(n))
        again, _, best2 = _probe(pkg, p, b.arrays, n)
        assert np.array_equal(perm, again) and best == best2  # a function of the scene
        spreads[order] = (given, best)
        # every tile of the engine's order is a box-shaped cluster: its bounding box holds few more clumps than the 128 of the tile
        X = pkg.model.decode_positions(b.arrays["voxelID"][:n], b.arrays["locX"][:n], b.arrays["locY"][:n], b.arrays["locZ"][:n],
                                       p.nvXp2, p.nvYp2, p.voxelSize, p.l)[perm]
        worst = 0.0
        for t0 in range(0, n - 128, 128 * 37):  # a sample of the tiles
            lo, hi = X[t0:t0 + 128].min(0), X[t0:t0 + 128].max(0)
            inside = int(np.all((X >= lo) & (X <= hi), axis=1).sum())
            worst = max(worst, inside / 128.0)
        assert worst < 1.6, worst
    # the same bed, the same compactness, whatever the caller's order was; a random order is far from it, a row-major one clearly
    bests = [v[1] for v in spreads.values()]
    assert max(bests) < 1.05 * min(bests), spreads
    assert spreads["random"][0] > 10 * spreads["random"][1] and spreads["lattice"][0] > 1.5 * spreads["lattice"][1], spreads


def test_an_order_that_is_compact_already_is_recognised(pkg):
    b = pkg.model.packed_bed(20000, seed=6, cd_freq=0, spacing_mult=3.0, aspect=(1.0, 1.0, 0.2), order="random")
    p, sc = b.Initialize()
    n = int(sc.nOwnerClumps)
    perm, given, best = _probe(pkg, p, b.arrays, n)
    arr = {k: np.asarray(b.arrays[k])[:n][perm] for k in ("voxelID", "locX", "locY", "locZ")}
    perm2, given2, best2 = _probe(pkg, p, arr, n)
    assert given2 <= 1.15 * best2 and abs(given2 - best) < 1e-9 * best  # (what deme_upload_scene tests before it reorders)
