"""The engine's own numbering (csrc/deme_order.inc): a scene handed over in an order that is not spatial -- the reference's ids are
load order, DEM/dT.cpp:700-800 -- is kept along a Z-order curve inside, and every id at the C-ABI stays the caller's.

What must hold, against the oracle (which works in the caller's numbering throughout):
  * the owner-tile force pass evaluates the list (that is what the order is for);
  * contact lists, bin incidences and history maps BIT-IDENTICAL in the caller's ids and canonical order (which sphere of a pair
    is A follows the caller's ids inside as well, so even the contact-point-in-bin decisions are the oracle's);
  * per-contact history, owner accelerations, trajectories within the fast mode's stated bounds (tests/test_fast_mode.py);
  * every per-owner / per-sphere / per-contact array crossing the boundary lands on the caller's index: state round trips,
    margins, wildcards, seeds, persistent marks, sphere geometry, added accelerations, inspection values.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KEYS = ("voxelID", "locX", "locY", "locZ", "oriQw", "oriQx", "oriQy", "oriQz", "vX", "vY", "vZ", "omgBarX", "omgBarY", "omgBarZ")


def _positions(pkg, p, st):
    return pkg.model.decode_positions(st["voxelID"], st["locX"], st["locY"], st["locZ"], p.nvXp2, p.nvYp2, p.voxelSize, p.l)


def _bed(pkg, n=6000, order="random", **kw):
    return pkg.model.packed_bed(n, seed=5, cd_freq=0, spacing_mult=3.0, init_vz=-1.0, aspect=(1.0, 1.0, 0.25), order=order, **kw)


def _settled(pkg, b, steps=9000):
    p, sc = b.Initialize()
    ctx = pkg.Context(0)
    ctx.set_arith_mode("exact")  # the caller's order is kept: this context is the plain reference for the state
    ctx.set_params(p), ctx.upload_scene(sc)
    assert not ctx.engine_order()[0]
    ctx.step(steps)
    st = ctx.download_state()
    ctx.close()
    return p, sc, {k: st[k] for k in KEYS}


def test_random_order_scene_is_tiled_and_matches_the_oracle(pkg, orc):
    b = _bed(pkg)
    p, sc, st = _settled(pkg, b)
    ctx = pkg.Context(0)
    ctx.set_arith_mode("fast")
    ctx.set_params(p), ctx.upload_scene(sc), ctx.upload_state(st)
    reordered, given, best = ctx.engine_order()
    assert reordered and given > 3 * best, (reordered, given, best)
    sim = orc.make_sim(pkg, p, sc)
    sim.upload_state(st)
    # a state round trip lands on the caller's indices
    back = ctx.download_state()
    assert all(np.array_equal(back[k], st[k]) for k in KEYS)
    for s in (ctx, sim):
        s.compute_margins(0), s.detect(), s.migrate(), s.calc_forces()
    assert ctx.force_kernel()[0] == "k_tile_forces<0, false>", ctx.force_kernel()
    ga, oa = ctx.contacts(), sim.contacts()
    assert len(ga[0]) > 4000 and all(np.array_equal(x, y) for x, y in zip(ga[:3], oa[:3]))
    gi, oi = ctx.bin_incidence(), sim.bin_incidence()
    assert np.array_equal(gi[0], oi[0]) and np.array_equal(gi[1], oi[1])
    gx, ox = ctx.sphere_geometry(), sim.sphere_geometry()
    assert all(np.array_equal(x, y) for x, y in zip(gx, ox))
    g, o = ctx.download_state(), sim.download_state()
    n = int(sc.nOwnerClumps)
    for keys in (("aX", "aY", "aZ"), ("alphaX", "alphaY", "alphaZ")):
        G = np.stack([g[k][:n] for k in keys], 1).astype(np.float64)
        O = np.stack([o[k][:n] for k in keys], 1).astype(np.float64)
        assert np.abs(G - O).max() <= 2e-4 * np.abs(O).max(), keys
    for w in range(4):
        gw, ow = ctx.wildcard(w), sim.wildcard(w)
        assert np.abs(gw - ow).max() <= 1e-5 * max(np.abs(ow).max(), 1e-12) + 1e-12, w
    # 100 further steps (detection at every step: the history map is exercised with the engine's list order every time)
    ctx.step(100), sim.step(100)
    ga, oa = ctx.contacts(), sim.contacts()
    assert all(np.array_equal(x, y) for x, y in zip(ga[:3], oa[:3]))
    g, o = ctx.download_state(), sim.download_state()
    dx = np.abs(_positions(pkg, p, g) - _positions(pkg, p, o)).max()
    dv = max(np.abs(g[k] - o[k]).max() for k in ("vX", "vY", "vZ"))
    print(f"random-order bed kept along the curve, 100 steps vs the oracle: |dx| {dx:.3e} m, |dv| {dv:.3e} m/s")
    assert dx <= 5e-8 and dv <= 2e-4
    # the map of an idempotent detection is the identity in the caller's order too
    ctx.detect()
    a2 = ctx.contacts()
    ctx.migrate()
    ctx.detect()
    a3 = ctx.contacts()
    assert all(np.array_equal(x, y) for x, y in zip(a2[:3], a3[:3]))
    assert np.array_equal(a3[3], np.arange(len(a3[0]), dtype=np.uint32))
    ctx.close()


def test_boundary_arrays_keep_the_callers_indices(pkg, orc):
    b = _bed(pkg, n=4000)
    p, sc, st = _settled(pkg, b, steps=7000)
    ctx = pkg.Context(0)
    ctx.set_arith_mode("fast")
    ctx.set_params(p), ctx.upload_scene(sc), ctx.upload_state(st)
    assert ctx.engine_order()[0]
    twin = pkg.Context(0)  # the same scene kept in the caller's order (deme_set_reorder(0)): the reference for what an index means
    twin.set_arith_mode("fast")
    twin.set_reorder(False)
    twin.set_params(p), twin.upload_scene(sc), twin.upload_state(st)
    assert not twin.engine_order()[0]
    nO, nS = int(sc.nOwners), int(sc.nSpheres)
    rng = np.random.default_rng(3)
    # margins: per owner
    m = rng.uniform(1e-5, 3e-5, nO).astype(np.float32)
    for c in (ctx, twin):
        c.set_margins(m), c.detect(), c.migrate()
    ga, ta = ctx.contacts(), twin.contacts()
    assert len(ga[0]) > 1500 and all(np.array_equal(x, y) for x, y in zip(ga[:3], ta[:3]))
    # contact wildcards: written by list row, read back, and seeded lists
    W = rng.normal(size=(len(ga[0]), 4)).astype(np.float32) * 1e-7
    for w in range(4):
        ctx.set_wildcard(w, W[:, w])
    assert all(np.array_equal(ctx.wildcard(w), W[:, w]) for w in range(4))
    sel = rng.permutation(len(ga[0]))[: len(ga[0]) // 2]  # a seed need not be sorted
    ctx.seed_contacts(ga[0][sel], ga[1][sel], ga[2][sel], W[sel])
    sa = ctx.contacts()
    order = np.sort(sel)
    assert all(np.array_equal(x, y[order]) for x, y in zip(sa[:3], ga[:3]))
    assert all(np.array_equal(ctx.wildcard(w), W[order, w]) for w in range(4))
    # the seeded history is found again by the next detection, row for row
    ctx.detect(), ctx.migrate()
    full = ctx.contacts()
    assert all(np.array_equal(x, y) for x, y in zip(full[:3], ga[:3]))
    got = np.stack([ctx.wildcard(w) for w in range(4)], 1)
    expect = np.zeros_like(W)
    expect[order] = W[order]
    assert np.array_equal(got, expect)
    # persistent marks travel in the caller's ids
    ctx.mark_persistent_contacts(0)
    pa = ctx.persistent_contacts()
    assert all(np.array_equal(x, y) for x, y in zip(pa, ga[:3]))
    ctx.mark_persistent_contacts(0, mark=False)
    # added accelerations reach the owner the caller named (the twin gets the same history first: same forces up to rounding)
    for w in range(4):
        twin.set_wildcard(w, expect[:, w])
    for c in (ctx, twin):
        c.calc_forces()
    s0 = ctx.download_state()
    pick = np.array([7, 1234, 3999])
    for o in pick:
        ctx.add_owner_acc(int(o), acc=np.array([[3.0, -2.0, 1.0]], np.float32))
    ctx.integrate()
    s1 = ctx.download_state()
    twin.integrate()
    t1 = twin.download_state()
    dv = np.stack([s1[k] - t1[k] for k in ("vX", "vY", "vZ")], 1)[: int(sc.nOwnerClumps)]
    hit = np.abs(dv).max(1) > 1e-6
    assert np.array_equal(np.nonzero(hit)[0], pick), np.nonzero(hit)[0]
    assert np.allclose(dv[pick], np.array([3.0, -2.0, 1.0]) * p.h, rtol=1e-3, atol=3e-7)
    # per-owner inspection values
    iv, tv = ctx.inspect_values("absv", nO), twin.inspect_values("absv", nO)
    assert np.allclose(iv[np.setdiff1d(np.arange(int(sc.nOwnerClumps)), pick)], tv[np.setdiff1d(np.arange(int(sc.nOwnerClumps)), pick)], rtol=1e-5, atol=1e-7)
    ctx.close(), twin.close()


def _probe(pkg, p, arrays, n):
    """deme_order_probe: the engine's order of n clumps and the two tile-surface figures (host code, no context)"""
    import ctypes as C
    lib = C.CDLL(pkg.library_path())
    vox = np.ascontiguousarray(arrays["voxelID"][:n], np.uint64)
    lx, ly, lz = (np.ascontiguousarray(arrays[k][:n], np.uint16) for k in ("locX", "locY", "locZ"))
    order, sp = np.zeros(n, np.uint32), (C.c_double * 2)()
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    assert lib.deme_order_probe(C.byref(p), C.c_size_t(n), vp(vox), vp(lx), vp(ly), vp(lz), vp(order), sp) == 0
    return order, float(sp[0]), float(sp[1])


def test_compact_order_is_left_alone_and_the_switch_works(pkg):
    b = _bed(pkg, n=4000, order="random")
    p, sc = b.Initialize()
    n = int(sc.nOwnerClumps)
    order, given, best = _probe(pkg, p, b.arrays, n)
    assert np.array_equal(np.sort(order), np.arange(n)) and given > 3 * best
    # the same bed loaded in the engine's order: nothing to improve, no translation at the boundary
    (bt,) = b.batches
    bt.xyz, bt.vel, bt.angvel, bt.oriq, bt.family = bt.xyz[order], bt.vel[order], bt.angvel[order], bt.oriq[order], bt.family[order]
    bt.templates = [bt.templates[i] for i in order]
    p2, sc2 = b.Initialize()
    ctx = pkg.Context(0)
    ctx.set_arith_mode("fast")
    ctx.set_params(p2), ctx.upload_scene(sc2)
    reordered, given2, best2 = ctx.engine_order()
    assert not reordered and given2 <= 1.15 * best2 and abs(best2 - best) < 1e-6 * best, (reordered, given2, best2, best)
    ctx.close()
    b = _bed(pkg, n=4000, order="random")
    p, sc = b.Initialize()
    ctx = pkg.Context(0)
    ctx.set_arith_mode("exact")  # bit-identity with the oracle is the exact mode's contract: the caller's order stays
    ctx.set_params(p), ctx.upload_scene(sc)
    assert not ctx.engine_order()[0]
    ctx.close()


# ---- tiles that do not fit LDS: the per-tile fallback (k_tile_forces_big) ------------------------------------------------------------
def _fast_vs_oracle(pkg, orc, p, sc, st, steps, ctx):
    sim = orc.make_sim(pkg, p, sc)
    sim.upload_state(st)
    for s in (ctx, sim):
        s.compute_margins(0), s.detect(), s.migrate(), s.calc_forces()
    ga, oa = ctx.contacts(), sim.contacts()
    assert all(np.array_equal(x, y) for x, y in zip(ga[:3], oa[:3]))
    g, o = ctx.download_state(), sim.download_state()
    n = int(sc.nOwnerClumps)
    for keys in (("aX", "aY", "aZ"), ("alphaX", "alphaY", "alphaZ")):
        G = np.stack([g[k][:n] for k in keys], 1).astype(np.float64)
        O = np.stack([o[k][:n] for k in keys], 1).astype(np.float64)
        assert np.abs(G - O).max() <= 2e-4 * np.abs(O).max(), (keys, np.abs(G - O).max() / np.abs(O).max())
    for w in range(4):
        gw, ow = ctx.wildcard(w), sim.wildcard(w)
        assert np.abs(gw - ow).max() <= 1e-5 * max(np.abs(ow).max(), 1e-12) + 1e-12, w
    ctx.step(steps), sim.step(steps)
    ga, oa = ctx.contacts(), sim.contacts()
    assert all(np.array_equal(x, y) for x, y in zip(ga[:3], oa[:3]))
    g, o = ctx.download_state(), sim.download_state()
    dx = np.abs(_positions(pkg, p, g) - _positions(pkg, p, o)).max()
    dv = max(np.abs(g[k] - o[k]).max() for k in ("vX", "vY", "vZ"))
    return dx, dv


def test_tiles_that_do_not_fit_are_evaluated_one_by_one(pkg, orc):
    """A random numbering kept as it is (deme_set_reorder(0)): the 128 owners of a tile are scattered over the bed, every tile
    touches hundreds of foreign owners and none fits the LDS area.  The list stays with the tile structures -- tSum, records,
    the integrator's gather -- and k_tile_forces_big evaluates every tile; against the oracle with the fast mode's bounds."""
    b = _bed(pkg)
    b.SetFamilyExtraMargin(0, 0.004)  # (lists every clump's whole neighbourhood: a dozen partners per clump, all over the bed)
    p, sc, st = _settled(pkg, b)
    ctx = pkg.Context(0)
    ctx.set_arith_mode("fast")
    ctx.set_reorder(False)
    ctx.set_params(p), ctx.upload_scene(sc), ctx.upload_state(st)
    assert not ctx.engine_order()[0]
    dx, dv = _fast_vs_oracle(pkg, orc, p, sc, st, 100, ctx)
    tiles, big, halo, _ = ctx.tile_stats()
    assert ctx.force_kernel()[0] == "k_tile_forces<0, false>" and tiles > 40 and big > 0.5 * tiles, (ctx.force_kernel(), tiles, big)
    print(f"{big} of {tiles} tiles through the per-tile fallback, 100 steps vs the oracle: |dx| {dx:.3e} m, |dv| {dv:.3e} m/s")
    assert dx <= 5e-8 and dv <= 2e-4
    ctx.close()


def test_one_tile_with_a_big_sphere_does_not_fit_the_rest_stays_tiled(pkg, orc):
    """A 12 mm sphere, loaded FIRST (the smallest id of all: it is sphere A of every pair it is in), among 1 mm spheres, with a
    margin that lists its whole first shell: its tile has to stage ~300 foreign owners, more than the LDS area takes (192).
    That ONE tile goes through k_tile_forces_big, the others through k_tile_forces; lists and physics against the oracle."""
    import math
    r, R = 0.001, 0.012
    b = pkg.model.packed_bed(20000, seed=31, cd_freq=0, scale=r, spacing_mult=2.02, jitter=0.01, three_sphere=False,
                             aspect=(1.0, 1.0, 1.0), init_vz=0.0)
    batch = b.batches[0]
    c = (batch.xyz.min(0) + batch.xyz.max(0)) / 2
    keep = np.linalg.norm(batch.xyz - c, axis=1) > (R + 0.9 * r)
    for name in ("xyz", "vel", "angvel", "oriq", "family"):
        setattr(batch, name, getattr(batch, name)[keep])
    if isinstance(batch.templates, list):
        batch.templates = [t for t, k in zip(batch.templates, keep) if k]
    t = b.LoadSphereType(2.6e3 * 4 / 3 * math.pi * R ** 3, R, 0)
    big = b.AddClumps(t, [c.tolist()])
    big.SetVel(np.array([[0.05, 0.02, -0.08]], np.float32))
    b.batches = [b.batches[-1]] + b.batches[:-1]  # the big sphere is owner 0 / sphere 0
    b.SetExpandSafetyAdder(0.0)
    p, sc = b.Initialize()
    assert abs(float(b.arrays["Radii"][b.arrays["clumpComponentOffset"][0]]) - R) < 1e-9
    st0 = pkg.Context(0)
    st0.set_arith_mode("exact")
    st0.set_params(p), st0.upload_scene(sc)
    st = {k: v for k, v in st0.download_state().items() if k in KEYS}
    st0.close()
    ctx = pkg.Context(0)
    ctx.set_arith_mode("fast")
    ctx.set_params(p), ctx.upload_scene(sc)
    sim = orc.make_sim(pkg, p, sc)
    m = np.full(int(sc.nOwners), 0.6 * r, np.float32)  # (every pair within 1.2 r of touching is listed: the big sphere's first shell)
    for s in (ctx, sim):
        s.set_margins(m), s.detect(), s.migrate(), s.calc_forces()
    ga, oa = ctx.contacts(), sim.contacts()
    assert all(np.array_equal(x, y) for x, y in zip(ga[:3], oa[:3]))
    assert int(((ga[0] == 0) & (ga[2] == 1)).sum()) > 200, int((ga[0] == 0).sum())  # the shell around the big sphere
    tiles, nbig, halo, _ = ctx.tile_stats()
    assert ctx.force_kernel()[0] == "k_tile_forces<1, false>" or ctx.force_kernel()[0] == "k_tile_forces<0, false>"
    assert 1 <= nbig <= 20 and tiles > 100, (tiles, nbig, halo)
    g, o = ctx.download_state(), sim.download_state()
    n = int(sc.nOwnerClumps)
    for keys in (("aX", "aY", "aZ"), ("alphaX", "alphaY", "alphaZ")):
        G = np.stack([g[k][:n] for k in keys], 1).astype(np.float64)
        O = np.stack([o[k][:n] for k in keys], 1).astype(np.float64)
        scale = max(np.abs(O).max(), 1e-30)
        assert np.abs(G - O).max() <= 2e-4 * scale, keys
    for s in (ctx, sim):
        for _ in range(30):  # the K-step policy by hand: the same margins, a detection every step
            s.set_margins(m), s.detect(), s.migrate(), s.calc_forces(), s.integrate()
    g, o = ctx.download_state(), sim.download_state()
    dx = np.abs(_positions(pkg, p, g) - _positions(pkg, p, o)).max()
    dv = max(np.abs(g[k] - o[k]).max() for k in ("vX", "vY", "vZ"))
    print(f"a big sphere's tile through the per-tile fallback ({nbig} of {tiles} tiles), 30 steps vs the oracle: |dx| {dx:.3e} m, |dv| {dv:.3e} m/s")
    assert dx <= 5e-8 and dv <= 2e-4
    ctx.close()


def test_order_is_renewed_in_a_running_simulation(pkg, orc):
    """The clumps of a settled bed swap places (their states are permuted among the caller's ids: the same bed, but no id is where
    the engine's tiles expect it -- what a mixer does to a numbering over time).  The tiles of the old order stage owners from all
    over the bed and most stop fitting; deme_renew_order recomputes the order from the current positions and carries everything
    kept in engine slots across -- records, the contact list with its history, re-keyed and re-sorted -- while the caller's ids
    stay what they were.  Against the oracle before and after: lists bit-identical, history and states within the fast mode's
    bounds, and the tiles fit again."""
    b = _bed(pkg)
    p, sc, st = _settled(pkg, b)
    n = int(sc.nOwnerClumps)
    rng = np.random.default_rng(17)
    perm = rng.permutation(n)
    st2 = {k: v.copy() for k, v in st.items()}
    for k in KEYS:
        st2[k][:n] = st[k][:n][perm]
    ctx = pkg.Context(0)
    ctx.set_arith_mode("fast")
    ctx.set_params(p), ctx.upload_scene(sc), ctx.upload_state(st2)
    assert ctx.engine_order()[0]
    sim = orc.make_sim(pkg, p, sc)
    sim.upload_state(st2)
    ctx.step(5), sim.step(5)
    ga, oa = ctx.contacts(), sim.contacts()
    assert len(ga[0]) > 4000 and all(np.array_equal(x, y) for x, y in zip(ga[:3], oa[:3]))
    tiles, big0, halo0, _ = ctx.tile_stats()
    spread_before = ctx.engine_order()[1]
    ctx.renew_order()
    reordered, given, best = ctx.engine_order()
    assert reordered and given > 3 * best  # (what the renewal found: the slots it inherited were far from compact)
    # the list survives the renewal in the caller's ids and order, with its history
    gb = ctx.contacts()
    assert all(np.array_equal(x, y) for x, y in zip(gb[:3], ga[:3]))
    for w in range(4):
        gw, ow = ctx.wildcard(w), sim.wildcard(w)
        assert np.abs(gw - ow).max() <= 1e-5 * max(np.abs(ow).max(), 1e-12) + 1e-12, w
    back = ctx.download_state()
    osb = sim.download_state()
    assert np.abs(_positions(pkg, p, back) - _positions(pkg, p, osb)).max() <= 5e-8
    ctx.step(60), sim.step(60)
    g2, o2 = ctx.contacts(), sim.contacts()
    assert all(np.array_equal(x, y) for x, y in zip(g2[:3], o2[:3]))
    g, o = ctx.download_state(), sim.download_state()
    dx = np.abs(_positions(pkg, p, g) - _positions(pkg, p, o)).max()
    dv = max(np.abs(g[k] - o[k]).max() for k in ("vX", "vY", "vZ"))
    tiles, big1, halo1, _ = ctx.tile_stats()
    print(f"order renewed in a running bed: tiles through the fallback {big0} -> {big1} of {tiles}, largest halo {halo0} -> {halo1}; "
          f"65 steps vs the oracle: |dx| {dx:.3e} m, |dv| {dv:.3e} m/s")
    assert dx <= 5e-8 and dv <= 2e-4
    assert big1 == 0 and halo1 < halo0
    ctx.close()


def test_the_engine_renews_a_degraded_order_by_itself(pkg, orc):
    """The same swapped bed, left alone: the first detection after the upload takes the tiles' mean halo as the baseline -- that IS
    the degraded state here, so the bed is swapped AFTER a first detection in good order -- and once twenty detections have passed
    with the mean halo half as large again, the next detection starts with a renewal.  Results against the oracle throughout."""
    b = _bed(pkg)
    p, sc, st = _settled(pkg, b)
    n = int(sc.nOwnerClumps)
    ctx = pkg.Context(0)
    ctx.set_arith_mode("fast")
    ctx.set_params(p), ctx.upload_scene(sc), ctx.upload_state(st)
    sim = orc.make_sim(pkg, p, sc)
    sim.upload_state(st)
    ctx.step(2), sim.step(2)  # detections in the order of the upload: the baseline
    halo_good = ctx.tile_stats()[2]
    g = ctx.download_state()
    perm = np.random.default_rng(23).permutation(n)
    st2 = {k: g[k].copy() for k in KEYS}
    for k in KEYS:
        st2[k][:n] = g[k][:n][perm]
    ctx.upload_state(st2), sim.upload_state(st2)
    ctx.step(3), sim.step(3)
    assert ctx.order_renewals() == 0 and ctx.tile_stats()[2] > 2 * halo_good
    ctx.step(40), sim.step(40)  # > 20 detections (one per step) later
    assert ctx.order_renewals() == 1, ctx.order_renewals()
    assert ctx.tile_stats()[2] < 1.5 * halo_good and ctx.tile_stats()[1] == 0
    ga, oa = ctx.contacts(), sim.contacts()
    assert all(np.array_equal(x, y) for x, y in zip(ga[:3], oa[:3]))
    g, o = ctx.download_state(), sim.download_state()
    dx = np.abs(_positions(pkg, p, g) - _positions(pkg, p, o)).max()
    dv = max(np.abs(g[k] - o[k]).max() for k in ("vX", "vY", "vZ"))
    print(f"order renewed by the engine after the bed was swapped: largest halo back to {ctx.tile_stats()[2]} (was {halo_good}); |dx| {dx:.3e} m, |dv| {dv:.3e} m/s vs the oracle")
    assert dx <= 5e-8 and dv <= 2e-4
    ctx.close()


def test_exact_mode_on_a_reordered_context_goes_back_to_the_callers_order(pkg, orc):
    """deme_set_arith_mode(EXACT) on a context the engine had reordered: the exact mode's contract is bit-identity with the oracle,
    which sums an owner's contributions in the caller's order -- so the records, the sphere references and the current list (re-keyed,
    as the seed of the next detection) go back to the caller's slots, and the run that follows is the oracle's bit for bit.  The
    renewal entry points leave an exact-mode context alone."""
    b = _bed(pkg)
    p, sc, st = _settled(pkg, b)
    ctx = pkg.Context(0)
    ctx.set_arith_mode("fast")
    ctx.set_params(p), ctx.upload_scene(sc), ctx.upload_state(st)
    assert ctx.engine_order()[0]
    ctx.compute_margins(0), ctx.detect(), ctx.migrate()  # a list in the engine's slots (no force evaluation: the history stays zero)
    la = ctx.contacts()
    ctx.set_arith_mode("exact")
    assert not ctx.engine_order()[0]
    lb = ctx.contacts()  # the list survives in the caller's ids and canonical order
    assert all(np.array_equal(x, y) for x, y in zip(la[:3], lb[:3]))
    assert (lb[3] == 0xFFFFFFFF).all()  # ... as a seed: no previous list
    with pytest.raises(pkg.abi.DemeError):
        ctx.calc_forces()  # a seed is not evaluated: detect first
    with pytest.raises(pkg.abi.DemeError):
        ctx.renew_order()  # the exact mode keeps the caller's order
    back = ctx.download_state()
    assert all(np.array_equal(back[k], st[k]) for k in KEYS)
    sim = orc.make_sim(pkg, p, sc)
    sim.upload_state(st)
    ctx.step(25), sim.step(25)
    ga, oa = ctx.contacts(), sim.contacts()
    assert len(ga[0]) > 4000 and all(np.array_equal(x, y) for x, y in zip(ga[:3], oa[:3]))
    g, o = ctx.download_state(), sim.download_state()
    assert all(np.array_equal(g[k], o[k]) for k in KEYS), "exact mode after a restored order: states must be bit-identical to the oracle"
    assert ctx.order_renewals() == 0 and not ctx.engine_order()[0]
    # back in the fast mode the next lock-step detection reorders again
    ctx.set_arith_mode("fast")
    ctx.step(2)
    assert ctx.engine_order()[0] and ctx.force_kernel()[0] == "k_tile_forces<0, false>", (ctx.engine_order(), ctx.force_kernel())
    ctx.close()


def test_a_renewed_order_leaves_a_seed_not_a_list(pkg, orc):
    """deme_renew_order without a step behind it: the per-contact and per-sphere products of the last detection sit in the old slots,
    so the re-keyed list is offered as a SEED only -- forces are refused, the history map reads 'no previous contact', contact records
    and sphere geometry are refused -- until the next detection, after which everything agrees with the oracle again."""
    b = _bed(pkg)
    p, sc, st = _settled(pkg, b)
    n = int(sc.nOwnerClumps)
    perm = np.random.default_rng(29).permutation(n)
    st2 = {k: v.copy() for k, v in st.items()}
    for k in KEYS:
        st2[k][:n] = st[k][:n][perm]
    ctx = pkg.Context(0)
    ctx.set_arith_mode("fast")
    ctx.set_record_contacts(True)
    ctx.set_params(p), ctx.upload_scene(sc), ctx.upload_state(st2)
    sim = orc.make_sim(pkg, p, sc)
    sim.upload_state(st2)
    ctx.step(3), sim.step(3)
    la = ctx.contacts()
    ctx.contact_records(), ctx.sphere_geometry()  # (fine before the renewal)
    ctx.renew_order()
    assert ctx.order_renewals() == 1
    lb = ctx.contacts()
    assert all(np.array_equal(x, y) for x, y in zip(la[:3], lb[:3])) and (lb[3] == 0xFFFFFFFF).all()
    for refused in (ctx.calc_forces, ctx.contact_records, ctx.sphere_geometry):
        with pytest.raises(pkg.abi.DemeError):
            refused()
    # accelerations added through the caller's ids land on the right slots after the renewal (one upload of the touched range)
    acc = np.zeros((n, 3), np.float32)
    acc[:, 2] = np.linspace(0.0, 1.0, n, dtype=np.float32)
    ctx.add_owner_acc(0, acc)
    sim.add_owner_acc(0, acc)
    ctx.step(4), sim.step(4)
    ga, oa = ctx.contacts(), sim.contacts()
    assert all(np.array_equal(x, y) for x, y in zip(ga[:3], oa[:3]))
    fr, gx = ctx.contact_records(), ctx.sphere_geometry()
    ox = sim.sphere_geometry()
    assert np.abs(gx[0] - ox[0]).max() <= 5e-8 and np.array_equal(gx[3], ox[3])  # (fast-mode positions; radii land on the caller's spheres)
    g, o = ctx.download_state(), sim.download_state()
    dx = np.abs(_positions(pkg, p, g) - _positions(pkg, p, o)).max()
    dv = max(np.abs(g[k] - o[k]).max() for k in ("vX", "vY", "vZ"))
    assert dx <= 5e-8 and dv <= 2e-4, (dx, dv)
    ctx.close()


def test_spheres_must_be_clump_major(pkg):
    """the A / B roles of a pair follow the owner numbers, which is the reference's 'smaller sphere id first' only for clump-major
    spheres with ascending owners: a scene that breaks the contract is refused at upload"""
    b = pkg.model.packed_bed(400, seed=3, cd_freq=0)
    p, sc = b.Initialize()
    own = sc._keep["ownerClumpBody"]  # (the array the struct points at)
    own[[0, len(own) - 1]] = own[[len(own) - 1, 0]]  # the first sphere now belongs to the last clump
    ctx = pkg.Context(0)
    ctx.set_params(p)
    with pytest.raises(pkg.abi.DemeError, match="clump-major"):
        ctx.upload_scene(sc)
    ctx.close()
