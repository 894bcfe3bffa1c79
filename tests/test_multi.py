"""The decomposition behind the C-ABI (csrc/deme_decomp.inc), on the GPU: deme_multi_* is what the C++ shell's DEMSolver(nGPUs /
device ids) opens (reference: DEM/API.h:52-56, DEM/APIPublic.cpp:22-110 pick the devices in the constructor).  One GPU is available
to the tests, so a deme_multi of ONE device is cut into several slabs: the same plan, contexts, exchange lists, migration books and
step loop as one slab per device -- the records travel by RCCL sends to self."""
import os

import numpy as np
import pytest

from tests.test_decomp import GKEYS, _sheared_bed, build_global

pytestmark = pytest.mark.gpu


def _bed(pkg, n=1600, seed=4, cd_freq=0):
    b = pkg.model.packed_bed(n, seed=seed, cd_freq=cd_freq, spacing_mult=2.5, init_vz=-0.4, aspect=(2.0, 1.0, 0.5))
    if cd_freq:
        b.SetExpandSafetyAdder(0.5)
    p, sc = b.Initialize()
    return b, p, sc


def _positions(pkg, p, st, n):
    return pkg.model.decode_positions(st["voxelID"], st["locX"], st["locY"], st["locZ"], p.nvXp2, p.nvYp2, p.voxelSize, p.l)[:n]


@pytest.mark.parametrize("n_slabs,axis", [(2, -1), (4, 0), (3, 1)])
def test_multi_slabs_equal_the_single_domain_oracle(pkg, orc, n_slabs, axis):
    """exact arithmetic: the owner states gathered by GLOBAL id (deme_multi_download_state) after 60 steps with a detection every
    7 are those of the oracle's single-domain run to fp32 summation order (a slab numbers its clumps its own way: 5e-9 m), the
    union of the slabs' contact lists in global ids is the oracle's list, and the y-cut (axis 1) works like the x-cut"""
    b, p, sc = _bed(pkg, cd_freq=7)
    nc = int(sc.nOwnerClumps)
    m = pkg.abi.Multi(devices=(0,))
    m.build(p, sc, slabs_per_device=n_slabs, axis=axis, halo=0.03, arith="exact")
    assert m.num_slabs() == n_slabs
    sim = orc.make_sim(pkg, p, sc)
    st0 = m.download_state()
    so0 = sim.download_state()
    assert all(np.array_equal(st0[k], so0[k]) for k in GKEYS)  # the gather by global id lands on the caller's rows
    m.step(60), sim.step(60)
    m.sync()
    g, o = m.download_state(), sim.download_state()
    dx = np.abs(_positions(pkg, p, g, nc) - _positions(pkg, p, o, nc)).max()
    dv = max(np.abs(g[k][:nc] - o[k][:nc]).max() for k in ("vX", "vY", "vZ"))
    assert dx < 5e-9 and dv < 1e-4, (dx, dv)  # (measured 1.3e-9 m / 1.6e-5 m/s: a slab numbers its clumps in the engine's order)
    cnt, moved = m.counts()
    assert int(cnt.nSteps) == 60 and int(cnt.nContacts) >= int(sim.counts().nContacts)  # (cross-cut contacts are on two lists)
    # the merged list in global sphere ids: every pair once, the oracle's list row for row; history in the same rows, B -> A vector
    # wildcards with the sign of the GLOBAL pair order wherever a slab held the pair the other way round
    ga, gb, gt = m.contacts()
    oa, ob, ot, _ = sim.contacts()
    assert len(ga) == len(oa) > 500 and np.array_equal(ga, oa) and np.array_equal(gb, ob) and np.array_equal(gt, ot)
    for w in range(4):
        gw, ow = m.wildcard(w), sim.wildcard(w)
        scale = max(float(np.abs(ow).max()), 1e-12)
        off = np.abs(gw - ow) > 1e-5 * scale + 1e-12
        # (a pair that begins to touch within rounding of a step boundary starts its history one step apart in the two runs --
        # tools/slab_diag4.py: one row of 2 255, 2.4 % = one step of forty -- everything else to fp32 rounding, signs included)
        assert off.sum() <= 3 and np.abs(gw - ow).max() <= 0.05 * scale, (w, int(off.sum()), float(np.abs(gw - ow).max()), scale)
    # unreduced inspector values by global id: what a single context answers for the same state
    one = pkg.Context(0)
    one.set_arith_mode("exact"), one.set_params(p), one.upload_scene(sc)
    one.upload_state({k: g[k] for k in GKEYS})
    n_o, n_s = int(sc.nOwners), int(sc.nSpheres)
    assert np.array_equal(m.inspect_values("absv", n_o)[:nc], one.inspect_values("absv", n_o)[:nc])
    assert np.array_equal(m.inspect_values("clump_max_z", n_s), one.inspect_values("clump_max_z", n_s))
    # a state uploaded by global id reaches own clumps, ghost copies and replicated owners alike
    m.upload_state({k: so0[k] for k in GKEYS})
    back = m.download_state()
    assert all(np.array_equal(back[k], so0[k]) for k in GKEYS)
    for s in range(n_slabs):
        c = m.slab_ctx(s)
        assert c.n_owners > 0 and c.counts().nSteps == 60
    m.close()


def test_multi_adds_accelerations_by_global_owner_id(pkg, orc):
    """deme_multi_add_owner_acc (a tracker's AddAcc on a decomposed run): a clump's entry reaches the slab that owns it, the next
    step takes it once -- two steps later the velocities are the oracle's"""
    b, p, sc = _bed(pkg, cd_freq=7)
    nc = int(sc.nOwnerClumps)
    m = pkg.abi.Multi(devices=(0,))
    m.build(p, sc, slabs_per_device=3, axis=0, halo=0.03, arith="exact")
    sim = orc.make_sim(pkg, p, sc)
    v0 = sim.download_state()["vX"][:nc].copy()
    acc = np.zeros((nc, 3), np.float32)
    acc[:, 0] = np.linspace(-2.0e3, 2.0e3, nc, dtype=np.float32)
    m.add_owner_acc(0, acc), sim.add_owner_acc(0, acc)
    m.step(2), sim.step(2)
    m.sync()
    g, o = m.download_state(), sim.download_state()
    assert np.abs(g["vX"][:nc] - o["vX"][:nc]).max() < 1e-5 and np.abs(o["vX"][:nc] - v0).max() > 5e-3
    m.close()


def test_multi_migrates_a_drifting_bed(pkg, orc):
    """a sheared bed in three library-made slabs with deme_multi_set_migration(50): clumps change slabs inside deme_multi_step, the
    gather by global id follows them (the books come from the library), and the run stays on the oracle's single-domain trajectory
    like the hand-attached slabs of tests/test_config2_slabs.py do"""
    b, p, sc, x = _sheared_bed(pkg, 20_000, 6)
    nc = int(sc.nOwnerClumps)
    m = pkg.abi.Multi(devices=(0,))
    m.build(p, sc, slabs_per_device=3, axis=0, halo=0.035, arith="exact")
    m.set_migration(50)
    sim = orc.make_sim(pkg, p, sc)
    orc.set_num_threads(min(16, os.cpu_count() or 1))
    try:
        m.step(151), sim.step(151)
        m.sync()
    finally:
        orc.set_num_threads(min(8, os.cpu_count() or 1))
    cnt, moved = m.counts()
    assert moved > 20, moved
    g, o = m.download_state(), sim.download_state()
    dx = np.abs(_positions(pkg, p, g, nc) - _positions(pkg, p, o, nc)).max()
    print(f"drifting bed in 3 library-made slabs, {moved} clumps migrated inside deme_multi_step: |dx| {dx:.3e} m vs the single-domain oracle")
    assert dx < 1e-4
    own = sum(m.slab_ctx(s).n_owners for s in range(3))
    assert own > nc  # (own clumps + ghosts + replicated owners)
    m.close()


def test_multi_fast_mode_takes_the_tile_pass_per_slab(pkg, orc):
    b, p, sc = _bed(pkg, n=6000, cd_freq=10)
    nc = int(sc.nOwnerClumps)
    m = pkg.abi.Multi(devices=(0,))
    m.build(p, sc, slabs_per_device=2, axis=-1, halo=0.03, arith="fast")
    sim = orc.make_sim(pkg, p, sc)
    m.step(40), sim.step(40)
    m.sync()
    assert m.slab_ctx(0).force_kernel()[0] == "k_tile_forces<0, false>", m.slab_ctx(0).force_kernel()
    g, o = m.download_state(), sim.download_state()
    dx = np.abs(_positions(pkg, p, g, nc) - _positions(pkg, p, o, nc)).max()
    dv = max(np.abs(g[k][:nc] - o[k][:nc]).max() for k in ("vX", "vY", "vZ"))
    assert dx < 5e-8 and dv < 1e-3, (dx, dv)
    m.close()


def test_multi_of_one_slab_is_a_plain_context(pkg, orc):
    b, p, sc = _bed(pkg)
    m = pkg.abi.Multi(devices=(0,))
    m.build(p, sc, slabs_per_device=1, arith="exact")
    sim = orc.make_sim(pkg, p, sc)
    m.step(20), sim.step(20)
    m.sync()
    g, o = m.download_state(), sim.download_state()
    assert all(np.array_equal(g[k], o[k]) for k in GKEYS)
    with pytest.raises(pkg.abi.DemeError, match="device id 99 is not present"):
        pkg.abi.Multi(devices=(0, 99))
    m.close()


def test_multi_gives_the_slabs_of_a_random_order_bed_the_engines_order(pkg, orc):
    """a bed handed over in random order (the reference's ids are load order, DEM/dT.cpp:700-800) cut into three slabs: a scene with
    ghosts keeps the order it is uploaded in, so deme_multi_build numbers every slab's clumps along the engine's own order
    (DEME_DECOMP_SPATIAL_ORDER) -- the owner-tile pass then evaluates every tile of every slab itself, where the caller's order sends
    most of them through the per-tile fallback; both runs sit on the oracle's single-domain trajectory within the fast mode's bounds"""
    b = pkg.model.packed_bed(9000, seed=5, cd_freq=10, spacing_mult=3.0, init_vz=-1.0, aspect=(2.0, 1.0, 0.25), order="random")
    b.SetExpandSafetyAdder(0.5)
    p, sc = b.Initialize()
    nc = int(sc.nOwnerClumps)
    sim = orc.make_sim(pkg, p, sc)
    sim.step(40)
    o = sim.download_state()
    big = {}
    for caller_order in (False, True):
        m = pkg.abi.Multi(devices=(0,))
        m.build(p, sc, slabs_per_device=3, axis=0, halo=0.03, arith="fast", caller_order=caller_order)
        m.step(40)
        m.sync()
        stats = [m.slab_ctx(s).tile_stats() for s in range(3)]
        assert all(m.slab_ctx(s).force_kernel()[0] == "k_tile_forces<0, false>" for s in range(3))
        big[caller_order] = (sum(t[1] for t in stats), sum(t[0] for t in stats), max(t[2] for t in stats))
        g = m.download_state()
        dx = np.abs(_positions(pkg, p, g, nc) - _positions(pkg, p, o, nc)).max()
        dv = max(np.abs(g[k][:nc] - o[k][:nc]).max() for k in ("vX", "vY", "vZ"))
        assert dx < 5e-8 and dv < 1e-3, (caller_order, dx, dv)
        m.close()
    print(f"(tiles through the per-tile fallback, tiles, largest halo): engine order {big[False]}, caller's random order {big[True]}")
    # (the bed is still loose after 40 steps -- few contacts per tile, so even the random numbering's tiles fit; the foreign owners a
    # tile stages tell the two numberings apart)
    assert big[False][0] == 0 and big[False][2] < 0.6 * big[True][2]


def test_multi_of_one_slab_in_the_fast_mode_gets_the_engines_order(pkg, orc):
    """a plan of one slab holds no ghost copy: its context is a plain one, and in the fast mode the engine reorders a bed handed
    over in random order as it does for deme_upload_scene -- every id at the boundary (the state by global id) stays the caller's"""
    b = pkg.model.packed_bed(6000, seed=5, cd_freq=0, spacing_mult=3.0, init_vz=-1.0, aspect=(1.0, 1.0, 0.25), order="random")
    p, sc = b.Initialize()
    nc = int(sc.nOwnerClumps)
    m = pkg.abi.Multi(devices=(0,))
    m.build(p, sc, slabs_per_device=1, arith="fast")
    c = m.slab_ctx(0)
    assert c.engine_order()[0], c.engine_order()
    sim = orc.make_sim(pkg, p, sc)
    st0, so0 = m.download_state(), sim.download_state()
    assert all(np.array_equal(st0[k], so0[k]) for k in GKEYS)
    m.step(30), sim.step(30)
    m.sync()
    g, o = m.download_state(), sim.download_state()
    dx = np.abs(_positions(pkg, p, g, nc) - _positions(pkg, p, o, nc)).max()
    assert dx < 5e-8, dx
    m.close()


def test_multi_migrates_along_another_axis(pkg, orc):
    """the cut axis is the plan's (the longest side of the bed: y here), and the migration kernels classify clumps by THAT coordinate
    (deme_migrate.h: mig_world_coord): a bed sheared in y, three slabs cut along y, clumps change slabs inside deme_multi_step and the
    run stays on the oracle's single-domain trajectory"""
    b = pkg.model.packed_bed(20_000, seed=6, cd_freq=0, spacing_mult=2.5, init_vz=-0.4, aspect=(1.0, 2.0, 0.5))
    p, sc = b.Initialize()
    nc = int(sc.nOwnerClumps)
    b.arrays["vY"][:nc] = np.where(np.arange(nc) % 2 == 0, 0.6, 0.3).astype(np.float32)
    sc = pkg.abi.make_scene_struct(b.arrays, b.counts)
    m = pkg.abi.Multi(devices=(0,))
    m.build(p, sc, slabs_per_device=3, axis=-1, halo=0.035, arith="exact")
    plan_axis = 1
    sim = orc.make_sim(pkg, p, sc)
    m.set_migration(50)
    orc.set_num_threads(min(16, os.cpu_count() or 1))
    try:
        m.step(151), sim.step(151)
        m.sync()
    finally:
        orc.set_num_threads(min(8, os.cpu_count() or 1))
    cnt, moved = m.counts()
    assert moved > 20, moved
    g, o = m.download_state(), sim.download_state()
    dx = np.abs(_positions(pkg, p, g, nc) - _positions(pkg, p, o, nc)).max()
    print(f"bed sheared in y, 3 slabs along axis {plan_axis}, {moved} clumps migrated: |dx| {dx:.3e} m vs the single-domain oracle")
    assert dx < 1e-4
    m.close()


def test_multi_rebalance_moves_the_slab_boundaries(pkg, orc):
    """deme_multi_rebalance: a bed whose clumps all drift in +x empties its first slab and crowds its last one; the boundaries are
    recomputed from the current positions (equal counts, bin-aligned), the library's migration moves the clumps that now lie beyond
    them -- the counts are evener, every clump is owned once.  The run that rebalanced and a twin that did not start from the same
    state and are compared 101 steps later (the bed is landing and hitting a wall at 2 m/s by then: against the oracle's
    single-domain trajectory both carry the same amplified summation-order difference, which is recorded, and which the rebalance
    must not add to)."""
    b = pkg.model.packed_bed(20_000, seed=6, cd_freq=0, spacing_mult=2.5, init_vz=-0.2, aspect=(2.0, 1.0, 0.5))
    p, sc = b.Initialize()
    nc = int(sc.nOwnerClumps)
    b.arrays["vX"][:nc] = 2.0  # the whole bed drifts
    sc = pkg.abi.make_scene_struct(b.arrays, b.counts)
    runs = []
    for _ in range(2):
        m = pkg.abi.Multi(devices=(0,))
        m.build(p, sc, slabs_per_device=4, axis=0, halo=0.035, arith="exact")
        m.set_migration(100)
        runs.append(m)
    m, twin = runs
    sim = orc.make_sim(pkg, p, sc)
    orc.set_num_threads(min(16, os.cpu_count() or 1))
    try:
        m.step(1500), twin.step(1500), sim.step(1500)  # 3 cm of drift: the lower slabs thin out, the top one fills
        m.sync(), twin.sync()
        x0, xt0, xo0 = (_positions(pkg, p, r.download_state(), nc) for r in (m, twin, sim))
        assert np.array_equal(x0, xt0)  # (exact mode: the twins are bit-identical so far)
        dx_o0 = np.abs(x0 - xo0).max()
        own0 = [m.slab_counts(s)[0][0] for s in range(4)]
        e0 = [m.slab_counts(s)[1] for s in range(4)]
        moved, edges = m.rebalance()
        own1 = [m.slab_counts(s)[0][0] for s in range(4)]
        assert sum(own0) == sum(own1) == nc and moved > 0
        assert max(own1) - min(own1) < max(own0) - min(own0), (own0, own1)
        inner = (edges[1:-1] - float(p.LBFX)) / float(p.binSize)
        assert np.abs(inner - np.rint(inner)).max() < 1e-9 and all(edges[i + 1] - edges[i] >= 0.035 for i in range(1, 3))
        assert np.array_equal(_positions(pkg, p, m.download_state(), nc), x0)  # moving clumps between slabs moves none of them in space
        m.step(101), twin.step(101), sim.step(101)
        m.sync(), twin.sync()
    finally:
        orc.set_num_threads(min(8, os.cpu_count() or 1))
    x1, xt1, xo1 = (_positions(pkg, p, r.download_state(), nc) for r in (m, twin, sim))
    dx_t, dx_o1, dx_to1 = np.abs(x1 - xt1).max(), np.abs(x1 - xo1).max(), np.abs(xt1 - xo1).max()
    print(f"drifting bed, 4 slabs: own clumps {own0} -> {own1} after deme_multi_rebalance ({moved} moved; boundaries "
          f"{np.round(edges[1:-1], 4).tolist()}, were {[round(e[1], 4) for e in e0[:-1]]}); 101 steps later |dx| {dx_t:.3e} m vs the twin that "
          f"kept its boundaries; vs the oracle {dx_o0:.3e} m before, {dx_o1:.3e} m after (the twin: {dx_to1:.3e} m)")
    assert dx_t < max(1e-6, 3 * dx_o0)
    assert dx_o1 < 3 * max(dx_o0, dx_to1) + 1e-7
    m.close(), twin.close()


def test_slabs_that_grow_keep_their_scratch_in_step(pkg, monkeypatch):
    """A slab that a migration left with MORE spheres than it was uploaded with needs every per-sphere / per-owner buffer of the
    context at the new size (one list of them serves deme_upload_scene and the migration).  The drifting bed fills its upper slabs:
    twelve migrations with the library's own check of the books after each (DEME_MIG_CHECK: every id of every re-assembled slab
    names a row of the global scene), and the state read back by global id every time.  (Before the lists were one, the ghosts'
    family words of a grown slab were written past their buffer -- into the neighbour allocation, which here was the id table.)"""
    monkeypatch.setenv("DEME_MIG_CHECK", "1")
    b = pkg.model.packed_bed(20_000, seed=6, cd_freq=0, spacing_mult=2.5, init_vz=-0.2, aspect=(2.0, 1.0, 0.5))
    p, sc = b.Initialize()
    nc = int(sc.nOwnerClumps)
    b.arrays["vX"][:nc] = 2.0
    sc = pkg.abi.make_scene_struct(b.arrays, b.counts)
    spread = {}
    for caller_order in (True, False):
        m = pkg.abi.Multi(devices=(0,))
        m.build(p, sc, slabs_per_device=4, axis=0, halo=0.035, arith="exact", caller_order=caller_order)
        m.set_migration(100)
        if not caller_order:
            m.set_rebalance(3)  # (the second run also moves its boundaries, at every third migration)
        grew = 0
        n0 = [m.slab_counts(s)[0][4] for s in range(4)]
        for _ in range(12):
            m.step(100), m.sync()
            st = m.download_state()
            assert np.isfinite(st["vX"][:nc]).all()
            grew = max(grew, max(m.slab_counts(s)[0][4] - n0[s] for s in range(4)))
        own = [m.slab_counts(s)[0][0] for s in range(4)]
        assert sum(own) == nc
        spread[caller_order] = max(own) - min(own)
        if caller_order:
            assert grew > 500, grew  # (a slab did outgrow its upload by hundreds of spheres)
        m.close()
    print(f"own-clump spread over 4 slabs after 1200 steps of drift: {spread[True]} with fixed boundaries, {spread[False]} rebalanced at every third migration")
    assert spread[False] < spread[True]  # (a boundary sits on a bin face: one 1.6 cm column of this bed holds ~500 clumps)


def test_multi_restart_from_a_merged_list_and_marks(pkg, orc):
    """Restart of a decomposed run: the merged contact list with its history (global ids) and the state by global id go into a FRESH
    decomposition of another slab count (deme_multi_seed_contacts: every slab that holds a pair gets it in its own ids, vector
    wildcards with its own sign) and the run continues like the one that was never interrupted -- and like the oracle's single-domain
    run restarted the same way.  Marked (persistent) pairs: marked on every slab, reported once in global ids, carried into the new
    decomposition."""
    b, p, sc = _bed(pkg, n=2400, seed=8, cd_freq=0)
    nc = int(sc.nOwnerClumps)
    a = pkg.abi.Multi(devices=(0,))
    a.build(p, sc, slabs_per_device=3, axis=0, halo=0.03, arith="exact")
    a.step(300), a.sync()
    st = a.download_state()
    ga, gb, gt = a.contacts()
    W = np.stack([a.wildcard(w) for w in range(4)], 1)
    assert len(ga) > 800 and np.abs(W[:, :3]).max() > 0  # (a list with tangential history)
    a.mark_persistent_contacts(0)
    pa, pb, pt = a.persistent_contacts()
    assert len(pa) == len(ga) and np.array_equal(pa, ga) and np.array_equal(pb, gb) and np.array_equal(pt, gt)  # once each, canonical
    # the restarted run: two slabs this time
    r = pkg.abi.Multi(devices=(0,))
    r.build(p, sc, slabs_per_device=2, axis=0, halo=0.03, arith="exact")
    r.upload_state({k: st[k] for k in GKEYS})
    r.seed_contacts(ga, gb, gt, W)
    r.set_persistent_contacts(pa, pb, pt)
    qa, qb, qt = r.persistent_contacts()
    assert np.array_equal(qa, pa) and np.array_equal(qb, pb) and np.array_equal(qt, pt)
    sim = orc.make_sim(pkg, p, sc)
    sim.upload_state({k: st[k] for k in GKEYS})
    sim.seed_contacts(ga, gb, gt, W)
    a.step(40), r.step(40), sim.step(40)
    a.sync(), r.sync()
    xa, xr, xo = (_positions(pkg, p, m.download_state(), nc) for m in (a, r, sim))
    d_ar, d_ro = np.abs(xa - xr).max(), np.abs(xr - xo).max()
    print(f"restart of a decomposed run: 40 steps later |dx| {d_ar:.3e} m vs the uninterrupted 3-slab run, {d_ro:.3e} m vs the oracle restarted alike")
    assert d_ar < 5e-9 and d_ro < 5e-9
    # without the history the restarted run is somewhere else (the seeded rows matter)
    n = pkg.abi.Multi(devices=(0,))
    n.build(p, sc, slabs_per_device=2, axis=0, halo=0.03, arith="exact")
    n.upload_state({k: st[k] for k in GKEYS})
    n.step(40), n.sync()
    assert np.abs(_positions(pkg, p, n.download_state(), nc) - xa).max() > 20 * max(d_ar, 1e-10)
    # the marks are live in the restarted run: every marked pair stays listed
    la, lb, lt = r.contacts()
    listed = set(zip(la.tolist(), lb.tolist(), lt.tolist()))
    assert set(zip(pa.tolist(), pb.tolist(), pt.tolist())) <= listed
    for m in (a, r, n):
        m.close()
