"""BASELINE configs[2] (the bed cut into x-slabs with a ghost halo, RCCL exchange every step) through the LIBRARY-side loop
(deme_halo_group_*: C++ orchestration, ncclSend / ncclRecv in one group per step on an exchange stream, overlapped with the
interior force pass).  One GPU is available to the tests, so the slabs are contexts of one process and their records travel
through RCCL sends to self on a one-rank communicator -- the same code path, peers aside, as one process per GPU.

  * small: 2 and 3 slabs, 60 steps with detection every 7: bit-identical to the ordered pack -> unpack -> deme_step loop;
  * full size: 1e6 clumps (configs[1] recipe, packed) cut into 2 and into 8 slabs -- the union of the slabs' contact lists,
    mapped to global ids, equals the single-domain ORACLE list pair for pair, and after 100 steps every clump is within the
    stated tolerance of the oracle's single-domain run."""
import os

import numpy as np
import pytest

from tests.test_decomp import GKEYS, _global_rows, _sheared_bed, _state_x, _two_family_bed, build_global, gather_positions

pytestmark = pytest.mark.gpu


def _make(pkg, p, scene, mode="exact"):
    ctx = pkg.Context(0)
    ctx.set_arith_mode(mode)
    ctx.set_params(p), ctx.upload_scene(scene)
    return ctx


def _group(pkg, ctxs, parts):
    g = pkg.abi.HaloGroup(rank=0, world=1, device=0)
    for i, (c, pt) in enumerate(zip(ctxs, parts)):
        g.attach(c, pt, left=ctxs[i - 1] if i > 0 else None, right=ctxs[i + 1] if i + 1 < len(ctxs) else None)
    return g


@pytest.mark.parametrize("n_slabs", [2, 3])
def test_library_halo_loop_equals_ordered_exchange(pkg, n_slabs):
    import ctypes as C
    b, p, sc, x = build_global(pkg, n=3000, seed=6, cd_freq=7)
    parts = pkg.decomp.decompose(b.arrays, b.counts, x, n_slabs, halo=0.035)
    hip = C.CDLL("libamdhip64.so")

    def dev(host=None, nbytes=0):
        ptr = C.c_void_p()
        nbytes = host.nbytes if host is not None else nbytes
        assert hip.hipMalloc(C.byref(ptr), C.c_size_t(max(nbytes, 16))) == 0
        if host is not None and host.nbytes:
            assert hip.hipMemcpy(ptr, C.c_void_p(host.ctypes.data), C.c_size_t(host.nbytes), 1) == 0
        return ptr.value

    steps = 60
    plain = [_make(pkg, p, pt["scene"]) for pt in parts]
    ids = [{k: dev(np.ascontiguousarray(pt[k].astype(np.uint32))) for k in ("send_left", "send_right", "recv_left", "recv_right")}
           for pt in parts]
    bufs = [(dev(nbytes=len(parts[i]["send_right"]) * pkg.abi.GHOST_BYTES), dev(nbytes=len(parts[i + 1]["send_left"]) * pkg.abi.GHOST_BYTES))
            for i in range(n_slabs - 1)]
    for _ in range(steps):
        for i in range(n_slabs - 1):
            plain[i].halo_pack(ids[i]["send_right"], len(parts[i]["send_right"]), bufs[i][0])
            plain[i + 1].halo_pack(ids[i + 1]["send_left"], len(parts[i + 1]["send_left"]), bufs[i][1])
        for c in plain:
            c.sync()
        for i in range(n_slabs - 1):
            plain[i + 1].halo_unpack(ids[i + 1]["recv_left"], len(parts[i]["send_right"]), bufs[i][0])
            plain[i].halo_unpack(ids[i]["recv_right"], len(parts[i + 1]["send_left"]), bufs[i][1])
        for c in plain:
            c.step(1)
    lib = [_make(pkg, p, pt["scene"]) for pt in parts]
    g = _group(pkg, lib, parts)
    g.step(steps)
    g.sync()
    n_ex, nbytes = g.stats()
    assert n_ex == steps and nbytes == sum(len(pt["send_left"]) + len(pt["send_right"]) for pt in parts) * pkg.abi.GHOST_BYTES
    for a, b_ in zip(plain, lib):
        sa, sb = a.download_state(), b_.download_state()
        for k in GKEYS:
            assert np.array_equal(sa[k], sb[k]), k
        assert int(a.counts().nContacts) == int(b_.counts().nContacts) > 100
        assert np.array_equal(a.wildcard(3), b_.wildcard(3))
    g.close()


@pytest.mark.parametrize("mode", ["exact", "fast"])
def test_family_masks_and_margins_hold_across_cuts_on_the_gpu(pkg, orc, mode):
    """three slabs through the library loop, two families that must not touch and a family margin: the union of the slabs'
    contact lists equals the single-domain ORACLE list (ghost copies keep their family; ghost-ghost pairs are nobody's here)"""
    b, p, sc, x = _two_family_bed(pkg, n=3000, seed=6, cd_freq=5)
    parts = pkg.decomp.decompose(b.arrays, b.counts, x, 3, halo=0.035)
    ctxs = [_make(pkg, p, pt["scene"], mode) for pt in parts]
    g = _group(pkg, ctxs, parts)
    one = orc.make_sim(pkg, p, sc)
    steps = 61  # detections at steps 0, 5, ..., 60: the lists below are of the same instant
    g.step(steps), one.step(steps)
    g.sync()
    rows = np.unique(np.concatenate([_global_rows(pt, c) for pt, c in zip(parts, ctxs)]), axis=0)
    a, bb, t, _ = one.contacts()
    ref = np.unique(np.stack([a.astype(np.int64), bb.astype(np.int64), t.astype(np.int64)], 1), axis=0)
    assert len(ref) > 300 and np.array_equal(rows, ref)
    own, fam = np.asarray(b.arrays["ownerClumpBody"], np.int64), np.asarray(b.arrays["familyID"])
    ss = ref[:, 2] == 1
    assert ss.sum() > 100 and (fam[own[ref[ss, 0]]] == fam[own[ref[ss, 1]]]).all()
    X, V = gather_positions(pkg, parts, ctxs, p, sc.nOwnerClumps)
    st = one.download_state()
    n = sc.nOwnerClumps
    X1 = pkg.model.decode_positions(st["voxelID"], st["locX"], st["locY"], st["locZ"], p.nvXp2, p.nvYp2, p.voxelSize, p.l)[:n]
    assert np.abs(X - X1).max() < 2e-7
    g.close()


@pytest.mark.parametrize("mode,n_slabs,migrate", [("exact", 2, False), ("fast", 3, False), ("fast", 3, True)])
def test_free_mesh_body_across_slabs(pkg, orc, mode, n_slabs, migrate):
    """A mesh body that moves under contact forces (a plate lying on the bed, wider than any slab) is kept on every slab
    (DemeScene.ownerGhost = 2): each slab sums the forces of its own spheres on it, deme_halo_group_step all-reduces the sums, and
    every replica takes the same step.  Against the single-domain oracle: same contact list, plate and clumps within the stated
    tolerance (the per-slab sums are added in a different order than the single domain's one sum: fp32 rounding of a 0.5 kg
    body's acceleration), replicas bit-identical to each other."""
    from tests.test_mesh import mesh_bed
    b = mesh_bed(pkg, 1500, seed=3, cd_freq=5, fixed=False)
    p, sc = b.Initialize()
    x = np.concatenate([bb.xyz for bb in b.batches])[:, 0]
    with pytest.raises(ValueError, match="move under contact forces"):
        pkg.decomp.decompose(b.arrays, b.counts, x, n_slabs, halo=0.035)
    parts = pkg.decomp.decompose(b.arrays, b.counts, x, n_slabs, halo=0.035, shared_free=True)
    mesh_local = [int(np.nonzero(pt["arrays"]["ownerGhost"] == 2)[0][0]) for pt in parts]
    ctxs = [_make(pkg, p, pt["scene"], mode) for pt in parts]
    with pytest.raises(RuntimeError, match="replicated free owners"):
        ctxs[0].step(1)  # its share of the plate's force alone would be integrated
    g = _group(pkg, ctxs, parts)
    one = orc.make_sim(pkg, p, sc)
    steps = 121  # detections at steps 0, 5, ..., 120
    if migrate:  # the `migrate` leg: the slabs are re-assembled inside the library half way (deme_halo_group_migrate with a
                 # replicated free body: its owner number moves with the slab's clump count, the all-reduce list follows)
        for c, pt in zip(ctxs, parts):
            g.set_slab(c, pt, 0.035)
        g.step(60)
        g.migrate()
        g.step(steps - 60)
        one.step(steps)
        g.sync()
        ids = [g.slab_ids(c) for c in ctxs]
        cnts = [g.slab_counts(c) for c in ctxs]
        parts = [dict(pt, sphere_global=np.asarray(i[1], np.int64), owner_global=np.asarray(i[0], np.int64), n_own=k[0],
                      global_ids=np.asarray(i[0], np.int64)[:k[0]],
                      arrays=dict(pt["arrays"], ownerClumpBody=np.asarray(i[2], np.int64))) for pt, i, k in zip(parts, ids, cnts)]
        mesh_local = [int(k[3]) - 1 for k in cnts]  # the plate is the last owner of every re-assembled slab
    else:
        g.step(steps), one.step(steps)
        g.sync()
    rows = np.unique(np.concatenate([_global_rows(pt, c) for pt, c in zip(parts, ctxs)]), axis=0)
    a, bb, t, _ = one.contacts()
    ref = np.unique(np.stack([a.astype(np.int64), bb.astype(np.int64), t.astype(np.int64)], 1), axis=0)
    assert (ref[:, 2] == 2).sum() > 50 and np.array_equal(rows, ref)  # sphere-triangle pairs included, each on one slab
    so = one.download_state()
    sts = [c.download_state() for c in ctxs]
    keys = ("voxelID", "locX", "locY", "locZ", "oriQw", "oriQx", "oriQy", "oriQz", "vX", "vY", "vZ", "omgBarX", "omgBarY", "omgBarZ")
    for st, m in zip(sts[1:], mesh_local[1:]):  # the replicas took the same steps
        assert all(st[k][m] == sts[0][k][mesh_local[0]] for k in keys)
    m0, mg = mesh_local[0], int(sc.nOwners) - 1
    assert so["vZ"][mg] != 0.0
    for k in ("vX", "vY", "vZ", "omgBarX", "omgBarY", "omgBarZ"):
        assert abs(sts[0][k][m0] - so[k][mg]) <= 2e-4 * max(1.0, abs(so[k][mg])), k
    Xm = pkg.model.decode_positions(sts[0]["voxelID"], sts[0]["locX"], sts[0]["locY"], sts[0]["locZ"], p.nvXp2, p.nvYp2, p.voxelSize, p.l)[m0]
    Xo = pkg.model.decode_positions(so["voxelID"], so["locX"], so["locY"], so["locZ"], p.nvXp2, p.nvYp2, p.voxelSize, p.l)
    assert np.abs(Xm - Xo[mg]).max() < 1e-7
    X, V = gather_positions(pkg, parts, ctxs, p, sc.nOwnerClumps)
    assert np.abs(X - Xo[:sc.nOwnerClumps]).max() < 1e-6
    g.close()


def test_neighbour_migration_between_slabs_on_the_gpu(pkg):
    """a sheared bed in three slabs through the library loop: after 150 steps clumps have crossed the cuts; they move to the face
    neighbour with state, template data and contact history (decomp.migrate_neighbours_in_process: the per-rank functions of the
    distributed path, lists as the transport), new contexts take over, and 20 steps later every clump is where the old slabs,
    simply continuing, put it -- to fp32 summation order (local numbering differs)"""
    b, p, sc, x = _sheared_bed(pkg, 3000, 6)
    halo = 0.035
    parts = pkg.decomp.decompose(b.arrays, b.counts, x, 3, halo=halo)
    ctxs = [_make(pkg, p, pt["scene"]) for pt in parts]
    g = _group(pkg, ctxs, parts)
    g.step(150)
    g.sync()
    states = [c.download_state() for c in ctxs]
    cnts = [c.contacts() for c in ctxs]
    Ws = [np.stack([c.wildcard(w) for w in range(4)], 1) for c in ctxs]
    parts2, seeds = pkg.decomp.migrate_neighbours_in_process(parts, states, cnts, Ws, parts[0]["all_edges"], halo, _state_x(pkg, p))
    moved = sum(len(np.setdiff1d(a["global_ids"], b_["global_ids"])) for a, b_ in zip(parts2, parts))
    assert moved >= 3
    ctxs2 = [_make(pkg, p, pt["scene"]) for pt in parts2]
    for c, sd in zip(ctxs2, seeds):
        c.seed_contacts(*sd)
    g2 = _group(pkg, ctxs2, parts2)
    g.step(20), g2.step(20)
    g.sync(), g2.sync()
    X, V = gather_positions(pkg, parts2, ctxs2, p, sc.nOwnerClumps)
    X0, V0 = gather_positions(pkg, parts, ctxs, p, sc.nOwnerClumps)
    assert np.abs(X - X0).max() < 1e-9 and np.abs(V - V0).max() < 1e-5, (np.abs(X - X0).max(), np.abs(V - V0).max())
    g.close(), g2.close()


def test_library_migration_equals_the_numpy_statement(pkg):
    """deme_halo_group_migrate (device kernels + buffer hand-over / RCCL inside the library) against decomp.migrate_neighbours (the
    numpy statement of the same algorithm): a sheared bed in three slabs, 150 steps, clumps have crossed the cuts.  Both leave the
    same slabs behind -- own clumps, ghosts, numbering, seeded history (pairs and wildcards) -- and 20 steps later the two sets of
    slabs are in the SAME state bit for bit (same numbering, same summation order), which is also the state of the old slabs
    simply continuing to fp32 summation order (1e-9 m)."""
    b, p, sc, x = _sheared_bed(pkg, 3000, 6)
    halo = 0.035
    parts = pkg.decomp.decompose(b.arrays, b.counts, x, 3, halo=halo)
    runs = []
    for _ in range(3):  # old slabs continuing / numpy migration / library migration: three identical starts
        ctxs = [_make(pkg, p, pt["scene"]) for pt in parts]
        g = _group(pkg, ctxs, parts)
        g.step(150)
        g.sync()
        runs.append((ctxs, g))
    (c0, g0), (c1, g1), (c2, g2) = runs
    # numpy statement
    states = [c.download_state() for c in c1]
    cnts = [c.contacts() for c in c1]
    Ws = [np.stack([c.wildcard(w) for w in range(4)], 1) for c in c1]
    partsP, seeds = pkg.decomp.migrate_neighbours_in_process(parts, states, cnts, Ws, parts[0]["all_edges"], halo, _state_x(pkg, p))
    ctxsP = [_make(pkg, p, pt["scene"]) for pt in partsP]
    for c, sd in zip(ctxsP, seeds):
        c.seed_contacts(*sd)
    gP = _group(pkg, ctxsP, partsP)
    # library
    for c, pt in zip(c2, parts):
        g2.set_slab(c, pt, halo)
    moved = g2.migrate()
    assert moved >= 3 and moved == sum(len(np.setdiff1d(a["global_ids"], b_["global_ids"])) for a, b_ in zip(parts, partsP))
    for c, pt, sd in zip(c2, partsP, seeds):
        n_own, n_gl, n_gr, n_o, n_s, n_seed = g2.slab_counts(c)
        assert (n_own, n_gl, n_gr) == (pt["n_own"], len(pt["ghost_left_g"]), len(pt["ghost_right_g"]))
        og, sg, so, scmp = g2.slab_ids(c)
        assert np.array_equal(og, np.asarray(pt["owner_global"], np.uint32))
        assert np.array_equal(sg, np.asarray(pt["sphere_global"], np.uint32))
        assert np.array_equal(so, np.asarray(pt["arrays"]["ownerClumpBody"], np.uint32))
        assert np.array_equal(scmp, np.asarray(pt["arrays"]["clumpComponentOffset"], np.uint16))
        a, bb, t, _ = c.contacts()  # the seeded list
        key = lambda A, B, T: np.lexsort((B, T, A))
        ka, ks = key(a, bb, t), key(sd[0], sd[1], sd[2])
        assert n_seed == len(sd[0]) == len(a)
        assert np.array_equal(a[ka], sd[0][ks]) and np.array_equal(bb[ka], sd[1][ks]) and np.array_equal(t[ka], sd[2][ks])
        Wl = np.stack([c.wildcard(w) for w in range(4)], 1)
        assert np.array_equal(Wl[ka], np.asarray(sd[3], np.float32)[ks])
    g0.step(20), gP.step(20), g2.step(20)
    g0.sync(), gP.sync(), g2.sync()
    for cl, cp in zip(c2, ctxsP):  # library == numpy statement, bit for bit
        sl, sp = cl.download_state(), cp.download_state()
        for k in GKEYS:
            assert np.array_equal(sl[k], sp[k]), k
    X, V = gather_positions(pkg, partsP, c2, p, sc.nOwnerClumps)
    X0, V0 = gather_positions(pkg, parts, c0, p, sc.nOwnerClumps)
    assert np.abs(X - X0).max() < 1e-9 and np.abs(V - V0).max() < 1e-5, (np.abs(X - X0).max(), np.abs(V - V0).max())
    for g in (g0, g1, gP, g2):
        g.close()


def test_library_migration_carries_persistent_marks(pkg):
    """Marked (persistent) contacts across a migration inside the library: the marks ride in the history rows (a bit of the row
    header), so a pair stays marked wherever its clumps go.  A sheared bed in three slabs, every contact of every slab marked after
    150 steps, then deme_halo_group_migrate: the union of the slabs' marked sets, in GLOBAL sphere ids, is what it was (a pair is
    kept by every slab that owns one of its clumps), every slab's marks are pairs of its own re-assembled list, and the slabs step
    on (a detection with the marks appended)."""
    b, p, sc, x = _sheared_bed(pkg, 3000, 6)
    halo = 0.035
    parts = pkg.decomp.decompose(b.arrays, b.counts, x, 3, halo=halo)
    ctxs = [_make(pkg, p, pt["scene"]) for pt in parts]
    g = _group(pkg, ctxs, parts)
    g.step(150)
    g.sync()

    def marked_global(c, sphere_global, n_own, owner_of):
        a, bb, t = c.persistent_contacts()
        sg = np.asarray(sphere_global, np.int64)
        ss = t == 1
        gA = sg[a]
        gB = np.where(ss, sg[np.where(ss, bb, 0)], bb.astype(np.int64))
        lo, hi = np.where(ss & (gA > gB), gB, gA), np.where(ss & (gA > gB), gA, gB)
        own = (owner_of[a] < n_own) | (ss & (owner_of[np.where(ss, bb, 0)] < n_own))
        return {(int(l), int(h), int(k)) for l, h, k, o in zip(lo, hi, t, own) if o}

    before = set()
    for c, pt in zip(ctxs, parts):
        c.mark_persistent_contacts(0)
        assert c.num_persistent_contacts() == int(c.counts().nContacts) > 100
        before |= marked_global(c, pt["sphere_global"], pt["n_own"], np.asarray(pt["arrays"]["ownerClumpBody"]))
    for c, pt in zip(ctxs, parts):
        g.set_slab(c, pt, halo)
    moved = g.migrate()
    assert moved >= 3
    after = set()
    for c in ctxs:
        n_own, n_gl, n_gr, n_o, n_s, n_seed = g.slab_counts(c)
        og, sg, so, scmp = g.slab_ids(c)
        after |= marked_global(c, sg, n_own, so)
        a, bb, t = c.persistent_contacts()
        la, lb, lt, _ = c.contacts()
        have = set(zip(la.tolist(), lb.tolist(), lt.tolist()))
        assert all((int(x), int(y), int(z)) in have for x, y, z in zip(a, bb, t))
    assert after == before and len(before) > 1000, (len(before), len(after), len(before ^ after))
    g.step(20)
    g.sync()
    for c in ctxs:
        st = c.download_state()
        assert np.isfinite(st["vX"]).all()
        assert int(c.counts().nContacts) >= c.num_persistent_contacts() > 100
    g.close()


def test_drifting_bed_with_repeated_library_migrations_against_single_domain_oracle(pkg, orc):
    """1e5 clumps with a lateral drift (neighbouring clumps fly at 0.6 and 0.3 m/s in x: the bed shears, hundreds of clumps cross
    the two cuts) in three slabs on one GPU, deme_halo_group_migrate every 50 steps, 200 steps -- against the ORACLE's
    single-domain run of the same bed and against the same three slabs simply continuing (valid for these 200 steps: nobody
    leaves a halo yet).  STATED TOLERANCE: after the last step the union of the slabs' contact lists is the oracle's list; the
    migrated slabs are within 1e-7 m / 1e-3 m/s of the continuing ones (measured 1.3e-8 m / 2e-4 m/s: the migrations carry every
    history row, the two differ by local numbering = fp32 summation order, which this colliding bed amplifies; 20 steps after ONE
    migration the difference is < 1e-9 m and the numpy statement is met bit for bit -- the test above); against the single-domain oracle both sets of slabs sit at the SAME distance --
    this bed of clumps colliding at 0.3 m/s amplifies a summation-order ulp to ~1e-5 m in 200 steps, slabs or no slabs -- bound
    1e-4 m, and the two distances must agree to 1e-7 m."""
    b, p, sc, x = _sheared_bed(pkg, 100_000, 6)
    nc = int(sc.nOwnerClumps)
    halo = 0.035
    parts = pkg.decomp.decompose(b.arrays, b.counts, x, 3, halo=halo)
    ctxs = [_make(pkg, p, pt["scene"]) for pt in parts]
    grp = _group(pkg, ctxs, parts)
    for c, pt in zip(ctxs, parts):
        grp.set_slab(c, pt, halo)
    still = [_make(pkg, p, pt["scene"]) for pt in parts]  # the same slabs, never migrated
    gst = _group(pkg, still, parts)
    gst.step(201)
    gst.sync()
    Xs, Vs = gather_positions(pkg, parts, still, p, nc)
    gst.close()
    for c in still:
        c.close()
    sim = orc.make_sim(pkg, p, sc)
    orc.set_num_threads(min(32, os.cpu_count() or 1))
    moved = 0
    try:
        for _ in range(4):
            grp.step(50), sim.step(50)
            grp.sync()
            moved += grp.migrate()
        grp.step(1), sim.step(1)  # (a migration leaves the slabs due for a detection: compare lists of the same state)
        grp.sync()
    finally:
        orc.set_num_threads(min(8, os.cpu_count() or 1))
    assert moved > 100, moved
    # the slabs' current books from the library
    pos = np.zeros((nc, 3)), np.zeros((nc, 3))
    seen = np.zeros(nc, bool)
    rows = []
    for c in ctxs:
        n_own, n_gl, n_gr, n_o, n_s, _ = grp.slab_counts(c)
        og, sg, so, _ = grp.slab_ids(c)
        st = c.download_state()
        X = pkg.model.decode_positions(st["voxelID"], st["locX"], st["locY"], st["locZ"], p.nvXp2, p.nvYp2, p.voxelSize, p.l)
        gid = og[:n_own].astype(np.int64)
        assert not seen[gid].any()
        seen[gid] = True
        pos[0][gid] = X[:n_own]
        pos[1][gid] = np.stack([st["vX"], st["vY"], st["vZ"]], 1)[:n_own]
        a, bb, t, _ = c.contacts()
        ss = t == 1
        gA = sg[a].astype(np.int64)
        gB = np.where(ss, sg[np.where(ss, bb, 0)].astype(np.int64), bb.astype(np.int64))
        flip = ss & (gA > gB)
        rows.append(np.stack([np.where(flip, gB, gA), np.where(flip, gA, gB), t.astype(np.int64)], 1))
    assert seen.all()  # every clump owned exactly once
    ref = sim.contacts()
    ref_rows = np.unique(np.stack([ref[0].astype(np.int64), ref[1].astype(np.int64), ref[2].astype(np.int64)], 1), axis=0)
    got = np.unique(np.concatenate(rows), axis=0)
    assert got.shape == ref_rows.shape and np.array_equal(got, ref_rows), (got.shape, ref_rows.shape)
    so = sim.download_state()
    Xo = pkg.model.decode_positions(so["voxelID"], so["locX"], so["locY"], so["locZ"], p.nvXp2, p.nvYp2, p.voxelSize, p.l)[:nc]
    Vo = np.stack([so["vX"], so["vY"], so["vZ"]], 1)[:nc]
    dx, dv = np.abs(pos[0] - Xo).max(), np.abs(pos[1] - Vo).max()
    dxs, dvs = np.abs(pos[0] - Xs).max(), np.abs(pos[1] - Vs).max()
    dx0 = np.abs(Xs - Xo).max()
    print(f"drifting bed, 3 slabs, 4 library migrations ({moved} clumps moved): |dx| {dxs:.3e} m, |dv| {dvs:.3e} m/s vs the slabs continuing; "
          f"{dx:.3e} m vs the single-domain oracle (the continuing slabs: {dx0:.3e} m)")
    assert dxs < 1e-7 and dvs < 1e-3
    assert dx < 1e-4 and abs(dx - dx0) < 1e-7
    grp.close()


@pytest.mark.parametrize("mode", ["exact", "fast"])
def test_cross_cut_contacts_evaluated_once(pkg, mode):
    """deme_halo_group_set_cross_contacts on three slabs of a settling bed (detection every 7 steps), against the same slabs in
    the default mode (both sides evaluate, each with its own history): the union of the lists is the same set, but now no pair
    is on two lists and the lists together are exactly as long as the set; the clumps land where the default mode puts them
    within 2e-7 m / 2e-4 m/s after 120 steps (the reaction a neighbour returns is added to a clump's own sum in another order than
    the slab's own evaluation would add it); the exchange ran every step in both directions."""
    b, p, sc, x = build_global(pkg, n=3000, seed=6, cd_freq=7)
    parts = pkg.decomp.decompose(b.arrays, b.counts, x, 3, halo=0.035)
    nc = int(sc.nOwnerClumps)
    res = []
    for once in (False, True):
        ctxs = [_make(pkg, p, pt["scene"], mode) for pt in parts]
        g = _group(pkg, ctxs, parts)
        if once:
            g.set_cross_contacts(True)
        g.step(120)
        g.sync()
        rows_all = np.concatenate([_global_pairs(pt, c) for pt, c in zip(parts, ctxs)])
        X, V = gather_positions(pkg, parts, ctxs, p, nc)
        res.append((rows_all, X, V))
        g.close()
        for c in ctxs:
            c.close()
    (r0, X0, V0), (r1, X1, V1) = res
    u0, u1 = np.unique(r0, axis=0), np.unique(r1, axis=0)
    assert len(r0) > len(u0) + 100, "the default mode lists every straddling pair twice"
    assert len(r1) == len(u1), "a pair on two lists"
    # the two runs are 120 steps apart from a common start: their lists agree up to the handful of near-pairs whose margins the
    # trajectories' last bits decide
    common = len(set(map(tuple, u0.tolist())) & set(map(tuple, u1.tolist())))
    assert common >= 0.995 * max(len(u0), len(u1)), (common, len(u0), len(u1))
    dx, dv = np.abs(X1 - X0).max(), np.abs(V1 - V0).max()
    print(f"one evaluation per cross-cut contact, {mode}: {len(r0) - len(u0)} straddling pairs, |dx| {dx:.3e} m, |dv| {dv:.3e} m/s vs both-sides evaluation")
    assert dx <= 2e-7 and dv <= 2e-4, (dx, dv)


@pytest.fixture(scope="module")
def packed_million(pkg):
    """configs[1] bed, settled on one GPU context (exact mode): params, scene, builder, state, contact list + history"""
    import bench
    b = bench.build_bed(pkg, 1_000_000, 2024, 40, order="morton")  # bench.py's numbering (owner tiles fit LDS)
    p, sc = b.Initialize()
    ctx = _make(pkg, p, sc)
    ctx.step(22000)
    st = ctx.download_state()
    cnt = ctx.contacts()
    W = np.stack([ctx.wildcard(w) for w in range(4)], 1)
    ctx.close()
    return b, p, sc, st, cnt, W


def _global_pairs(part, ctx):
    """the slab's contact list in global ids: (sphere A, geometry B, type) rows, sphere pairs smaller id first"""
    a, bb, t, _ = ctx.contacts()
    sg = part["sphere_global"]
    ss = t == 1
    gA = sg[a]
    gB = np.where(ss, sg[np.where(ss, bb, 0)], bb.astype(np.int64))
    lo, hi = np.where(ss & (gA > gB), gB, gA), np.where(ss & (gA > gB), gA, gB)
    return np.stack([lo, hi, t.astype(np.int64)], 1)


@pytest.mark.parametrize("n_slabs,mode,once,lead", [(2, "exact", False, 0), (8, "exact", False, 0), (8, "fast", False, 0), (2, "exact", True, 0),
                                                     (8, "fast", True, 0), (2, "exact", False, 10), (8, "fast", False, 10)])
def test_million_clumps_in_slabs_against_single_domain_oracle(pkg, orc, packed_million, n_slabs, mode, once, lead):
    """STATED TOLERANCE: contact sets identical (every pair of the single-domain oracle list is found by the slab that owns
    either clump, and nothing else is); after 100 steps (h = 5e-6 s, detection every 40 with the bench's margins) positions
    within 5e-9 m and velocities within 5e-6 m/s of the oracle's single-domain run.  The slabs number their clumps locally, so
    an owner's contributions are summed in another order than in the single domain: fp32 rounding, then the bed's own
    sensitivity -- not an error of the exchange (the small test above is bit-identical to the ordered exchange).
    The `fast` leg is the library's default arithmetic (what bench.py --gpus N times: owner-tile force pass, split into the
    interior / ghost-dependent passes of the halo overlap): same contact sets, the fast mode's own bounds of
    tests/test_fast_mode.py (5e-8 m, 2e-4 m/s after 100 steps).
    The `once` legs: deme_halo_group_set_cross_contacts -- every contact that straddles a cut is evaluated by ONE slab (the left
    one), which returns the reaction every step: the slabs' lists put together ARE the single-domain list, row for row, no pair
    twice; same bounds.
    The legs with a `lead`: deme_set_async_detection in every slab -- the lists that retire at steps 41 and 81 are built beside the ten
    steps before, from owner snapshots taken inside the halo loop, with margins for K + D steps; same bounds against the oracle's
    lock-step run."""
    b, p, sc, st, cnt, W = packed_million
    if lead:
        # The bench's margins (1.2 x own speed + 0.02 m/s, the reference demos' knobs) are an approximation already for K steps (DESIGN
        # 6a: ~1 % of the clumps differ by 1e-5 m from a run that detects every step); a list that serves from D steps after its
        # snapshot to K + D steps after it leans on them harder.  With a safety velocity that covers the bed's speed changes the
        # asynchronous run IS the lock-step one, which is what this leg pins (slabs and oracle alike).
        p = type(p).from_buffer_copy(p)
        p.expSafetyAdder = 0.5
    nc = int(sc.nOwnerClumps)
    g_arrays = dict(b.arrays)
    for k in GKEYS:
        g_arrays[k] = np.asarray(st[k]).copy()
    X0 = pkg.model.decode_positions(st["voxelID"], st["locX"], st["locY"], st["locZ"], p.nvXp2, p.nvYp2, p.voxelSize, p.l)
    x_now = X0[:nc, 0] + p.LBFX
    # the single domain as a one-slab decomposition gives the payload format the re-decomposition takes: state + history in
    # global ids; cutting it into n slabs hands every slab its clumps, its ghosts and its share of the contact history
    one = pkg.decomp.decompose(g_arrays, b.counts, x_now, 1, halo=0.03)[0]
    payload = pkg.decomp.owned_payload(one, st, cnt, W)
    halo = 0.03
    _, parts, seeds = pkg.decomp.redecompose(g_arrays, b.counts, [payload], n_slabs, halo,
                                             lambda arr: pkg.model.decode_positions(arr["voxelID"], arr["locX"], arr["locY"], arr["locZ"], p.nvXp2,
                                                                                     p.nvYp2, p.voxelSize, p.l)[:, 0] + p.LBFX)
    ctxs = [_make(pkg, p, pt["scene"], mode) for pt in parts]
    for c, sd in zip(ctxs, seeds):
        c.seed_contacts(*sd)
    grp = _group(pkg, ctxs, parts)
    if once:
        grp.set_cross_contacts(True)
    if lead:
        for c in ctxs:
            c.set_timing(1)
            c.set_async_detection(lead)
    sim = orc.make_sim(pkg, p, sc)
    sim.upload_state({k: st[k] for k in GKEYS})
    sim.seed_contacts(cnt[0], cnt[1], cnt[2], W)
    orc.set_num_threads(min(64, os.cpu_count() or 1))
    try:
        grp.step(1), sim.step(1)  # the first step detects: the lists of this state
        grp.sync()
        ref = sim.contacts()
        ref_rows = np.stack([ref[0].astype(np.int64), ref[1].astype(np.int64), ref[2].astype(np.int64)], 1)
        rows_all = np.concatenate([_global_pairs(pt, c) for pt, c in zip(parts, ctxs)])
        rows = np.unique(rows_all, axis=0)
        if once:  # no pair on two lists (and no ghost-wall pair on the list of a slab that does not own the clump)
            assert len(rows_all) == len(ref_rows) == len(rows), (len(rows_all), len(ref_rows), len(rows))
        # a slab also lists ghost-wall pairs of its ghosts?  No: a ghost's sphere-analytical contacts belong to its owner rank, but
        # the slab evaluates them too (forces on ghosts are discarded) -- as a set the union is still the single-domain list
        ref_sorted = np.unique(ref_rows, axis=0)
        assert rows.shape == ref_sorted.shape and np.array_equal(rows, ref_sorted), (rows.shape, ref_sorted.shape)
        n_cross = sum(int(((pt["arrays"]["ownerClumpBody"][c.contacts()[1][c.contacts()[2] == 1]] >= pt["n_own"])).sum()) for pt, c in zip(parts, ctxs))
        assert n_cross > 1000 * (n_slabs - 1)  # thousands of contacts straddle every cut
        N = 100
        grp.step(N - 1), sim.step(N - 1)
        grp.sync()
        if mode == "fast":  # the kernel bench.py --gpus N times, in every slab
            for c in ctxs:
                assert c.force_kernel()[0] == "k_tile_forces<0, false>", c.force_kernel()
        if lead:  # the lists of steps 41 and 81 came from the asynchronous cycle, in every slab
            for c in ctxs:
                assert c.kernel_time_ms("detect_async_part1")[1] == 2, c.kernel_time_ms("detect_async_part1")
    finally:
        orc.set_num_threads(min(8, os.cpu_count() or 1))
    X, V = gather_positions(pkg, parts, ctxs, p, nc)
    so = sim.download_state()
    Xo = pkg.model.decode_positions(so["voxelID"], so["locX"], so["locY"], so["locZ"], p.nvXp2, p.nvYp2, p.voxelSize, p.l)[:nc]
    Vo = np.stack([so["vX"], so["vY"], so["vZ"]], 1)[:nc]
    dx, dv = np.abs(X - Xo).max(), np.abs(V - Vo).max()
    print(f"configs[2], {n_slabs} slabs x {[pt['n_own'] for pt in parts][:3]}... clumps, {len(ref[0])} contacts, {n_cross} across cuts: "
          f"|dx| {dx:.3e} m, |dv| {dv:.3e} m/s after {N} steps vs the single-domain oracle")
    assert (dx <= 5e-9 and dv <= 5e-6) if mode == "exact" else (dx <= 5e-8 and dv <= 2e-4)
    grp.close()
    for c in ctxs:
        c.close()


def test_config5_cohesive_bed_in_four_slabs_against_single_domain_oracle(pkg, orc):
    """BASELINE configs[4] as it is stated -- a polydisperse bed with a run-time compiled cohesion model on FOUR devices -- at
    test size on the one GPU there is: bench.build_config5 (1e5 spheres of 8 radii, the cohesion + contact-age fragment through
    hipRTC) settled in a single context, cut into 4 x-slabs with its contact history (the age wildcard travels), stepped 60
    times through the library's halo loop, against the ORACLE's single-domain run from the same state.  STATED TOLERANCE: the
    union of the slabs' lists equals the oracle's list; positions within 1e-8 m and velocities within 2e-5 m/s (the bounds of
    tests/test_force_hook.py's single-domain leg: the fragment's log / sqrt are the device's on one side and libm's on the
    other, and the slabs sum an owner's contributions in their own local order)."""
    import bench
    n = 100_000
    b = bench.build_config5(pkg, n, seed=2024, cd_freq=20)
    p, sc = b.Initialize()
    ctx = _make(pkg, p, sc)
    b.compile_into(ctx)
    for _ in range(20):
        ctx.step(3000)
        if int(ctx.counts().nContacts) > 1.5 * n:
            break
    st = ctx.download_state()
    cnt = ctx.contacts()
    assert len(cnt[0]) > 1.5 * n
    W = ctx.wildcard(0).reshape(-1, 1)
    ctx.close()
    nc = int(sc.nOwnerClumps)
    g_arrays = dict(b.arrays)
    for k in GKEYS:
        g_arrays[k] = np.asarray(st[k]).copy()
    decode_x = lambda arr: pkg.model.decode_positions(arr["voxelID"], arr["locX"], arr["locY"], arr["locZ"], p.nvXp2, p.nvYp2,
                                                       p.voxelSize, p.l)[:, 0] + p.LBFX
    halo = 0.03
    one = pkg.decomp.decompose(g_arrays, b.counts, decode_x(g_arrays)[:nc], 1, halo=halo)[0]
    payload = pkg.decomp.owned_payload(one, st, cnt, W, flip_sign_wildcards=())  # (the age is a scalar: no sign to flip)
    _, parts, seeds = pkg.decomp.redecompose(g_arrays, b.counts, [payload], 4, halo, decode_x, flip_sign_wildcards=())
    ctxs = []
    for pt, sd in zip(parts, seeds):
        c = _make(pkg, p, pt["scene"])
        b.compile_into(c)
        c.seed_contacts(*sd)
        ctxs.append(c)
    grp = _group(pkg, ctxs, parts)
    sim = orc.make_sim(pkg, p, sc)
    sim.set_custom_model(1, np.full((int(sc.nMat), int(sc.nMat)), 0.002, np.float32))
    sim.upload_state({k: st[k] for k in GKEYS})
    sim.seed_contacts(cnt[0], cnt[1], cnt[2], W)
    orc.set_num_threads(min(32, os.cpu_count() or 1))
    try:
        N = 60
        grp.step(N), sim.step(N)
        grp.sync()
    finally:
        orc.set_num_threads(min(8, os.cpu_count() or 1))
    ref = sim.contacts()
    ref_rows = np.unique(np.stack([ref[0].astype(np.int64), ref[1].astype(np.int64), ref[2].astype(np.int64)], 1), axis=0)
    rows = np.unique(np.concatenate([_global_pairs(pt, c) for pt, c in zip(parts, ctxs)]), axis=0)
    assert rows.shape == ref_rows.shape and np.array_equal(rows, ref_rows), (rows.shape, ref_rows.shape)
    X, V = gather_positions(pkg, parts, ctxs, p, nc)
    so = sim.download_state()
    Xo = pkg.model.decode_positions(so["voxelID"], so["locX"], so["locY"], so["locZ"], p.nvXp2, p.nvYp2, p.voxelSize, p.l)[:nc]
    Vo = np.stack([so["vX"], so["vY"], so["vZ"]], 1)[:nc]
    dx, dv = np.abs(X - Xo).max(), np.abs(V - Vo).max()
    print(f"configs[4] in 4 slabs: {len(ref[0])} contacts, |dx| {dx:.3e} m, |dv| {dv:.3e} m/s after {N} steps vs the single-domain oracle")
    assert dx <= 1e-8 and dv <= 2e-5
    grp.close()
    for c in ctxs:
        c.close()


def test_ten_million_clumps_in_eight_slabs_properties(pkg):
    """BASELINE configs[2] at ITS size: 1e7 three-sphere clumps cut into 8 x-slabs (one GPU holds them all: ~30 GB of the 288),
    the library's default arithmetic.  No oracle at this size -- properties that do not depend on one: the union of the eight
    slabs' contact lists, in global ids, is the single-context list of the same state pair for pair (more than 1e7 contacts); detection is
    idempotent in every slab; the accelerations obey Newton's third law over the whole bed (every force appears twice with
    opposite signs: the mass-weighted sum over clumps and walls vanishes to rounding)."""
    import bench
    n = 10_000_000
    b = bench.build_bed(pkg, n, 2024, 40, order="morton")
    p, sc = b.Initialize()
    ctx = _make(pkg, p, sc, "fast")
    ctx.step(24000)  # (the bed is 3.2 times as deep as the 1e6 one: it is still closing up, which is all the better for a list test)
    st = ctx.download_state()
    nc = int(sc.nOwnerClumps)
    g_arrays = dict(b.arrays)
    for k in GKEYS:
        g_arrays[k] = np.asarray(st[k]).copy()
    X0 = pkg.model.decode_positions(st["voxelID"], st["locX"], st["locY"], st["locZ"], p.nvXp2, p.nvYp2, p.voxelSize, p.l)
    parts = pkg.decomp.decompose(g_arrays, b.counts, X0[:nc, 0] + p.LBFX, 8, halo=0.03)
    del X0
    ctx.compute_margins(0), ctx.detect(), ctx.migrate()
    a, bb, t, _ = ctx.contacts()
    assert len(a) > 10_000_000
    pack = lambda lo, hi, ty: (lo.astype(np.uint64) << np.uint64(36)) | (ty.astype(np.uint64) << np.uint64(32)) | hi.astype(np.uint64)
    ref = np.sort(pack(a, bb, t))
    # Newton's third law, single context
    ctx.calc_forces()
    assert ctx.force_kernel()[0] == "k_tile_forces<0, false>", ctx.force_kernel()
    s1 = ctx.download_state()
    mass = b.arrays["MassProperties"][b.arrays["inertiaPropOffsets"]].astype(np.float64)
    acc = np.stack([s1["aX"], s1["aY"], s1["aZ"]], 1).astype(np.float64)
    total = (mass[:, None] * acc).sum(0)
    scale = (mass[:, None] * np.abs(acc)).sum()
    assert np.abs(total).max() < 1e-5 * scale, (total, scale)
    ctx.close()
    del a, bb, t, s1, acc
    keys = []
    for pt in parts:
        c = _make(pkg, p, pt["scene"], "fast")
        c.compute_margins(0), c.detect(), c.migrate()
        la, lb, lt, _ = c.contacts()
        c.detect()
        la2, lb2, lt2, m2 = c.contacts()
        assert np.array_equal(la, la2) and np.array_equal(lb, lb2) and np.array_equal(lt, lt2)
        assert np.array_equal(m2, np.arange(len(la), dtype=np.uint32))  # idempotent: every contact maps onto itself
        sg = pt["sphere_global"]
        ss = lt == 1
        gA = sg[la]
        gB = np.where(ss, sg[np.where(ss, lb, 0)], lb.astype(np.int64))
        flip = ss & (gA > gB)
        keys.append(pack(np.where(flip, gB, gA), np.where(flip, gA, gB), lt))
        c.close()
    union = np.unique(np.concatenate(keys))
    assert union.shape == ref.shape and np.array_equal(union, ref), (union.shape, ref.shape)


@pytest.mark.parametrize("n_slabs,mode", [(8, "fast"), (2, "exact")])
def test_million_clumps_through_deme_multi_against_single_domain_oracle(pkg, orc, packed_million, n_slabs, mode):
    """configs[2] through the C++ decomposition (csrc/deme_decomp.inc: what DEMSolver(nGPUs) runs): the settled 1e6-clump bed handed
    to deme_multi_build, which cuts it into bin-aligned slabs along its longest side, numbers every slab in the engine's order and
    steps them with the ghost exchange overlapped -- against the ORACLE's single-domain run from the same state (no history on
    either side).  The merged contact list in global ids IS the oracle's list, row for row; after 100 steps (detection every 40, the
    bench's margins) the state gathered by global id is within the bounds of the hand-attached slabs above; in the fast mode every
    slab's list goes through the owner-tile pass with no tile through the fallback."""
    b, p, sc, st, cnt, W = packed_million
    nc = int(sc.nOwnerClumps)
    g_arrays = dict(b.arrays)
    for k in GKEYS:
        g_arrays[k] = np.asarray(st[k]).copy()
    sc2 = pkg.abi.make_scene_struct(g_arrays, b.counts)
    m = pkg.abi.Multi(devices=(0,))
    m.build(p, sc2, slabs_per_device=n_slabs, axis=-1, halo=0.03, arith=mode)
    sim = orc.make_sim(pkg, p, sc2)
    orc.set_num_threads(min(64, os.cpu_count() or 1))
    try:
        m.step(1), sim.step(1)
        ga, gb, gt = m.contacts()
        oa, ob, ot, _ = sim.contacts()
        assert len(oa) > 3_000_000 and np.array_equal(ga, oa) and np.array_equal(gb, ob) and np.array_equal(gt, ot)
        m.step(99), sim.step(99)
        m.sync()
    finally:
        orc.set_num_threads(min(8, os.cpu_count() or 1))
    if mode == "fast":
        for s in range(n_slabs):
            c = m.slab_ctx(s)
            assert c.force_kernel()[0] == "k_tile_forces<0, false>" and c.tile_stats()[1] == 0, (s, c.force_kernel(), c.tile_stats())
    g, o = m.download_state(), sim.download_state()
    X = pkg.model.decode_positions(g["voxelID"], g["locX"], g["locY"], g["locZ"], p.nvXp2, p.nvYp2, p.voxelSize, p.l)[:nc]
    Xo = pkg.model.decode_positions(o["voxelID"], o["locX"], o["locY"], o["locZ"], p.nvXp2, p.nvYp2, p.voxelSize, p.l)[:nc]
    dx = float(np.abs(X - Xo).max())
    dv = max(float(np.abs(g[k][:nc] - o[k][:nc]).max()) for k in ("vX", "vY", "vZ"))
    print(f"configs[2] through deme_multi, {n_slabs} slabs ({mode}): {len(oa)} contacts, |dx| {dx:.3e} m, |dv| {dv:.3e} m/s after 100 steps vs the single-domain oracle")
    from tests.conftest import record_measured
    record_measured(f"test_config2_slabs deme_multi {n_slabs} slabs {mode} 1e6 clumps 100 steps", dx_m=dx, dv_m_per_s=dv)
    # (exact mode: a slab numbers its clumps in the engine's order, so an owner's contributions add up in another order than in the
    # single domain -- fp32 rounding, measured 1.5e-9 m / 1.2e-5 m/s on this bed restarted without its friction history; fast mode:
    # the bounds of tests/test_fast_mode.py, measured 9.8e-10 m / 2.6e-5 m/s)
    assert (dx <= 5e-9 and dv <= 5e-5) if mode == "exact" else (dx <= 5e-8 and dv <= 2e-4)
    m.close()


@pytest.mark.gpu
def test_library_migration_carries_user_wildcard_arrays(pkg):
    """Owner and sphere wildcard arrays of a run-time compiled model across deme_halo_group_migrate: values tagged with the GLOBAL
    ids before the clumps cross the cuts sit under the re-assembled slabs' numbering afterwards -- own clumps, ghost copies (their
    values come from the slab that owns them) -- and a replicated owner keeps the slab's own value.  (The numpy statement of the same
    rule: tests/test_decomp.py::test_redecomposition_carries_wildcard_arrays_and_persistent_marks.)"""
    from tests.test_force_hook import PLAIN
    b, p, sc, x = _sheared_bed(pkg, 3000, 6)
    halo = 0.035
    parts = pkg.decomp.decompose(b.arrays, b.counts, x, 3, halo=halo)
    names = [f"unused_{k}" for k in range(int(p.nContactWildcards))]
    ctxs = []
    for pt in parts:
        c = _make(pkg, p, pt["scene"])
        c.compile_force_model(PLAIN, names, "", owner_wildcards=("tag", "twice"), geo_wildcards=("stag",))
        n_c, n_o = int(pt["counts"]["nOwnerClumps"]), int(pt["counts"]["nOwners"])
        og, sg = np.asarray(pt["owner_global"], np.float64), np.asarray(pt["sphere_global"], np.float64)
        c.set_wildcard_array("owner", 0, np.r_[og[:n_c] + 0.5, np.full(n_o - n_c, -7.0)])
        c.set_wildcard_array("owner", 1, np.r_[2 * og[:n_c], np.full(n_o - n_c, -9.0)])
        c.set_wildcard_array("sphere", 0, sg + 0.25)
        ctxs.append(c)
    g = _group(pkg, ctxs, parts)
    g.step(150)
    g.sync()
    for c, pt in zip(ctxs, parts):
        g.set_slab(c, pt, halo)
    g._ctxs = ctxs
    moved = g.migrate()
    assert moved >= 3
    changed = 0
    for c, pt in zip(ctxs, parts):
        n_own, n_gl, n_gr, n_o, n_s, _ = g.slab_counts(c)
        og, sg, _, _ = g.slab_ids(c)
        n_c = n_own + n_gl + n_gr
        changed += int(n_o != int(pt["counts"]["nOwners"]) or not np.array_equal(og[:n_c], np.asarray(pt["owner_global"][:n_c], np.uint32)))
        tag, twice, stag = c.wildcard_array("owner", 0, n_o), c.wildcard_array("owner", 1, n_o), c.wildcard_array("sphere", 0, n_s)
        assert np.array_equal(tag[:n_c], (og[:n_c] + 0.5).astype(np.float32)) and (tag[n_c:] == -7.0).all()
        assert np.array_equal(twice[:n_c], (2.0 * og[:n_c]).astype(np.float32)) and (twice[n_c:] == -9.0).all()
        assert np.array_equal(stag, (sg + 0.25).astype(np.float32))
    assert changed >= 2  # (the slabs really were re-assembled)
    g.step(20), g.sync()  # the model runs on with the arrays of the new sizes
    for c in ctxs:
        assert np.isfinite(c.download_state()["vX"]).all()
    g.close()


@pytest.mark.gpu
def test_million_clumps_migrate_and_rebalance_inside_deme_multi(pkg, packed_million, monkeypatch):
    """The settled 1e6-clump bed drifting at 1 m/s along x and y through deme_multi in 4 slabs, fast mode: migrations every 50 steps inside
    deme_multi_step, every second one recomputing the slab boundaries (deme_multi_rebalance), the library checking the re-assembled
    books after each (DEME_MIG_CHECK) -- against the SAME bed stepped by one context (the comparison at this size against the oracle
    is the test above; here the subject is what migration and moving boundaries add): every clump owned once, thousands changed
    slabs, every slab still through the owner-tile pass, the state by global id on the single context's."""
    monkeypatch.setenv("DEME_MIG_CHECK", "1")
    b, p, sc, st, cnt, W = packed_million
    nc = int(sc.nOwnerClumps)
    g_arrays = dict(b.arrays)
    for k in GKEYS:
        g_arrays[k] = np.asarray(st[k]).copy()
    for k in ("vX", "vY"):  # (the cut runs along the longer side of the bed: x or y)
        g_arrays[k] = g_arrays[k].copy()
        g_arrays[k][:nc] += 1.0
    sc2 = pkg.abi.make_scene_struct(g_arrays, b.counts)
    m = pkg.abi.Multi(devices=(0,))
    m.build(p, sc2, slabs_per_device=4, axis=-1, halo=0.03, arith="fast")
    m.set_migration(50), m.set_rebalance(2)
    twin = pkg.abi.Multi(devices=(0,))  # the same slabs without migration: the 3 cm halo covers 1.5 mm of drift
    twin.build(p, sc2, slabs_per_device=4, axis=-1, halo=0.03, arith="fast")
    one = _make(pkg, p, sc2, "fast")
    own0 = [m.slab_counts(s)[0][0] for s in range(4)]
    e0 = [m.slab_counts(s)[1] for s in range(4)]
    m.step(301), twin.step(301), one.step(301)  # (six migrations; one more step: every slab has detected and evaluated its re-assembled list)
    m.sync(), twin.sync()
    own1 = [m.slab_counts(s)[0][0] for s in range(4)]
    e1 = [m.slab_counts(s)[1] for s in range(4)]
    _, moved = m.counts()
    assert sum(own0) == sum(own1) == nc and moved > 1000, (own0, own1, moved)
    # (every second migration recomputed the boundaries: on this bed 1.5 mm of drift leaves them on the same bin faces; the 3 m/s
    # variant of this run moved the first one by a bin -- tests/test_multi.py moves them on a smaller bed)
    for s in range(4):
        c = m.slab_ctx(s)
        assert c.force_kernel()[0] == "k_tile_forces<0, false>", (s, c.force_kernel())
    big = [m.slab_ctx(s).tile_stats() for s in range(4)]
    g, o = m.download_state(), one.download_state()
    X = pkg.model.decode_positions(g["voxelID"], g["locX"], g["locY"], g["locZ"], p.nvXp2, p.nvYp2, p.voxelSize, p.l)[:nc]
    Xo = pkg.model.decode_positions(o["voxelID"], o["locX"], o["locY"], o["locZ"], p.nvXp2, p.nvYp2, p.voxelSize, p.l)[:nc]
    dx = float(np.abs(X - Xo).max())
    dv = max(float(np.abs(g[k][:nc] - o[k][:nc]).max()) for k in ("vX", "vY", "vZ"))
    tw = twin.download_state()
    Xt = pkg.model.decode_positions(tw["voxelID"], tw["locX"], tw["locY"], tw["locZ"], p.nvXp2, p.nvYp2, p.voxelSize, p.l)[:nc]
    dx_t, dx_to = float(np.abs(X - Xt).max()), float(np.abs(Xt - Xo).max())
    print(f"against the twin slabs that never migrated: |dx| {dx_t:.3e} m (the twin against one context: {dx_to:.3e} m)")
    print(f"tile statistics per slab after the migrations (tiles, tiles through the fallback, ...): {big}")
    print(f"1e6 clumps drifting through 4 slabs: own {own0} -> {own1}, {moved} clumps changed slabs, boundaries {[round(e[1], 4) for e in e0[:-1]]} -> "
          f"{[round(e[1], 4) for e in e1[:-1]]}; |dx| {dx:.3e} m, |dv| {dv:.3e} m/s vs one context after 301 steps")
    from tests.conftest import record_measured
    record_measured("test_config2_slabs deme_multi 4 slabs fast 1e6 clumps drifting 300 steps migrate + rebalance", dx_m=dx, dv_m_per_s=dv, clumps_moved=moved, dx_vs_twin_without_migration_m=dx_t, twin_vs_one_context_m=dx_to)
    # (the drifting bed presses into two walls: the slabs' own summation order grows to dx_to against one context with or without
    # migration; what migration and moving boundaries may add is held to the same size)
    assert dx_t <= max(3 * dx_to, 1e-8) and dx <= max(3 * dx_to, 1e-8)
    m.close(), twin.close()
