"""GPU parity tests (run with -m gpu on an MI355X): every stage of the HIP hot path, called through
the C-ABI, against the CPU oracle on the same seeded inputs.

Bars (BASELINE.json north_star): contact sets, bin assignments, history maps and the position codec
are integer work -> bit-exact.  Forces, accelerations and whole trajectories are fp32 physics over fp64 geometry; the
accumulation is atomics-free with a fixed summation order that the oracle shares, so they are compared for EQUALITY too.
The exceptions carry their tolerance in the test: owners reduced by the workgroup tree (more than 256 contacts) and sums
compared with numpy.
"""
import numpy as np
import pytest


pytestmark = pytest.mark.gpu

NULL = 0xFFFFFFFF


STATE_KEYS = ("voxelID", "locX", "locY", "locZ", "oriQw", "oriQx", "oriQy", "oriQz", "vX", "vY", "vZ", "omgBarX", "omgBarY",
              "omgBarZ")


def pair(pkg, orc, builder):
    p, sc = builder.Initialize()
    ctx = pkg.Context(0)
    ctx.set_params(p)
    ctx.upload_scene(sc)
    sim = orc.make_sim(pkg, p, sc)
    return ctx, sim, p, sc


def positions(pkg, st, p):
    return pkg.model.decode_positions(st["voxelID"], st["locX"], st["locY"], st["locZ"], p.nvXp2, p.nvYp2, p.voxelSize, p.l)


def assert_same_contacts(ctx, sim):
    a, b, t, m = ctx.contacts()
    oa, ob, ot, om = sim.contacts()
    assert len(a) == len(oa), f"contact count {len(a)} vs oracle {len(oa)}"
    assert (a == oa).all() and (b == ob).all() and (t == ot).all()
    return a, b, t, m, om


def test_geometry_and_bin_assignment_bit_exact(pkg, orc):
    b = pkg.model.packed_bed(3000, seed=11, cd_freq=0, spacing_mult=2.6)
    ctx, sim, p, sc = pair(pkg, orc, b)
    for drift in (0, 25):  # zero margin and a velocity-based margin
        ctx.compute_margins(drift)
        sim.compute_margins(drift)
        ctx.detect()
        sim.detect()
        for g, o in zip(ctx.sphere_geometry(), sim.sphere_geometry()):
            assert (g == o).all()
        gb, gs = ctx.bin_incidence()
        ob, os_ = sim.bin_incidence()
        assert len(gb) == len(ob) and (gb == ob).all() and (gs == os_).all()
        cg, co = ctx.counts(), sim.counts()
        assert cg.nActiveBins == co.nActiveBins and cg.maxSpheresInBin == co.maxSpheresInBin
        assert_same_contacts(ctx, sim)
    # sortedness + uniqueness of the canonical list
    a, bb, t, _ = ctx.contacts()
    ss = t == 1
    key = a[ss].astype(np.uint64) << np.uint64(32) | bb[ss].astype(np.uint64)
    assert (np.diff(key.astype(np.int64)) > 0).all() and (a[ss] < bb[ss]).all()


def test_margin_inflated_detection_with_velocities(pkg, orc):
    b = pkg.model.packed_bed(2500, seed=5, cd_freq=10, spacing_mult=3.0, init_vz=-0.8)
    b.SetExpandSafetyAdder(0.5)
    ctx, sim, p, sc = pair(pkg, orc, b)
    ctx.compute_margins(10)
    sim.compute_margins(10)
    ctx.detect(), sim.detect()
    a, *_ = assert_same_contacts(ctx, sim)
    assert len(a) > 0


def test_history_map_and_wildcard_migration(pkg, orc):
    b = pkg.model.packed_bed(2000, seed=3, cd_freq=0, spacing_mult=2.5, init_vz=-0.3)
    ctx, sim, p, sc = pair(pkg, orc, b)
    ctx.step(3), sim.step(3)
    a, bb, t, m, om = assert_same_contacts(ctx, sim)
    assert (m == om).all()
    assert (m != NULL).sum() > 0.5 * len(m)  # most contacts persist between consecutive detections
    # mapped contacts point at the same (A, B, type) in the previous list: check through a 2nd detection
    prev = set(zip(a.tolist(), bb.tolist(), t.tolist()))
    ctx.step(1), sim.step(1)
    a2, b2, t2, m2, om2 = assert_same_contacts(ctx, sim)
    assert (m2 == om2).all()
    for i in np.nonzero(m2 != NULL)[0][:2000]:
        assert (int(a2[i]), int(b2[i]), int(t2[i])) == (int(a[m2[i]]), int(bb[m2[i]]), int(t[m2[i]]))
    for i in np.nonzero(m2 == NULL)[0][:2000]:
        assert (int(a2[i]), int(b2[i]), int(t2[i])) not in prev


@pytest.mark.parametrize("model", [0, 1])
def test_forces_against_oracle(pkg, orc, model):
    b = pkg.model.packed_bed(3000, seed=21, cd_freq=0, spacing_mult=2.5, init_vz=-0.4, force_model=model, Crr=0.0)
    ctx, sim, p, sc = pair(pkg, orc, b)
    ctx.set_record_contacts(True)
    # a few steps so contact history and angular velocities are non-trivial, then one staged force pass
    ctx.step(5), sim.step(5)
    st = sim.download_state()
    ctx.upload_state({k: st[k] for k in st if k not in ("aX", "aY", "aZ", "alphaX", "alphaY", "alphaZ")})
    ctx.compute_margins(0), sim.compute_margins(0)
    ctx.detect(), sim.detect()
    ctx.migrate(), sim.migrate()
    nC = assert_same_contacts(ctx, sim)[0].size
    if model == 0:
        for w in range(4):  # identical history going in
            ctx.set_wildcard(w, sim.wildcard(w))
    ctx.calc_forces()
    sim.calc_forces(record=True)
    F, T, PA, PB = ctx.contact_records()
    oF, oT, oPA, oPB = sim.contact_records()
    scale = np.abs(oF).max()
    assert scale > 0
    # per-contact quantities: the same IEEE arithmetic on both sides -> bit-identical forces, contact points and history
    assert np.array_equal(F, oF) and np.array_equal(T, oT)
    assert np.array_equal(PA, oPA) and np.array_equal(PB, oPB)
    if model == 0:
        for w in range(4):
            assert np.array_equal(ctx.wildcard(w), sim.wildcard(w))
    gs, os_ = ctx.download_state(), sim.download_state()
    for k in ("aX", "aY", "aZ", "alphaX", "alphaY", "alphaZ"):
        # clumps: contributions summed in the same fixed order (A-side run, then B-side list) -> bit-identical
        assert np.array_equal(gs[k][:-1], os_[k][:-1]), k
        # the wall owner sums thousands of terms with a tree on the device: tolerance
        assert abs(gs[k][-1] - os_[k][-1]) <= 1e-3 * max(abs(os_[k][-1]), 1e-6), k
    assert nC > 1000


@pytest.mark.parametrize("integrator", [0, 1, 2])
def test_integrator_bit_exact_given_same_accelerations(pkg, orc, integrator):
    b = pkg.model.packed_bed(1500, seed=8, cd_freq=0, spacing_mult=2.6, init_vz=-0.2)
    b.SetIntegrator(integrator)
    ctx, sim, p, sc = pair(pkg, orc, b)
    rng = np.random.default_rng(4)
    n = sc.nOwners
    st = sim.download_state()
    for k in ("aX", "aY", "aZ"):
        st[k] = rng.uniform(-50, 50, n).astype(np.float32)
    for k in ("alphaX", "alphaY", "alphaZ"):
        st[k] = rng.uniform(-500, 500, n).astype(np.float32)
    for k in ("omgBarX", "omgBarY", "omgBarZ"):
        st[k] = rng.uniform(-20, 20, n).astype(np.float32)
    sim.upload_state(st), ctx.upload_state(st)
    for _ in range(3):
        ctx.integrate(), sim.integrate()
        sim.upload_state({k: st[k] for k in ("aX", "aY", "aZ", "alphaX", "alphaY", "alphaZ")})
        ctx.upload_state({k: st[k] for k in ("aX", "aY", "aZ", "alphaX", "alphaY", "alphaZ")})
    gs, os_ = ctx.download_state(), sim.download_state()
    for k in ("voxelID", "locX", "locY", "locZ", "oriQw", "oriQx", "oriQy", "oriQz", "vX", "vY", "vZ", "omgBarX",
              "omgBarY", "omgBarZ"):
        assert (gs[k] == os_[k]).all(), k
    # the fixed wall owner did not move
    assert gs["vZ"][-1] == 0 and gs["voxelID"][-1] == st["voxelID"][-1]


def test_trajectory_parity_mode(pkg, orc):
    """End-to-end (SURVEY G10 analogue): 1000 clumps, detection every step, 200 steps.  The HIP path and the oracle run
    independently and stay BIT-IDENTICAL: same contact lists, same history, same owner state (the per-contact arithmetic is
    the same IEEE sequence and both sum an owner's contributions in the same order: A-side run, then B-side list)."""
    b = pkg.model.packed_bed(1000, seed=2, cd_freq=0, spacing_mult=2.7, init_vz=-0.5)
    ctx, sim, p, sc = pair(pkg, orc, b)
    for chunk in range(4):
        ctx.step(50), sim.step(50)
        assert ctx.counts().nContacts == sim.counts().nContacts
        assert_same_contacts(ctx, sim)
        gs, os_ = ctx.download_state(), sim.download_state()
        for k in STATE_KEYS:
            assert np.array_equal(gs[k], os_[k]), (chunk, k)
        for w in range(4):
            assert np.array_equal(ctx.wildcard(w), sim.wildcard(w)), (chunk, w)
    assert ctx.counts().nContacts > 100


def test_trajectory_throughput_mode(pkg, orc):
    """Detection every 10 steps with the velocity margin: list built at step n serves n..n+9."""
    b = pkg.model.packed_bed(1000, seed=9, cd_freq=10, spacing_mult=2.8, init_vz=-0.6)
    b.SetExpandSafetyAdder(0.2)
    ctx, sim, p, sc = pair(pkg, orc, b)
    ctx.step(100), sim.step(100)
    assert ctx.counts().nDetections == sim.counts().nDetections == 10
    assert ctx.counts().nContacts == sim.counts().nContacts
    gs, os_ = ctx.download_state(), sim.download_state()
    for k in STATE_KEYS:  # bit-identical also with the K-step (drift) policy
        assert np.array_equal(gs[k], os_[k]), k


def test_family_masks_and_fixed_family(pkg, orc):
    b = pkg.model.packed_bed(1500, seed=13, cd_freq=0, spacing_mult=2.5, init_vz=-0.3)
    fam = (np.arange(len(b.batches[0].xyz)) % 3).astype(np.uint8)
    b.batches[0].SetFamily(fam)
    b.DisableContactBetweenFamilies(1, 2)
    b.SetFamilyFixed(2)
    b.SetFamilyExtraMargin(0, 2e-4)
    ctx, sim, p, sc = pair(pkg, orc, b)
    ctx.step(20), sim.step(20)
    a, bb, t, *_ = assert_same_contacts(ctx, sim)
    own = b.arrays["ownerClumpBody"]
    ss = t == 1
    fa, fb = fam[own[a[ss]]], fam[own[bb[ss]]]
    assert not (((fa == 1) & (fb == 2)) | ((fa == 2) & (fb == 1))).any()
    gs = ctx.download_state()
    fixed = np.nonzero(fam == 2)[0]
    assert (gs["vZ"][fixed] == 0).all()


def test_crowded_bins_span_chunks(pkg, orc):
    """Bins holding more spheres than one 256-entry sweep chunk exercise the global look-back path."""
    b = pkg.model.packed_bed(1200, seed=17, cd_freq=0, spacing_mult=2.6)
    b.SetInitBinSize(0.12)  # ~8 clump spacings per bin edge => hundreds of spheres per bin
    ctx, sim, p, sc = pair(pkg, orc, b)
    ctx.detect(), sim.detect()
    assert ctx.counts().maxSpheresInBin == sim.counts().maxSpheresInBin > 256
    assert_same_contacts(ctx, sim)


def test_empty_and_single(pkg, orc):
    # no clumps at all: only the wall owner
    b = pkg.SceneBuilder()
    m = b.LoadMaterial({"E": 1e8, "nu": 0.3, "CoR": 0.5, "mu": 0.3, "Crr": 0.0})
    b.InstructBoxDomainDimension(1, 1, 1)
    b.InstructBoxDomainBoundingBC("all", m)
    b.LoadSphereType(1e-3, 0.01, m)
    b.SetCDUpdateFreq(0)
    ctx, sim, p, sc = pair(pkg, orc, b)
    ctx.step(3), sim.step(3)
    assert ctx.counts().nContacts == 0 and ctx.counts().nBinSphereTouches == 0
    # one sphere resting into the floor
    b = pkg.SceneBuilder()
    m = b.LoadMaterial({"E": 1e8, "nu": 0.3, "CoR": 0.5, "mu": 0.3, "Crr": 0.0})
    b.InstructBoxDomainDimension(1, 1, 1)
    b.InstructBoxDomainBoundingBC("all", m)
    t = b.LoadSphereType(1e-3, 0.01, m)
    b.AddClumps(t, [[0.0, 0.0, -0.4905]])
    b.SetCDUpdateFreq(0)
    b.SetInitTimeStep(1e-5)
    ctx, sim, p, sc = pair(pkg, orc, b)
    ctx.step(10), sim.step(10)
    a, bb, tt, _ = ctx.contacts()
    assert len(a) == 1 and tt[0] == 11 and a[0] == 0 and bb[0] == 0  # sphere 0 on the bottom plane (component 0)
    gs, os_ = ctx.download_state(), sim.download_state()
    assert abs(gs["vZ"][0] - os_["vZ"][0]) < 1e-5 and gs["vZ"][0] > 1.0  # pushed out of the floor
    ctx.step(40), sim.step(40)  # it has bounced off by now: the list empties on both sides
    assert ctx.counts().nContacts == sim.counts().nContacts == 0


def test_contact_arena_growth(pkg, orc):
    """Dense start (many more contacts than the initial arena guess per sphere) still yields the full set."""
    b = pkg.model.packed_bed(800, seed=23, cd_freq=0, spacing_mult=1.5)  # heavy overlaps: >6 contacts per sphere
    ctx, sim, p, sc = pair(pkg, orc, b)
    ctx.detect(), sim.detect()
    a = assert_same_contacts(ctx, sim)[0]
    assert len(a) > 6 * sc.nSpheres


def test_error_paths(pkg):
    ctx = pkg.Context(0)
    with pytest.raises(pkg.abi.DemeError):
        ctx.detect()  # nothing uploaded
    b = pkg.model.packed_bed(500, seed=1, cd_freq=0, spacing_mult=2.6, init_vz=-0.3)
    b.SetErrorOutVelocity(0.1)  # below the initial speed: detection must refuse (kT.cpp:136-149)
    p, sc = b.Initialize()
    ctx.set_params(p)
    ctx.upload_scene(sc)
    with pytest.raises(pkg.abi.DemeError, match="velocity"):
        ctx.step(1)
    # id-width limits of the key / info records are refused before any array is read
    import ctypes as C
    ctx2 = pkg.Context(0)
    ctx2.set_params(p)
    for field, n, what in (("nSpheres", 1 << 31, "31 bits"), ("nOwners", 1 << 30, "30 bits")):
        big = pkg.abi.DemeScene()
        C.memmove(C.byref(big), C.byref(sc), C.sizeof(big))
        setattr(big, field, n)
        with pytest.raises(pkg.abi.DemeError, match=what):
            ctx2.upload_scene(big)
    q = pkg.abi.DemeParams()
    C.memmove(C.byref(q), C.byref(p), C.sizeof(q))
    q.nContactWildcards = 17
    with pytest.raises(pkg.abi.DemeError, match="wildcards"):
        ctx2.set_params(q)
    q.nContactWildcards, q.binSize = 4, 0.0
    with pytest.raises(pkg.abi.DemeError, match="sizing"):
        ctx2.set_params(q)


@pytest.mark.gpu
def test_cylindrical_boundaries(pkg, orc):
    """analytical cylinders (checkSphereEntityOverlap type 2, DEMHelperKernels.cuh:489-520): a drum wall (normal inward)
    around the bed and a post (normal outward) through it; contact sets bit-exact, trajectories within tolerance"""
    b = pkg.model.packed_bed(1500, seed=19, cd_freq=0, spacing_mult=2.5, init_vz=-0.3, aspect=(1.0, 1.0, 0.6))
    lo, hi = b.user_box_min, b.user_box_max
    cx, cy = float(lo[0] + hi[0]) / 2, float(lo[1] + hi[1]) / 2
    drum = b.AddExternalObject()
    drum.AddCylinder((cx, cy, 0.0), (0, 0, 1), 0.42 * float(hi[0] - lo[0]), 0, normal_inward=True)
    post = b.AddExternalObject()
    post.AddCylinder((cx, cy, 0.0), (0, 0, 1), 0.012, 0, normal_inward=False)
    ctx, sim, p, sc = pair(pkg, orc, b)
    ctx.step(150), sim.step(150)
    a, bb, t, *_ = assert_same_contacts(ctx, sim)
    assert (t == 13).sum() > 20  # SPHERE_CYL_CONTACT entries from both cylinders
    anal_owner = b.arrays["objOwner"][bb[t == 13]]
    assert len(np.unique(anal_owner)) == 2
    gs, os_ = ctx.download_state(), sim.download_state()
    X = pkg.model.decode_positions(gs["voxelID"], gs["locX"], gs["locY"], gs["locZ"], p.nvXp2, p.nvYp2, p.voxelSize, p.l)
    Y = pkg.model.decode_positions(os_["voxelID"], os_["locX"], os_["locY"], os_["locZ"], p.nvXp2, p.nvYp2, p.voxelSize, p.l)
    assert np.abs(X - Y).max() == 0.0  # bit-identical trajectories
    # nothing ended up inside the post or outside the drum
    n = int(sc.nOwnerClumps)
    r = np.hypot(X[:n, 0] + p.LBFX - cx, X[:n, 1] + p.LBFY - cy)
    assert r.min() > 0.012 - 0.002 and r.max() < 0.42 * float(hi[0] - lo[0]) + 0.002


def persistent_scene(pkg):
    b = pkg.model.packed_bed(1000, seed=21, cd_freq=0, spacing_mult=2.6, init_vz=-0.8)
    fam = (np.arange(len(b.batches[0].xyz)) % 2).astype(np.uint8)
    b.batches[0].SetFamily(fam)
    return b


def test_persistent_contacts(pkg, orc):
    """MarkFamilyPersistentContactEither / MarkPersistentContact / Remove* (DEM/API.h:874-905): marked contacts of the current
    list stay in every later list whether or not the sweep finds them; lists, history and state stay bit-identical to the
    oracle, the physics is that of an unmarked twin, and removing the marks returns the twin's list."""
    ctx, sim, p, sc = pair(pkg, orc, persistent_scene(pkg))
    twin = orc.make_sim(pkg, p, sc)
    ctx.step(60), sim.step(60), twin.step(60)
    a0, b0, t0, *_ = assert_same_contacts(ctx, sim)
    assert len(a0) > 200
    ctx.mark_persistent_contacts(ctx.PERSIST_EITHER, 1), sim.mark_persistent_contacts(1, 1)
    nP = ctx.num_persistent_contacts()
    assert nP == sim.num_persistent_contacts() and 0 < nP < len(a0)
    # the marked set can be read back and loaded into another context (restart / re-decomposition)
    pa, pb, pt = ctx.persistent_contacts()
    oa, ob, ot = sim.persistent_contacts()
    assert np.array_equal(pa, oa) and np.array_equal(pb, ob) and np.array_equal(pt, ot) and len(pa) == nP
    other = pkg.Context(0)
    other.set_params(p), other.upload_scene(sc)
    other.set_persistent_contacts(pa[::-1], pb[::-1], pt[::-1])  # any order; duplicates would be dropped
    assert all(np.array_equal(x, y) for x, y in zip(other.persistent_contacts(), (pa, pb, pt)))
    with pytest.raises(pkg.abi.DemeError, match="out of range"):
        other.set_persistent_contacts([int(sc.nSpheres)], [0], [1])
    del other
    marked = set(zip(a0.tolist(), b0.tolist(), t0.tolist()))
    grew = False
    for chunk in range(6):
        ctx.step(40), sim.step(40), twin.step(40)
        a, b, t, *_ = assert_same_contacts(ctx, sim)
        now = set(zip(a.tolist(), b.tolist(), t.tolist()))
        ta, tb, tt, _ = twin.contacts()
        plain = set(zip(ta.tolist(), tb.tolist(), tt.tolist()))
        assert plain <= now and len(now - plain) == len(now) - len(plain)
        assert (now - plain) <= marked  # what the sweep did not find is there only because it was marked
        grew = grew or len(now) > len(plain)
        gs, os_, ts = ctx.download_state(), sim.download_state(), twin.download_state()
        for k in STATE_KEYS:
            assert np.array_equal(gs[k], os_[k]) and np.array_equal(gs[k], ts[k]), (chunk, k)
        for w in range(4):
            assert np.array_equal(ctx.wildcard(w), sim.wildcard(w)), (chunk, w)
    assert grew, "no marked contact ever separated: the scenario does not exercise persistence"
    # every contact marked on top (mode 0), then all marks removed: the next list is the plain detection again
    ctx.mark_persistent_contacts(), sim.mark_persistent_contacts()
    assert ctx.num_persistent_contacts() == sim.num_persistent_contacts() >= nP
    ctx.step(20), sim.step(20), twin.step(20)
    assert_same_contacts(ctx, sim)
    ctx.mark_persistent_contacts(mark=False), sim.mark_persistent_contacts(mark=False)
    assert ctx.num_persistent_contacts() == sim.num_persistent_contacts() == 0
    ctx.step(1), sim.step(1), twin.step(1)
    a, b, t, *_ = assert_same_contacts(ctx, sim)
    ta, tb, tt, _ = twin.contacts()
    assert np.array_equal(a, ta) and np.array_equal(b, tb) and np.array_equal(t, tt)


def test_persistent_contacts_need_history(pkg):
    b = persistent_scene(pkg)
    b.UseFrictionlessHertzianModel()
    p, sc = b.Initialize()
    ctx = pkg.Context(0)
    ctx.set_params(p), ctx.upload_scene(sc)
    ctx.step(5)
    with pytest.raises(pkg.abi.DemeError, match="history-less"):
        ctx.mark_persistent_contacts()


def test_contact_points_on_bin_faces(pkg, orc):
    """The contact-point-in-this-bin rule when the contact point sits within rounding of a bin face: equal spheres placed
    symmetrically about faces of the bin grid (on the position codec's integer lattice, bin size a whole number of lattice
    units), thousands of them crowded into a few bins so that the sweep meets the two entries of a pair in a different
    order in different bins.  The pair must be claimed by exactly one bin and that bin must be the oracle's: with the pair's
    roles taken in meeting order (instead of list order, as the reference's i < j loop does) the two bins could each
    decide the point belongs to the other -- once per ~1e9 pairs in a settling bed, most pairs here."""
    rng = np.random.default_rng(77)
    b = pkg.SceneBuilder()
    m = b.LoadMaterial({"E": 1e8, "nu": 0.3, "CoR": 0.5, "mu": 0.3, "Crr": 0.0})
    b.InstructBoxDomainDimension((0.0, 1.0), (0.0, 1.0), (0.0, 1.0))
    b.InstructBoxDomainBoundingBC("none", m)
    nv, l, voxel = b._figure_out_nv()
    M = int(round(0.02 / l))  # lattice units per bin (l ~ 1e-11 m: 2^37 lattice points per metre)
    b.SetInitBinSize(M * l)
    r = 0.27 * M * l
    t = b.LoadSphereType(1e-3, r, m)
    n_pairs = 3000
    face_axis = rng.integers(0, 3, n_pairs)
    C = rng.integers(20 * M, 23 * M, (n_pairs, 3))  # pair centres: a 3 x 3 x 3 block of bins ...
    C[np.arange(n_pairs), face_axis] = rng.integers(20, 24, n_pairs) * M  # ... with one coordinate ON a bin face
    u = rng.normal(size=(n_pairs, 3))
    u = u / np.linalg.norm(u, axis=1, keepdims=True) * rng.uniform(0.55, 0.98, (n_pairs, 1)) * (r / l)
    u = np.rint(u).astype(np.int64)
    G = np.empty((2 * n_pairs, 3), np.int64)
    swap = rng.random(n_pairs) < 0.5  # which of the two gets the smaller sphere id
    G[0::2] = np.where(swap[:, None], C + u, C - u)
    G[1::2] = np.where(swap[:, None], C - u, C + u)
    b.AddClumps(t, (G * l).astype(np.float32))  # placeholder positions; the codec fields are overwritten below
    b.SetCDUpdateFreq(0)
    p, sc = b.Initialize()
    assert abs(p.binSize - M * p.l) < 1e-18 and p.l == l
    n = 2 * n_pairs
    vox = G // 65536
    b.arrays["voxelID"][:n] = (vox[:, 0] + (vox[:, 1] << p.nvXp2) + (vox[:, 2] << (p.nvXp2 + p.nvYp2))).astype(np.uint64)
    for k, name in enumerate(("locX", "locY", "locZ")):
        b.arrays[name][:n] = (G[:, k] % 65536).astype(np.uint16)
    sc = pkg.abi.make_scene_struct(b.arrays, b.counts)
    ctx = pkg.Context(0)
    ctx.set_params(p), ctx.upload_scene(sc)
    sim = orc.make_sim(pkg, p, sc)
    ctx.compute_margins(0), sim.compute_margins(0)
    ctx.detect(), sim.detect()
    a, bb, tt, *_ = assert_same_contacts(ctx, sim)
    made = set(zip(a.tolist(), bb.tolist()))
    assert all((2 * i, 2 * i + 1) in made for i in range(n_pairs))  # every constructed pair is listed (once: keys are unique)
    key = (a.astype(np.uint64) << np.uint64(33)) | bb.astype(np.uint64)
    assert np.all(key[1:] > key[:-1])
    assert int(ctx.counts().maxSpheresInBin) > 256  # crowded: both the in-LDS and the tiled (giant bin) paths ran


def raft_scene(pkg):
    """a bed of 1 mm spheres with a rigid raft of 283 spheres (one clump) set down on its top layer: > 256 contacts on one FREE
    owner"""
    import math
    r = 0.001
    b = pkg.model.packed_bed(6000, seed=31, cd_freq=0, scale=r, spacing_mult=2.02, jitter=0.0, three_sphere=False,
                             aspect=(1.0, 1.0, 0.6), init_vz=0.0)
    xyz = b.batches[0].xyz
    top = xyz[xyz[:, 2] > xyz[:, 2].max() - 0.2 * r]
    c = top.mean(0)
    rel = (top - c).astype(np.float32)
    n = len(rel)
    L = float(rel[:, 0].max() - rel[:, 0].min())
    mass = n * 2.6e3 * 4 / 3 * math.pi * r ** 3
    t = b.LoadClumpType(mass, (mass * L * L / 12, mass * L * L / 12, mass * L * L / 6), np.full(n, r, np.float32), rel, 0)
    raft = b.AddClumps(t, [[float(c[0]), float(c[1]), float(c[2]) + 1.9995 * r]])
    raft.SetVel(np.array([[0.01, 0.0, -0.02]], np.float32))
    return b, n


def test_free_owner_with_hundreds_of_contacts(pkg, orc):
    """Owners with more than 256 contacts are reduced by a workgroup each (k_reduce_heavy, fixed-shape tree) instead of the
    per-owner gather; the walls exercise that path in every test but are fixed -- here the heavy owner is a free clump whose
    motion depends on the sum.  Lists bit-exact; states to fp32 summation-order tolerance (the tree's order differs from the
    oracle's list order)."""
    b, n_raft = raft_scene(pkg)
    ctx, sim, p, sc = pair(pkg, orc, b)
    own = b.arrays["ownerClumpBody"]
    raft = int(sc.nOwnerClumps) - 1
    assert n_raft > 256 and int((own == raft).sum()) == n_raft
    seen_heavy = 0
    for chunk in range(6):
        ctx.step(10), sim.step(10)
        a, bb, t, *_ = assert_same_contacts(ctx, sim)
        on_raft = int(((own[a] == raft) | ((t == 1) & (own[np.minimum(bb, len(own) - 1)] == raft))).sum())
        seen_heavy += on_raft > 256
        gs, os_ = ctx.download_state(), sim.download_state()
        X, Xo = positions(pkg, gs, p), positions(pkg, os_, p)
        assert np.abs(X - Xo).max() < 1e-9, chunk
        for k in ("vX", "vY", "vZ"):
            assert np.abs(gs[k] - os_[k]).max() < 2e-6, (chunk, k)
        for k in ("omgBarX", "omgBarY", "omgBarZ"):
            assert np.abs(gs[k] - os_[k]).max() < 2e-3, (chunk, k)  # (1 mm spheres: 1e-6 m/s at the surface)
        assert abs(gs["vZ"][raft] - os_["vZ"][raft]) < 1e-7
    assert seen_heavy >= 2  # the raft really carried > 256 contacts while it was being decelerated
    assert gs["vZ"][raft] > -0.02 + 0.005


def test_large_size_ratio_incidence_and_contacts(pkg, orc):
    """a 12 mm sphere among 1 mm spheres (bins of 4 mm): the big sphere is registered in hundreds of bins and meets every
    small neighbour in several of them; incidences, contact list and 40 steps of motion against the oracle"""
    import math
    r, R = 0.001, 0.012
    b = pkg.model.packed_bed(9000, seed=31, cd_freq=0, scale=r, spacing_mult=2.02, jitter=0.01, three_sphere=False,
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
    big.SetVel(np.array([[0.3, 0.1, -0.5]], np.float32))
    ctx, sim, p, sc = pair(pkg, orc, b)
    ctx.compute_margins(0), sim.compute_margins(0)
    ctx.detect(), sim.detect()
    gi, oi = ctx.bin_incidence(), sim.bin_incidence()
    assert np.array_equal(gi[0], oi[0]) and np.array_equal(gi[1], oi[1])
    big_sphere = int(sc.nSpheres) - 1
    assert int((gi[1] == big_sphere).sum()) > 200  # bins the big sphere is registered in
    assert_same_contacts(ctx, sim)
    ctx.migrate(), sim.migrate()
    ctx.step(40), sim.step(40)
    a, bb, tt, *_ = assert_same_contacts(ctx, sim)
    assert int(((a == big_sphere) | ((tt == 1) & (bb == big_sphere))).sum()) >= 10  # it has ploughed into its neighbours
    gs, os_ = ctx.download_state(), sim.download_state()
    for k in STATE_KEYS:
        assert np.array_equal(gs[k], os_[k]), k


def test_added_acceleration_for_one_step(pkg, orc):
    """DEMTracker::AddAcc / AddAngAcc (co-simulation hand-over): accelerations the script adds act on the coming step only, on
    top of the contact sums; bit-identical to the oracle; a later call for the same owner replaces the earlier one"""
    b = pkg.model.packed_bed(1000, seed=8, cd_freq=0, spacing_mult=2.5, init_vz=-0.4)
    ctx, sim, p, sc = pair(pkg, orc, b)
    twin = orc.make_sim(pkg, p, sc)
    ctx.step(60), sim.step(60), twin.step(60)
    rng = np.random.default_rng(0)
    for k in range(5):
        acc = rng.uniform(-50, 50, (40, 3)).astype(np.float32)
        ang = rng.uniform(-500, 500, (40, 3)).astype(np.float32)
        for s in (ctx, sim):
            s.add_owner_acc(100, acc * 9.0)        # replaced by the next call
            s.add_owner_acc(100, acc, ang)
            s.add_owner_acc(500, None, ang[:3])    # angular part only
        ctx.step(1), sim.step(1), twin.step(1)
        gs, os_ = ctx.download_state(), sim.download_state()
        for key in STATE_KEYS:
            assert np.array_equal(gs[key], os_[key]), (k, key)
        ctx.step(3), sim.step(3), twin.step(3)  # nothing pending any more
    gs, os_, ts = ctx.download_state(), sim.download_state(), twin.download_state()
    for key in STATE_KEYS:
        assert np.array_equal(gs[key], os_[key]), key
    dv = np.abs(gs["vX"] - ts["vX"]) + np.abs(gs["vY"] - ts["vY"]) + np.abs(gs["vZ"] - ts["vZ"])
    assert (dv[100:140] > 1e-5).all() and (np.abs(gs["omgBarX"][500:503] - ts["omgBarX"][500:503]) > 1e-6).all()
    with pytest.raises(pkg.abi.DemeError, match="out of range"):
        ctx.add_owner_acc(int(sc.nOwners) - 1, np.zeros((2, 3), np.float32))


def test_mixed_multi_sphere_templates_from_the_reference_data(pkg, orc):
    """a bed mixing the reference's own clump shapes (data/clumps: 3_clump, ellipsoid_2_1_1 with 5 components, 6_clump with 6) --
    different component counts per clump, templates re-ordered by component count at initialisation (APIPrivate.cpp:696-742);
    200 steps bit-identical to the oracle"""
    import os
    ref = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_data")
    b = pkg.model.packed_bed(900, seed=17, cd_freq=0, scale=0.004, spacing_mult=2.6, init_vz=-0.6, E=1e7)  # many shapes overlap at once
    rho, s = 2.6e3, 0.004
    shapes = []
    for name, vol, moi in (("ellipsoid_2_1_1.csv", 7.0, (6.0, 6.0, 3.0)), ("6_clump.csv", 2.7564385, (1.0352626, 0.9616627, 1.6978352))):
        rel, rad = pkg.io.read_clump_template_csv(os.path.join(ref, name))
        t = b.LoadClumpType(rho * vol, tuple(rho * m for m in moi), rad, rel, 0)
        shapes.append(t.Scale(s))
    assert [len(t.radii) for t in shapes] == [5, 6]
    batch = b.batches[0]
    rng = np.random.default_rng(3)
    pick = rng.integers(0, 3, len(batch.xyz))
    batch.templates = [batch.templates[i] if k == 0 else shapes[k - 1] for i, k in enumerate(pick)]
    ctx, sim, p, sc = pair(pkg, orc, b)
    n_comp = np.bincount(np.asarray(b.arrays["ownerClumpBody"]), minlength=int(sc.nOwnerClumps))[:int(sc.nOwnerClumps)]
    assert set(np.unique(n_comp).tolist()) == {3, 5, 6} and int(sc.nSpheres) == int(n_comp.sum())
    most = 0
    for chunk, n in enumerate((5, 15, 30, 50, 100)):
        ctx.step(n), sim.step(n)
        most = max(most, len(assert_same_contacts(ctx, sim)[0]))
        gs, os_ = ctx.download_state(), sim.download_state()
        for k in STATE_KEYS:
            assert np.array_equal(gs[k], os_[k]), (chunk, k)
    assert most > 300  # (the overlapping shapes push each other apart within a few dozen steps)


def test_metre_scale_scene(pkg, orc):
    """the same path at metre scale (0.4 m spheres, 2 m bins, a 100 m box): the sweep's fp32 pre-filter must stay conservative
    when its inputs are a million times larger than in the millimetre beds -- lists and 30 steps bit-identical to the oracle"""
    b = pkg.model.packed_bed(1200, seed=12, cd_freq=0, scale=0.5, spacing_mult=2.45, jitter=0.08, init_vz=-3.0, h=2e-4, E=1e7,
                             bin_multiple=5.0)
    lo, hi = b.user_box_min, b.user_box_max
    assert float((hi - lo).max()) > 10.0
    ctx, sim, p, sc = pair(pkg, orc, b)
    ctx.compute_margins(0), sim.compute_margins(0)
    ctx.detect(), sim.detect()
    a = assert_same_contacts(ctx, sim)[0]
    assert len(a) > 300  # the jittered lattice starts with grazing and overlapping neighbours
    ctx.migrate(), sim.migrate()
    ctx.step(30), sim.step(30)
    assert_same_contacts(ctx, sim)
    gs, os_ = ctx.download_state(), sim.download_state()
    for k in STATE_KEYS:
        assert np.array_equal(gs[k], os_[k]), k
