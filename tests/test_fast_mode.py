"""The FAST arithmetic mode (the library's default, include/deme_hip.h DEME_ARITH_FAST) against the CPU oracle.

Contact detection is the same decision code in both modes: lists, bins and history maps must be bit-identical.  The per-step
kernels re-associate the reference's formulas (world-frame evaluation, per-owner conversion, 1-ulp hardware rcp / sqrt, FMA
contraction), so forces and trajectories agree with the oracle to fp32 rounding.  STATED TOLERANCE (north_star: "positions /
velocities within a stated fp32 tolerance after N steps"):
  * owner accelerations of one force evaluation: 2e-4 of the largest acceleration in the scene per component (single
    contacts agree to ~1e-5; the bound leaves room for the sum over an owner's contacts),
  * after N = 100 steps of a settling bed (h = 5e-6 s, clumps moving at up to 1.5 m/s): positions within 5e-8 m (one fp32
    ulp of a coordinate in a 1 m box is 6e-8 m; 1e-5 of a sphere radius), velocities within 2e-4 m/s, contact sets identical.
    A granular bed is chaotic: measured on this scene the fast-vs-exact difference is 3e-12 m after 1 step (one fp32 ulp in
    the velocities), 6e-9 m after 100, 1.4e-8 m after 200, 7e-8 m after 500 -- the bound is for N = 100, not a growth law.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _positions(pkg, p, st):
    return pkg.model.decode_positions(st["voxelID"], st["locX"], st["locY"], st["locZ"], p.nvXp2, p.nvYp2, p.voxelSize, p.l)


def _bed(pkg, n=3000, cd_freq=0, **kw):
    # the bench recipe at test size: lattice without initial overlaps, dropped at 1 m/s; ~3.5 contacts per clump once settled
    return pkg.model.packed_bed(n, seed=11, cd_freq=cd_freq, spacing_mult=3.0, init_vz=-1.0, aspect=(1.0, 1.0, 0.25), **kw)


def _settled(pkg, b, steps=10000):
    """(params, scene, state) of the bed after `steps` steps in the exact mode: a packed starting point with live histories"""
    p, sc = b.Initialize()
    ctx = pkg.Context(0)
    ctx.set_arith_mode("exact")
    ctx.set_params(p), ctx.upload_scene(sc)
    ctx.step(steps)
    st = ctx.download_state()
    ctx.close()
    return p, sc, {k: st[k] for k in st if k not in ("aX", "aY", "aZ", "alphaX", "alphaY", "alphaZ")}


@pytest.mark.parametrize("model", ["hertz", "frictionless"])
def test_fast_accelerations_match_oracle(pkg, orc, model):
    b = _bed(pkg)
    if model == "frictionless":
        b.UseFrictionlessHertzianModel()
    p, sc, st = _settled(pkg, b)
    ctx = pkg.Context(0)
    ctx.set_arith_mode("fast")
    assert ctx.arith_mode() == "fast"
    ctx.set_params(p), ctx.upload_scene(sc), ctx.upload_state(st)
    sim = orc.make_sim(pkg, p, sc)
    sim.upload_state(st)
    for s in (ctx, sim):
        s.compute_margins(0), s.detect(), s.migrate(), s.calc_forces()
    assert ctx.force_kernel()[0] == ("k_tile_forces<0, false>" if model == "hertz" else "k_tile_forces<1, false>"), ctx.force_kernel()
    ga, oa = ctx.contacts(), sim.contacts()
    assert len(ga[0]) > 1500 and all(np.array_equal(x, y) for x, y in zip(ga[:3], oa[:3]))  # decisions: bit-exact
    g, o = ctx.download_state(), sim.download_state()
    n = int(sc.nOwnerClumps)
    for keys in (("aX", "aY", "aZ"), ("alphaX", "alphaY", "alphaZ")):
        G = np.stack([g[k][:n] for k in keys], 1).astype(np.float64)
        O = np.stack([o[k][:n] for k in keys], 1).astype(np.float64)
        scale = np.abs(O).max()
        assert scale > 0
        print(f"{model} {keys[0][:-1]}: max |fast - oracle| / max |oracle| = {np.abs(G - O).max() / scale:.3e} (scale {scale:.3e})")
        assert np.abs(G - O).max() <= 2e-4 * scale, (keys, np.abs(G - O).max() / scale)
    # the wall owner (one heavy owner, tree reduction): same bound as in the exact mode's test
    assert abs(g["aZ"][n] - o["aZ"][n]) <= 1e-3 * max(1.0, abs(o["aZ"][n]))
    if model == "hertz":  # contact history written by the fast kernel
        for w in range(4):
            gw, ow = ctx.wildcard(w), sim.wildcard(w)
            print(f"wildcard {w}: max diff {np.abs(gw - ow).max():.3e} of {np.abs(ow).max():.3e}")
            assert np.abs(gw - ow).max() <= 1e-5 * max(np.abs(ow).max(), 1e-12) + 1e-12
    ctx.close()


@pytest.mark.parametrize("model", ["hertz", "frictionless"])
def test_contact_records_of_the_tile_pass_match_oracle(pkg, orc, model):
    """Contact recording (the reference's default contact output: force and contact points per pair) is written by the tile pass
    itself (k_tile_forces<M, MESH, REC = true>): a script that leaves the default output content on keeps the fast kernel.
    Per-contact force, torque-only force and the contact point in both owners' body frames against the oracle's records:
    forces within 1e-5 of the largest, points within 2e-8 m."""
    b = _bed(pkg, Crr=0.05)  # (rolling resistance on: the torque-only force is not identically zero)
    if model == "frictionless":
        b.UseFrictionlessHertzianModel()
    p, sc, st = _settled(pkg, b)
    ctx = pkg.Context(0)
    ctx.set_arith_mode("fast")
    ctx.set_params(p), ctx.upload_scene(sc), ctx.upload_state(st)
    ctx.set_record_contacts(True)
    sim = orc.make_sim(pkg, p, sc)
    sim.upload_state(st)
    for s in (ctx, sim):
        s.compute_margins(0), s.detect(), s.migrate()
    ctx.calc_forces(), sim.calc_forces(record=True)
    assert ctx.force_kernel()[0] == ("k_tile_forces<0, false>" if model == "hertz" else "k_tile_forces<1, false>"), ctx.force_kernel()
    ga, oa = ctx.contacts(), sim.contacts()
    assert len(ga[0]) > 1500 and all(np.array_equal(x, y) for x, y in zip(ga[:3], oa[:3]))
    F, T, PA, PB = ctx.contact_records()
    oF, oT, oPA, oPB = sim.contact_records()
    scale = np.abs(oF).max()
    assert scale > 0 and np.abs(F - oF).max() <= 1e-5 * scale, np.abs(F - oF).max() / scale
    assert np.abs(T - oT).max() <= 1e-5 * scale + 1e-12
    ss = ga[2] == 1  # (a wall's body frame is the world frame shifted by metres: its contact point is compared relatively)
    assert np.abs(PA - oPA).max() <= 2e-8 and np.abs(PB - oPB)[ss].max() <= 2e-8
    assert np.abs(PB - oPB)[~ss].max() <= 1e-6 * max(np.abs(oPB)[~ss].max(), 1.0) if (~ss).any() else True
    ctx.close()


@pytest.mark.parametrize("fused", [False, True], ids=["two_kernels", "one_kernel_step"])
@pytest.mark.parametrize("cd_freq", [0, 10])
def test_fast_trajectory_within_stated_tolerance(pkg, orc, cd_freq, fused):
    b = _bed(pkg, cd_freq=cd_freq)
    if cd_freq:
        b.SetExpandSafetyAdder(0.5)
    p, sc, st = _settled(pkg, b)
    ctx = pkg.Context(0)
    ctx.set_arith_mode("fast")
    ctx.set_fused_step(fused)  # (deme_tile_step.h: closed tiles that integrate their own owners -- an option, off by default)
    ctx.set_params(p), ctx.upload_scene(sc), ctx.upload_state(st)
    sim = orc.make_sim(pkg, p, sc)
    sim.upload_state(st)
    N = 100
    ctx.step(N), sim.step(N)
    assert ctx.force_kernel()[0] == ("k_tile_step<0>" if fused else "k_tile_forces<0, false>"), ctx.force_kernel()
    ga, oa = ctx.contacts(), sim.contacts()
    assert len(ga[0]) == len(oa[0]) and all(np.array_equal(x, y) for x, y in zip(ga[:3], oa[:3]))
    g, o = ctx.download_state(), sim.download_state()
    dx = np.abs(_positions(pkg, p, g) - _positions(pkg, p, o)).max()
    dv = max(np.abs(g[k] - o[k]).max() for k in ("vX", "vY", "vZ"))
    dw = max(np.abs(g[k] - o[k]).max() for k in ("omgBarX", "omgBarY", "omgBarZ"))
    dq = max(np.abs(g[k] - o[k]).max() for k in ("oriQw", "oriQx", "oriQy", "oriQz"))
    print(f"fast vs oracle after {N} steps (K={cd_freq}): |dx| {dx:.3e} m, |dv| {dv:.3e} m/s, |dw| {dw:.3e} rad/s, |dq| {dq:.3e}")
    assert dx <= 5e-8 and dv <= 2e-4 and dw <= 5e-2 and dq <= 2e-5
    # a / alpha of the LAST step (the one-kernel step keeps none: the download replays its force evaluation on the buffers of its start)
    n = int(sc.nOwnerClumps)
    for keys in (("aX", "aY", "aZ"), ("alphaX", "alphaY", "alphaZ")):
        G = np.stack([g[k][:n] for k in keys], 1).astype(np.float64)
        O = np.stack([o[k][:n] for k in keys], 1).astype(np.float64)
        assert np.abs(G - O).max() <= 5e-3 * np.abs(O).max(), (keys, np.abs(G - O).max() / np.abs(O).max())
    for w in range(4):  # ... and the contact history it carries in its second buffer
        gw, ow = ctx.wildcard(w), sim.wildcard(w)
        print(f"wildcard {w} after {N} steps: max diff {np.abs(gw - ow).max():.3e} of {np.abs(ow).max():.3e}")
        assert np.abs(gw - ow).max() <= 2e-3 * max(np.abs(ow).max(), 1e-12) + 1e-9, (w, np.abs(gw - ow).max(), np.abs(ow).max())
    ctx.close()


def test_fast_and_exact_modes_share_the_contact_path(pkg):
    """switching the mode on a live context: the list is untouched, the next step runs the other kernels"""
    b = _bed(pkg, n=1500, cd_freq=50)
    b.SetExpandSafetyAdder(1.0)
    p, sc = b.Initialize()
    ctx = pkg.Context(0)
    ctx.set_arith_mode("exact")
    ctx.set_params(p), ctx.upload_scene(sc)
    ctx.step(280)  # detections at steps 0, 50, ..., 250
    a0 = ctx.contacts()
    ctx.set_arith_mode("fast")
    ctx.step(3)
    ctx.set_arith_mode("exact")
    ctx.step(2)  # 285 steps: the list is still the one of step 250
    a1 = ctx.contacts()
    assert all(np.array_equal(x, y) for x, y in zip(a0[:3], a1[:3]))
    st = ctx.download_state()
    assert np.isfinite(st["vZ"]).all()
    ctx.close()


def test_fast_mode_with_mesh_planes_and_families(pkg, orc):
    """sphere-mesh contacts (general kernel, world-frame contributions), a non-trivial family table and a cylinder in fast mode"""
    b = pkg.model.packed_bed(1200, seed=11, cd_freq=0, spacing_mult=3.0, init_vz=-1.0, aspect=(1.0, 1.0, 0.45))
    lo, hi = b.user_box_min, b.user_box_max
    v, f = pkg.model.plate_mesh(12, 12, float(hi[0] - lo[0]) * 0.9, float(hi[1] - lo[1]) * 0.9, z=0.0, wavy=0.001)
    m = b.AddMeshObject(v, f, 0)
    m.SetInitPos(((lo[0] + hi[0]) / 2, (lo[1] + hi[1]) / 2, float(lo[2]) + 0.004))
    m.SetFamily(3)
    b.SetFamilyFixed(3)
    b.SetFamilyExtraMargin(0, 1e-5)
    p, sc = b.Initialize()
    ctx = pkg.Context(0)
    ctx.set_arith_mode("fast")
    ctx.set_params(p), ctx.upload_scene(sc)
    sim = orc.make_sim(pkg, p, sc)
    N = 2500
    ctx.step(N), sim.step(N)
    ga, oa = ctx.contacts(), sim.contacts()
    assert (ga[2] == 2).sum() > 20, "the bed should be resting on the plate"
    assert len(ga[0]) == len(oa[0]) and all(np.array_equal(x, y) for x, y in zip(ga[:3], oa[:3]))
    g, o = ctx.download_state(), sim.download_state()
    dx = np.abs(_positions(pkg, p, g) - _positions(pkg, p, o)).max()
    dv = max(np.abs(g[k] - o[k]).max() for k in ("vX", "vY", "vZ"))
    print(f"fast vs oracle, mesh scene after {N} steps: |dx| {dx:.3e} m, |dv| {dv:.3e} m/s")
    assert dx <= 5e-8 and dv <= 5e-5
    ctx.close()


def test_a_tiled_list_serves_the_other_kernels_on_demand(pkg):
    """A list built for the owner-tile pass (fast mode, detection every 20 steps) holds no B-sorted form; the library builds it when
    the list is evaluated by the other kernels after all.  Two cases, each against a twin that used those kernels from the start
    of the same list: (a) contact recording is switched on in the middle of a list's service, (b) the arithmetic mode is switched
    to exact in the middle.  The twins must agree bit for bit (same kernels, same list, same state) -- a missing or stale B-sorted
    list would lose every B-side contribution."""
    K = 20
    b = pkg.model.packed_bed(3000, seed=21, cd_freq=K, spacing_mult=3.0, init_vz=-0.8, aspect=(1.0, 1.0, 0.5))
    b.SetExpandSafetyAdder(1.0)
    p, sc = b.Initialize()

    def fresh():
        c = pkg.Context(0)
        c.set_arith_mode("fast")
        c.set_params(p), c.upload_scene(sc)
        c.step(8000 + 7)  # settled on the floor; seven steps into a list's service
        return c

    keys = ("voxelID", "locX", "locY", "locZ", "oriQw", "oriQx", "oriQy", "oriQz", "vX", "vY", "vZ", "omgBarX", "omgBarY", "omgBarZ")
    # (a) recording
    a, t = fresh(), fresh()
    assert a.force_kernel()[0].startswith("k_tile_") and int(a.counts().nContacts) > 2000
    sa, st = a.download_state(), t.download_state()
    assert all(np.array_equal(sa[k], st[k]) for k in keys)
    a.set_record_contacts(True)  # (since round 4 the tile pass records itself: the list keeps its kernel; case (b) is the on-demand one)
    a.step(5)
    # the twin: the same state and history, re-detected with recording on from the start of its list (no tiles at all)
    t.set_record_contacts(True)
    t.upload_state({k: st[k] for k in keys})  # marks the list stale: a fresh detection (same positions -> same pairs, wider or equal margins)
    t.step(5)
    sa, st = a.download_state(), t.download_state()
    dx = max(float(np.abs(sa[k].astype(np.float64) - st[k].astype(np.float64)).max()) for k in ("vX", "vY", "vZ"))
    assert dx < 1e-6, dx  # (the twin's list is a different one -- rebuilt at step 3007 -- so sums differ in order, not in content)
    fa = a.contact_records()[0]
    assert np.isfinite(fa).all() and (np.abs(fa).sum(1) > 0).sum() > 300  # (most listed pairs are inside the margin, not touching)
    a.close(), t.close()
    # (b) the mode switch: both contexts switch at the same step of the same list; one of them had its B-sorted form built at the
    # detection already (DEME_TILE cannot be switched per context, so the twin switches one step earlier and back: its lists exist)
    a, t = fresh(), fresh()
    t.set_arith_mode("exact"), t.set_arith_mode("fast")  # no step in between: state untouched, nothing built yet
    a.set_arith_mode("exact"), t.set_arith_mode("exact")
    a.step(9), t.step(9)
    sa, st = a.download_state(), t.download_state()
    assert all(np.array_equal(sa[k], st[k]) for k in keys)
    # and against a context that ran the exact kernels on its own list from the same state: same physics within the fast/exact gap
    a.close(), t.close()
