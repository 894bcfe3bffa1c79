"""Multi-GPU slab decomposition.  CPU tests: bookkeeping, a two-slab run with the ghost exchange
done in host memory, and the same exchange over torch.distributed (gloo, world_size 2, one process per slab)
-- each slab stepped by the CPU oracle -- against a single-domain run.  The GPU test does the two-slab run with
two HIP contexts and the device-side pack/unpack kernels."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GKEYS = ("voxelID", "locX", "locY", "locZ", "oriQw", "oriQx", "oriQy", "oriQz", "vX", "vY", "vZ", "omgBarX", "omgBarY",
         "omgBarZ")


def build_global(pkg, n=1600, seed=4, cd_freq=0):
    b = pkg.model.packed_bed(n, seed=seed, cd_freq=cd_freq, spacing_mult=2.5, init_vz=-0.4, aspect=(2.0, 1.0, 0.5))
    p, sc = b.Initialize()
    x = np.concatenate([bb.xyz for bb in b.batches])[:, 0]
    return b, p, sc, x


def test_decomposition_bookkeeping(pkg):
    b, p, sc, x = build_global(pkg)
    parts = pkg.decomp.decompose(b.arrays, b.counts, x, 3, halo=0.03)
    assert sum(pt["n_own"] for pt in parts) == sc.nOwnerClumps
    all_own = np.sort(np.concatenate([pt["global_ids"] for pt in parts]))
    assert (all_own == np.arange(sc.nOwnerClumps)).all()  # every clump owned exactly once
    for r, pt in enumerate(parts):
        a, c = pt["arrays"], pt["counts"]
        assert len(a["voxelID"]) == c["nOwners"] and len(a["ownerClumpBody"]) == c["nSpheres"] == 3 * c["nOwnerClumps"]
        gh = a["ownerGhost"]
        assert gh[pt["n_own"]:c["nOwnerClumps"]].all() and not gh[:pt["n_own"]].any() and not gh[c["nOwnerClumps"]:].any()
        gids = np.concatenate([pt["ghost_left_g"], pt["ghost_right_g"]])
        assert (a["familyID"][pt["n_own"]:c["nOwnerClumps"]] == b.arrays["familyID"][gids]).all()  # ghosts keep their family
        assert a["objOwner"].min() >= c["nOwnerClumps"]  # walls renumbered behind the clumps
        if r + 1 < len(parts):  # what I send right is what my right neighbour receives from its left, same order
            nb = parts[r + 1]
            assert len(pt["send_right"]) == len(nb["recv_left"]) > 0
            assert (pt["global_ids"][pt["send_right"]] == nb["ghost_left_g"]).all()
            assert (a["voxelID"][pt["send_right"]] == nb["arrays"]["voxelID"][nb["recv_left"]]).all()


def run_slabs(pkg, make_sim, parts, p, steps, exchange):
    sims = [make_sim(p, pt["scene"]) for pt in parts]
    for _ in range(steps):
        exchange(sims)
        for s in sims:
            s.step(1)
    return sims


def gather_positions(pkg, parts, sims, p, n_clumps):
    X = np.zeros((n_clumps, 3))
    V = np.zeros((n_clumps, 3))
    for pt, s in zip(parts, sims):
        st = s.download_state()
        n = pt["n_own"]
        X[pt["global_ids"]] = pkg.model.decode_positions(st["voxelID"], st["locX"], st["locY"], st["locZ"], p.nvXp2, p.nvYp2,
                                                         p.voxelSize, p.l)[:n]
        V[pt["global_ids"]] = np.stack([st["vX"], st["vY"], st["vZ"]], 1)[:n]
    return X, V


def host_exchange(pkg, parts):
    def ex(sims):
        states = [s.download_state() for s in sims]
        pkg.decomp.exchange_host(parts, states)
        for s, st in zip(sims, states):
            s.upload_state({k: st[k] for k in GKEYS})
    return ex


def test_free_replicated_owner_is_flagged_not_refused(pkg):
    """a mesh that moves under contact forces: refused by default (its accelerations need the cross-slab sum), flagged 2 on every
    slab with shared_free=True (abi.HaloGroup / deme_halo_group_step then all-reduces them); the fixed walls stay unflagged"""
    from tests.test_mesh import mesh_bed
    b = mesh_bed(pkg, 600, seed=3, fixed=False)
    p, sc = b.Initialize()
    x = np.concatenate([bb.xyz for bb in b.batches])[:, 0]
    with pytest.raises(ValueError, match="shared_free=True"):
        pkg.decomp.decompose(b.arrays, b.counts, x, 2, halo=0.035)
    parts = pkg.decomp.decompose(b.arrays, b.counts, x, 2, halo=0.035, shared_free=True)
    for pt in parts:
        gh, c = pt["arrays"]["ownerGhost"], pt["counts"]
        assert list(np.nonzero(gh == 2)[0]) == [c["nOwners"] - 1]  # the mesh owner is the last one
        assert (gh[pt["n_own"]:c["nOwnerClumps"]] == 1).all() and not gh[:pt["n_own"]].any()
        assert pt["arrays"]["ownerMesh"].min() == c["nOwners"] - 1
    fixed = mesh_bed(pkg, 600, seed=3, fixed=True)
    fixed.Initialize()
    assert not any((pt["arrays"]["ownerGhost"] == 2).any() for pt in pkg.decomp.decompose(fixed.arrays, fixed.counts, x, 2, halo=0.035))


def test_slab_local_construction_equals_the_cut_of_the_whole_bed(pkg):
    """model.packed_bed(slab=(rank, n, halo)) + decompose(edges=, only_rank=): a rank builds its own slab of the bed (own clumps
    and ghosts) without ever holding the whole bed's scene -- same parameters, same arrays, same exchange lists as cutting the
    whole bed (what bench.py --gpus N does per rank)"""
    kw = dict(seed=4, cd_freq=5, spacing_mult=2.5, init_vz=-0.4, aspect=(3.0, 1.0, 0.4), order="morton")
    n, world, halo = 2400, 3, 0.035
    whole = pkg.model.packed_bed(n, **kw)
    p, sc = whole.Initialize()
    x = np.concatenate([bb.xyz for bb in whole.batches])[:, 0]
    parts = pkg.decomp.decompose(whole.arrays, whole.counts, x, world, halo)
    for r in range(world):
        b = pkg.model.packed_bed(n, slab=(r, world, halo), **kw)
        pr, scr = b.Initialize()
        for name, _ in pkg.abi.DemeParams._fields_:
            assert getattr(pr, name) == getattr(p, name), name
        assert b.slab_total == int(sc.nOwnerClumps) and np.array_equal(b.slab_edges, parts[r]["all_edges"])
        mine = pkg.decomp.decompose(b.arrays, b.counts, b.slab_x, world, halo, edges=b.slab_edges, only_rank=r)[r]
        ref = parts[r]
        assert mine["n_own"] == ref["n_own"]
        assert np.array_equal(b.slab_global_ids[mine["global_ids"]], ref["global_ids"])
        for k in ("send_left", "send_right", "recv_left", "recv_right"):
            assert np.array_equal(mine[k], ref[k]), k
        for k, v in ref["arrays"].items():
            assert np.array_equal(np.asarray(mine["arrays"][k]), np.asarray(v)), k
        assert mine["counts"] == ref["counts"]


def test_two_slabs_equal_single_domain_oracle(pkg, orc):
    b, p, sc, x = build_global(pkg)
    parts = pkg.decomp.decompose(b.arrays, b.counts, x, 2, halo=0.035)
    steps = 120
    sims = run_slabs(pkg, lambda pp, s: orc.make_sim(pkg, pp, s), parts, p, steps, host_exchange(pkg, parts))
    X, V = gather_positions(pkg, parts, sims, p, sc.nOwnerClumps)
    one = orc.make_sim(pkg, p, sc)
    one.step(steps)
    st = one.download_state()
    n = sc.nOwnerClumps
    X1 = pkg.model.decode_positions(st["voxelID"], st["locX"], st["locY"], st["locZ"], p.nvXp2, p.nvYp2, p.voxelSize, p.l)[:n]
    V1 = np.stack([st["vX"], st["vY"], st["vZ"]], 1)[:n]
    assert one.counts().nContacts > 200
    # contacts across the cut exist, so this really exercises the ghosts
    a, bb, t, _ = sims[0].contacts()
    own = parts[0]["arrays"]["ownerClumpBody"]
    cross = ((own[a] < parts[0]["n_own"]) != (own[bb] < parts[0]["n_own"])) & (t == 1)
    assert cross.sum() > 5
    assert np.abs(X - X1).max() < 2e-7 and np.abs(V - V1).max() < 2e-3


def _two_family_bed(pkg, n=1600, seed=4, cd_freq=0):
    """the bed of build_global with its clumps dealt into families 1 and 2 (checkerboard by id), contacts between the two
    families disabled and an extra margin on family 2: the family tables are not trivial on either side of any cut"""
    b = pkg.model.packed_bed(n, seed=seed, cd_freq=cd_freq, spacing_mult=2.5, init_vz=-0.4, aspect=(2.0, 1.0, 0.5))
    fam = (1 + (np.arange(len(b.batches[0].xyz)) % 2)).astype(np.uint8)
    b.batches[0].SetFamily(fam)
    b.DisableContactBetweenFamilies(1, 2)
    b.SetFamilyExtraMargin(2, 2e-4)
    p, sc = b.Initialize()
    x = np.concatenate([bb.xyz for bb in b.batches])[:, 0]
    return b, p, sc, x


def _global_rows(part, sim):
    a, bb, t, _ = sim.contacts()
    sg = part["sphere_global"]
    ss = t == 1
    gA = sg[a]
    gB = np.where(ss, sg[np.where(ss, bb, 0)], bb.astype(np.int64))
    flip = ss & (gA > gB)
    return np.stack([np.where(flip, gB, gA), np.where(flip, gA, gB), t.astype(np.int64)], 1)


def test_family_masks_and_margins_hold_across_a_cut_oracle(pkg, orc):
    """ghost copies keep their family (DemeScene.ownerGhost marks them): a contact mask between two families and a family's
    extra margin act across a slab cut as inside a slab -- the union of the slabs' contact lists equals the single-domain list,
    and it holds no pair of the masked families"""
    b, p, sc, x = _two_family_bed(pkg)
    parts = pkg.decomp.decompose(b.arrays, b.counts, x, 3, halo=0.035)
    steps = 60
    sims = run_slabs(pkg, lambda pp, s: orc.make_sim(pkg, pp, s), parts, p, steps, host_exchange(pkg, parts))
    one = orc.make_sim(pkg, p, sc)
    one.step(steps)
    rows = np.unique(np.concatenate([_global_rows(pt, s) for pt, s in zip(parts, sims)]), axis=0)
    a, bb, t, _ = one.contacts()
    ref = np.unique(np.stack([a.astype(np.int64), bb.astype(np.int64), t.astype(np.int64)], 1), axis=0)
    assert len(ref) > 150 and np.array_equal(rows, ref)
    own = np.asarray(b.arrays["ownerClumpBody"], np.int64)
    fam = np.asarray(b.arrays["familyID"])
    ss = ref[:, 2] == 1
    fa, fb = fam[own[ref[ss, 0]]], fam[own[ref[ss, 1]]]
    assert ss.sum() > 50 and not ((fa != fb)).any()  # families 1 and 2 never meet; same-family pairs do
    # and pairs straddle the cuts
    a0, b0, t0, _ = sims[0].contacts()
    o0 = parts[0]["arrays"]["ownerClumpBody"]
    assert (((o0[a0] < parts[0]["n_own"]) != (o0[b0] < parts[0]["n_own"])) & (t0 == 1)).sum() > 3
    X, V = gather_positions(pkg, parts, sims, p, sc.nOwnerClumps)
    st = one.download_state()
    n = sc.nOwnerClumps
    X1 = pkg.model.decode_positions(st["voxelID"], st["locX"], st["locY"], st["locZ"], p.nvXp2, p.nvYp2, p.voxelSize, p.l)[:n]
    assert np.abs(X - X1).max() < 2e-7


def test_gloo_world2_halo_exchange(pkg):
    """One process per slab, ghost records moved with torch.distributed (gloo) send/recv."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_gloo_halo_worker.py")], capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "GLOO_HALO_OK" in out.stdout
    assert "GLOO_REDECOMP_OK" in out.stdout  # clumps and their contact history migrated between the two processes
    assert "GLOO_NEIGHBOUR_MIGRATION_OK" in out.stdout  # ... and again with face-neighbour messages only (migrate_neighbours)


@pytest.mark.gpu
def test_two_slabs_on_gpu_match_single_domain(pkg):
    import ctypes as C
    b, p, sc, x = build_global(pkg, n=3000, seed=6)
    parts = pkg.decomp.decompose(b.arrays, b.counts, x, 2, halo=0.035)

    def make(pp, s):
        ctx = pkg.Context(0)
        ctx.set_params(pp)
        ctx.upload_scene(s)
        return ctx

    ctxs = [make(p, pt["scene"]) for pt in parts]
    # device buffers through the HIP runtime the library itself is linked against (no torch here: importing
    # torch after libdeme_hip.so would pull a second ROCm runtime into the process)
    hip = C.CDLL("libamdhip64.so")

    class DevArr:
        def __init__(self, host=None, nbytes=0):
            self.ptr = C.c_void_p()
            nbytes = host.nbytes if host is not None else nbytes
            assert hip.hipMalloc(C.byref(self.ptr), C.c_size_t(max(nbytes, 16))) == 0
            if host is not None and host.nbytes:
                assert hip.hipMemcpy(self.ptr, C.c_void_p(host.ctypes.data), C.c_size_t(host.nbytes), 1) == 0

        def data_ptr(self):
            return self.ptr.value

    ids = [{k: DevArr(np.ascontiguousarray(pt[k].astype(np.uint32))) for k in ("send_left", "send_right", "recv_left", "recv_right")}
           for pt in parts]
    n01 = len(parts[0]["send_right"])
    n10 = len(parts[1]["send_left"])
    buf01 = DevArr(nbytes=n01 * pkg.abi.GHOST_BYTES)
    buf10 = DevArr(nbytes=n10 * pkg.abi.GHOST_BYTES)
    steps = 100
    for _ in range(steps):
        ctxs[0].halo_pack(ids[0]["send_right"].data_ptr(), n01, buf01.data_ptr())
        ctxs[1].halo_pack(ids[1]["send_left"].data_ptr(), n10, buf10.data_ptr())
        ctxs[0].sync(), ctxs[1].sync()
        ctxs[1].halo_unpack(ids[1]["recv_left"].data_ptr(), n01, buf01.data_ptr())
        ctxs[0].halo_unpack(ids[0]["recv_right"].data_ptr(), n10, buf10.data_ptr())
        ctxs[0].step(1), ctxs[1].step(1)
    X, V = gather_positions(pkg, parts, ctxs, p, sc.nOwnerClumps)
    one = make(p, sc)
    one.step(steps)
    st = one.download_state()
    n = sc.nOwnerClumps
    X1 = pkg.model.decode_positions(st["voxelID"], st["locX"], st["locY"], st["locZ"], p.nvXp2, p.nvYp2, p.voxelSize, p.l)[:n]
    assert one.counts().nContacts > 300
    assert np.abs(X - X1).max() < 2e-7


@pytest.mark.gpu
def test_overlapped_halo_steps_equal_plain_steps(pkg):
    """the split force passes + halo stream (deme_step_overlap_begin / _end) give bit-identical states to the plain
    pack -> unpack -> deme_step(1) loop, with detection every K steps (K = 7: detection and split steps alternate)"""
    import ctypes as C
    b, p, sc, x = build_global(pkg, n=3000, seed=6, cd_freq=7)
    parts = pkg.decomp.decompose(b.arrays, b.counts, x, 2, halo=0.035)
    hip = C.CDLL("libamdhip64.so")

    def dev(host=None, nbytes=0):
        ptr = C.c_void_p()
        nbytes = host.nbytes if host is not None else nbytes
        assert hip.hipMalloc(C.byref(ptr), C.c_size_t(max(nbytes, 16))) == 0
        if host is not None and host.nbytes:
            assert hip.hipMemcpy(ptr, C.c_void_p(host.ctypes.data), C.c_size_t(host.nbytes), 1) == 0
        return ptr.value

    def make(pp, s):
        ctx = pkg.Context(0)
        ctx.set_params(pp), ctx.upload_scene(s)
        return ctx

    ids = [{k: dev(np.ascontiguousarray(pt[k].astype(np.uint32))) for k in ("send_left", "send_right", "recv_left", "recv_right")}
           for pt in parts]
    n01, n10 = len(parts[0]["send_right"]), len(parts[1]["send_left"])
    buf01, buf10 = dev(nbytes=n01 * pkg.abi.GHOST_BYTES), dev(nbytes=n10 * pkg.abi.GHOST_BYTES)
    steps = 60
    plain = [make(p, pt["scene"]) for pt in parts]
    for _ in range(steps):
        plain[0].halo_pack(ids[0]["send_right"], n01, buf01), plain[1].halo_pack(ids[1]["send_left"], n10, buf10)
        plain[0].sync(), plain[1].sync()
        plain[1].halo_unpack(ids[1]["recv_left"], n01, buf01), plain[0].halo_unpack(ids[0]["recv_right"], n10, buf10)
        plain[0].step(1), plain[1].step(1)
    over = [make(p, pt["scene"]) for pt in parts]
    n_split = 0
    for _ in range(steps):
        due = [c.step_overlap_begin() for c in over]  # interior forces start here on the compute streams
        over[0].halo_sync(), over[1].halo_sync()  # the other context's unpack of the previous step has read my send buffer
        n_split += int(not due[0])
        over[0].halo_pack_async(ids[0]["send_right"], n01, buf01), over[1].halo_pack_async(ids[1]["send_left"], n10, buf10)
        over[0].halo_sync(), over[1].halo_sync()  # stands in for the send / recv between the two ranks
        over[1].halo_unpack_async(ids[1]["recv_left"], n01, buf01), over[0].halo_unpack_async(ids[0]["recv_right"], n10, buf10)
        over[0].step_overlap_end(), over[1].step_overlap_end()
    assert 40 < n_split < steps  # most steps were split, the detection steps were not
    for a, b_ in zip(plain, over):
        sa, sb = a.download_state(), b_.download_state()
        for k in GKEYS:
            assert np.array_equal(sa[k], sb[k]), k
        assert int(a.counts().nContacts) == int(b_.counts().nContacts) > 100
        assert np.array_equal(a.wildcard(3), b_.wildcard(3))


def _decode_x(pkg, p):
    def f(arrays):
        X = pkg.model.decode_positions(arrays["voxelID"], arrays["locX"], arrays["locY"], arrays["locZ"], p.nvXp2, p.nvYp2,
                                       p.voxelSize, p.l)
        return X[:, 0] + p.LBFX
    return f


def _run_with_redecomposition(pkg, make_sim, b, p, sc, x, steps_a, steps_b, n_ranks=2, halo=0.035):
    """steps_a steps on the initial decomposition, then a re-decomposition at the current positions (clumps and their
    contact history migrate) and steps_b more -- next to the SAME slabs simply continuing for steps_b (the drift is
    far below the halo, so the static decomposition is still exact).  The two must agree to rounding: both continue
    from bit-identical states, only the local numbering (hence the fp32 summation order) differs.  Comparing with a
    single-domain run instead would measure the chaotic growth of those rounding differences over hundreds of steps."""
    parts = pkg.decomp.decompose(b.arrays, b.counts, x, n_ranks, halo=halo)
    sims = run_slabs(pkg, make_sim, parts, p, steps_a, host_exchange(pkg, parts))
    nW = int(p.nContactWildcards)
    payloads = []
    for pt, s in zip(parts, sims):
        cnt = s.contacts()
        W = np.stack([s.wildcard(w) for w in range(nW)], 1) if nW else np.zeros((len(cnt[0]), 0), np.float32)
        payloads.append(pkg.decomp.owned_payload(pt, s.download_state(), cnt, W))
    g, parts2, seeds = pkg.decomp.redecompose(b.arrays, b.counts, payloads, n_ranks, halo, _decode_x(pkg, p))
    moved = sum(len(np.setdiff1d(a["global_ids"], b_["global_ids"])) for a, b_ in zip(parts2, parts))
    sims2 = []
    for pt, sd in zip(parts2, seeds):
        s = make_sim(p, pt["scene"])
        s.seed_contacts(*sd)
        sims2.append(s)
    ex, ex2 = host_exchange(pkg, parts), host_exchange(pkg, parts2)
    for _ in range(steps_b):
        ex(sims), ex2(sims2)
        for s in sims + sims2:
            s.step(1)
    X, V = gather_positions(pkg, parts2, sims2, p, sc.nOwnerClumps)
    X0, V0 = gather_positions(pkg, parts, sims, p, sc.nOwnerClumps)
    return X, V, X0, V0, moved, seeds


def _sheared_bed(pkg, n, seed):
    # neighbouring clumps start with different x-velocities: the bed shears and clumps cross the cut
    b, p, sc, x = build_global(pkg, n=n, seed=seed)
    nc = int(sc.nOwnerClumps)
    b.arrays["vX"][:nc] = np.where(np.arange(nc) % 2 == 0, 0.6, 0.3).astype(np.float32)
    return b, p, pkg.abi.make_scene_struct(b.arrays, b.counts), x


def test_redecomposition_migrates_clumps_and_history_oracle(pkg, orc):
    b, p, sc, x = _sheared_bed(pkg, 1200, 9)
    X, V, X0, V0, moved, seeds = _run_with_redecomposition(pkg, lambda pp, s: orc.make_sim(pkg, pp, s), b, p, sc, x, 400, 20)
    assert moved > 3  # ownership really changed hands
    assert sum(len(s[0]) for s in seeds) > 200 and any((s[2] == 1).any() for s in seeds)
    # measured: 1.5e-12 m / 3e-8 m/s with the history carried; 5.7e-6 m / 0.09 m/s when it is dropped, 1.6e-6 m /
    # 0.02 m/s when the B-to-A vector wildcards of mirrored pairs are not negated
    assert np.abs(X - X0).max() < 1e-9 and np.abs(V - V0).max() < 1e-5


@pytest.mark.gpu
def test_redecomposition_on_gpu(pkg):
    b, p, sc, x = _sheared_bed(pkg, 3000, 6)

    def make(pp, s):
        ctx = pkg.Context(0)
        ctx.set_params(pp)
        ctx.upload_scene(s)
        return ctx

    X, V, X0, V0, moved, seeds = _run_with_redecomposition(pkg, make, b, p, sc, x, 400, 20)
    assert moved > 5
    assert np.abs(X - X0).max() < 1e-9 and np.abs(V - V0).max() < 1e-5


def _bed_on_moving_plate(pkg, n=1600, seed=4, cd_freq=0):
    """a bed over a wavy triangle plate that rises with a prescribed velocity (family 10): the mesh is replicated on both slabs"""
    b = pkg.model.packed_bed(n, seed=seed, cd_freq=cd_freq, spacing_mult=2.5, init_vz=-0.4, aspect=(2.0, 1.0, 0.5))
    b.SetExpandSafetyAdder(0.5)
    lo, hi = b.user_box_min, b.user_box_max
    v, f = pkg.model.plate_mesh(24, 12, float(hi[0] - lo[0]) * 0.95, float(hi[1] - lo[1]) * 0.95, z=0.0, wavy=0.002)
    m = b.AddMeshObject(v, f, 0)
    m.SetInitPos(((lo[0] + hi[0]) / 2, (lo[1] + hi[1]) / 2, 0.0165))
    m.SetFamily(10)
    b.SetFamilyPrescribedLinVel(10, "0", "0", "0.3f")
    p, sc = b.Initialize()
    x = np.concatenate([bb.xyz for bb in b.batches])[:, 0]
    return b, p, sc, x


def _plate_prescription(sim):
    c = np.zeros((15, 4), np.float32)
    c[2] = (0.3, 0, 0, 0)
    sim.set_prescription(10, has=0b111, flags=0b111111, coef=c)


def test_two_slabs_with_a_replicated_prescribed_mesh_oracle(pkg, orc):
    b, p, sc, x = _bed_on_moving_plate(pkg)
    parts = pkg.decomp.decompose(b.arrays, b.counts, x, 2, halo=0.035)
    assert all(int(pt["counts"]["nTri"]) == int(sc.nTri) for pt in parts)

    def mk(pp, s):
        sim = orc.make_sim(pkg, pp, s)
        _plate_prescription(sim)
        return sim

    steps = 120
    sims = run_slabs(pkg, mk, parts, p, steps, host_exchange(pkg, parts))
    X, V = gather_positions(pkg, parts, sims, p, sc.nOwnerClumps)
    one = mk(p, sc)
    one.step(steps)
    st = one.download_state()
    n = sc.nOwnerClumps
    X1 = pkg.model.decode_positions(st["voxelID"], st["locX"], st["locY"], st["locZ"], p.nvXp2, p.nvYp2, p.voxelSize, p.l)[:n]
    assert (one.contacts()[2] == 2).sum() > 30  # sphere-mesh contacts exist
    assert np.abs(X - X1).max() < 2e-7
    # a free mesh cannot be replicated
    b2, p2, sc2, x2 = _bed_on_moving_plate(pkg)
    b2.family_flags[10] = 0
    b2.arrays["familyFlags"][10] = 0
    with pytest.raises(ValueError):
        pkg.decomp.decompose(b2.arrays, b2.counts, x2, 2, halo=0.035)


@pytest.mark.gpu
def test_two_slabs_with_mesh_on_gpu_plain_and_overlapped(pkg):
    import ctypes as C
    b, p, sc, x = _bed_on_moving_plate(pkg, n=3000, seed=6, cd_freq=5)  # detection every 5 steps: 4 of 5 steps are split
    parts = pkg.decomp.decompose(b.arrays, b.counts, x, 2, halo=0.035)
    hip = C.CDLL("libamdhip64.so")

    def dev(host=None, nbytes=0):
        ptr = C.c_void_p()
        nbytes = host.nbytes if host is not None else nbytes
        assert hip.hipMalloc(C.byref(ptr), C.c_size_t(max(nbytes, 16))) == 0
        if host is not None and host.nbytes:
            assert hip.hipMemcpy(ptr, C.c_void_p(host.ctypes.data), C.c_size_t(host.nbytes), 1) == 0
        return ptr.value

    def make(pp, s):
        ctx = pkg.Context(0)
        ctx.set_params(pp), ctx.upload_scene(s)
        b.compile_into(ctx)  # the plate's prescription
        return ctx

    ids = [{k: dev(np.ascontiguousarray(pt[k].astype(np.uint32))) for k in ("send_left", "send_right", "recv_left", "recv_right")}
           for pt in parts]
    n01, n10 = len(parts[0]["send_right"]), len(parts[1]["send_left"])
    buf01, buf10 = dev(nbytes=n01 * pkg.abi.GHOST_BYTES), dev(nbytes=n10 * pkg.abi.GHOST_BYTES)
    steps = 120
    plain = [make(p, pt["scene"]) for pt in parts]
    for _ in range(steps):
        plain[0].halo_pack(ids[0]["send_right"], n01, buf01), plain[1].halo_pack(ids[1]["send_left"], n10, buf10)
        plain[0].sync(), plain[1].sync()
        plain[1].halo_unpack(ids[1]["recv_left"], n01, buf01), plain[0].halo_unpack(ids[0]["recv_right"], n10, buf10)
        plain[0].step(1), plain[1].step(1)
    over = [make(p, pt["scene"]) for pt in parts]
    n_split = 0
    for _ in range(steps):
        n_split += sum(int(not c.step_overlap_begin()) for c in over)
        over[0].halo_sync(), over[1].halo_sync()
        over[0].halo_pack_async(ids[0]["send_right"], n01, buf01), over[1].halo_pack_async(ids[1]["send_left"], n10, buf10)
        over[0].halo_sync(), over[1].halo_sync()
        over[1].halo_unpack_async(ids[1]["recv_left"], n01, buf01), over[0].halo_unpack_async(ids[0]["recv_right"], n10, buf10)
        over[0].step_overlap_end(), over[1].step_overlap_end()
    one = make(p, sc)
    one.step(steps)
    st = one.download_state()
    n = sc.nOwnerClumps
    X1 = pkg.model.decode_positions(st["voxelID"], st["locX"], st["locY"], st["locZ"], p.nvXp2, p.nvYp2, p.voxelSize, p.l)[:n]
    assert (one.contacts()[2] == 2).sum() > 30
    X, _ = gather_positions(pkg, parts, plain, p, n)
    assert np.abs(X - X1).max() < 2e-7
    assert n_split > steps
    for a, c in zip(plain, over):
        sa, sb = a.download_state(), c.download_state()
        for k in GKEYS:
            assert np.array_equal(sa[k], sb[k]), k


@pytest.mark.gpu
def test_overlapped_exchange_through_torch_streams():
    """bench.py's stream / event choreography with torch ExternalStreams around the contexts' halo streams (torch copies stand
    in for RCCL isend / irecv): bit-identical to the ordered exchange"""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_overlap_torch_worker.py")], capture_output=True, text=True,
                         timeout=900)
    assert out.returncode == 0 and "OVERLAP_TORCH_OK" in out.stdout, out.stdout[-1500:] + out.stderr[-1500:]


@pytest.mark.gpu
def test_overlapped_exchange_through_rccl_on_one_gpu():
    """the same two-slab run with the ghost records travelling through RCCL itself (torch.distributed, backend nccl): a
    world-size-1 communicator matches a send to self with the receive posted in the same group, so slab 0's records reach slab
    1's buffer and vice versa -- isend / irecv / batch / wait on the context's halo stream (a torch ExternalStream), exactly the
    calls bench.py makes per step; bit-identical to the ordered exchange"""
    env = dict(os.environ, DEME_OVERLAP_TEST_RCCL="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_overlap_torch_worker.py")], capture_output=True, text=True,
                         timeout=900, env=env)
    assert out.returncode == 0 and "OVERLAP_TORCH_OK" in out.stdout and "through RCCL" in out.stdout, out.stdout[-1500:] + out.stderr[-1500:]


def test_redecomposition_carries_wildcard_arrays_and_persistent_marks(pkg, orc):
    """owner / sphere wildcards of a user model and the persistent-contact marks move with the clumps: values tagged with the
    global ids come out under the new local numbering (ghost copies included), and the union of the ranks' marked pairs is the
    same set of global pairs before and after"""
    b, p, sc, x = _sheared_bed(pkg, 1200, 9)
    parts = pkg.decomp.decompose(b.arrays, b.counts, x, 2, halo=0.035)
    sims = run_slabs(pkg, lambda pp, s: orc.make_sim(pkg, pp, s), parts, p, 400, host_exchange(pkg, parts))
    nW = int(p.nContactWildcards)
    n_clumps = int(sc.nOwnerClumps)

    def global_pairs(part, trip):
        a, bb, t = trip
        sg = part["sphere_global"]
        ga = sg[a]
        gb = np.where(t == 1, sg[np.where(t == 1, bb, 0)], bb.astype(np.int64))
        lo, hi = np.where((t == 1) & (ga > gb), gb, ga), np.where((t == 1) & (ga > gb), ga, gb)
        own = np.asarray(part["arrays"]["ownerClumpBody"], np.int64)
        touches_own = (own[a] < part["n_own"]) | ((t == 1) & (own[np.where(t == 1, bb, 0)] < part["n_own"]))
        return set(zip(lo[touches_own].tolist(), hi[touches_own].tolist(), t[touches_own].tolist()))

    payloads, before = [], set()
    for pt, s in zip(parts, sims):
        s.mark_persistent_contacts()  # every contact of the current list
        cnt = s.contacts()
        W = np.stack([s.wildcard(w) for w in range(nW)], 1)
        n_o, n_s = int(pt["counts"]["nOwners"]), int(pt["counts"]["nSpheres"])
        n_c = int(pt["counts"]["nOwnerClumps"])
        ow = np.r_[pt["owner_global"][:n_c] + 0.5, np.full(n_o - n_c, -7.0)].astype(np.float32)  # tag = global id + 0.5
        sw = (pt["sphere_global"] + 0.25).astype(np.float32)
        payloads.append(pkg.decomp.owned_payload(pt, s.download_state(), cnt, W, persistent=s.persistent_contacts(),
                                                 owner_wildcards={"tag": ow}, sphere_wildcards={"stag": sw}))
        before |= global_pairs(pt, s.persistent_contacts())
    assert len(before) > 200
    g, parts2, seeds = pkg.decomp.redecompose(b.arrays, b.counts, payloads, 2, 0.035, _decode_x(pkg, p))
    assert sum(len(np.setdiff1d(a["global_ids"], b_["global_ids"])) for a, b_ in zip(parts2, parts)) > 3
    after = set()
    for pt, sd in zip(parts2, seeds):
        n_c = int(pt["counts"]["nOwnerClumps"])
        assert np.array_equal(pt["owner_wc"]["tag"][:n_c], (pt["owner_global"][:n_c] + 0.5).astype(np.float32))  # own + ghosts
        assert (pt["owner_wc"]["tag"][n_c:] == -7.0).all()  # replicated owners keep the rank's own values
        assert np.array_equal(pt["sphere_wc"]["stag"], (pt["sphere_global"] + 0.25).astype(np.float32))
        s = orc.make_sim(pkg, p, pt["scene"])
        s.seed_contacts(*sd)
        s.set_persistent_contacts(*pt["persistent"])
        assert s.num_persistent_contacts() == len(pt["persistent"][0]) > 50
        after |= global_pairs(pt, s.persistent_contacts())
        s.step(3)  # the marks are live: every marked pair is in the list the next detection builds
        a, bb, t, _ = s.contacts()
        listed = set(zip(a.tolist(), bb.tolist(), t.tolist()))
        assert set(zip(*[x.tolist() for x in s.persistent_contacts()])) <= listed
    assert after == before


def _state_x(pkg, p):
    def f(st):
        X = pkg.model.decode_positions(st["voxelID"], st["locX"], st["locY"], st["locZ"], p.nvXp2, p.nvYp2, p.voxelSize, p.l)
        return X[:, 0] + p.LBFX
    return f


def _migration_run(pkg, make_sim, n_ranks, steps_a, steps_b, halo=0.035):
    """steps_a steps on the initial slabs of a sheared bed, then a neighbour-to-neighbour migration (fixed edges: crossed clumps
    move to the face neighbour with state, template data and contact history; ghost sets rebuilt from neighbour packets) and
    steps_b more -- next to the SAME slabs simply continuing (the drift stays far below the halo, so they remain exact)"""
    b, p, sc, x = _sheared_bed(pkg, 1600, 4)
    parts = pkg.decomp.decompose(b.arrays, b.counts, x, n_ranks, halo=halo)
    edges = parts[0]["all_edges"]
    sims = run_slabs(pkg, make_sim, parts, p, steps_a, host_exchange(pkg, parts))
    nW = int(p.nContactWildcards)
    states = [s.download_state() for s in sims]
    cnts = [s.contacts() for s in sims]
    Ws = [np.stack([s.wildcard(w) for w in range(nW)], 1) for s in sims]
    parts2, seeds = pkg.decomp.migrate_neighbours_in_process(parts, states, cnts, Ws, edges, halo, _state_x(pkg, p))
    moved = sum(len(np.setdiff1d(a["global_ids"], b_["global_ids"])) for a, b_ in zip(parts2, parts))
    sims2 = []
    for pt, sd in zip(parts2, seeds):
        s = make_sim(p, pt["scene"])
        s.seed_contacts(*sd)
        sims2.append(s)
    ex, ex2 = host_exchange(pkg, parts), host_exchange(pkg, parts2)
    for _ in range(steps_b):
        ex(sims), ex2(sims2)
        for s in sims + sims2:
            s.step(1)
    X, V = gather_positions(pkg, parts2, sims2, p, sc.nOwnerClumps)
    X0, V0 = gather_positions(pkg, parts, sims, p, sc.nOwnerClumps)
    return parts, parts2, X, V, X0, V0, moved, sc


def test_neighbour_migration_oracle(pkg, orc):
    parts, parts2, X, V, X0, V0, moved, sc = _migration_run(pkg, lambda pp, s: orc.make_sim(pkg, pp, s), 3, 150, 20)
    assert moved >= 3, moved  # clumps really changed rank
    n = int(sc.nOwnerClumps)
    own = np.sort(np.concatenate([pt["global_ids"] for pt in parts2]))
    assert np.array_equal(own, np.arange(n))  # every clump owned exactly once afterwards
    for r, pt in enumerate(parts2):  # ownership follows the fixed edges; ghost lists pair up
        lo, hi = pt["edges"]
        if r + 1 < len(parts2):
            nb = parts2[r + 1]
            assert np.array_equal(pt["global_ids"][pt["send_right"]], nb["ghost_left_g"])
            assert np.array_equal(nb["global_ids"][nb["send_left"]], pt["ghost_right_g"])
        assert pt["arrays"]["ownerGhost"][pt["n_own"]:pt["counts"]["nOwnerClumps"]].all()
    # the history travelled: with it the two runs agree to fp32 summation order (different local numbering); without it a
    # migrated clump's tangential springs would restart from zero (1e-6 m after 20 steps in the all-gather twin of this test)
    assert np.abs(X - X0).max() < 1e-9 and np.abs(V - V0).max() < 1e-5


def test_migration_carries_the_current_state_of_moving_replicated_owners(pkg, orc):
    """a bed over a plate that rises with a prescribed velocity, in two slabs: after 100 steps the slabs are re-assembled by the
    neighbour migration and by the all-gather re-decomposition -- the plate (a replicated owner, the same on every slab) must keep
    the pose and the velocity it HAS, not the ones of the first decomposition (it used to jump back), and the run must carry on
    like the slabs that simply continue"""
    b, p, sc, x = _bed_on_moving_plate(pkg, n=900)
    halo = 0.035
    parts = pkg.decomp.decompose(b.arrays, b.counts, x, 2, halo=halo)
    edges = parts[0]["all_edges"]

    def mk(pp, s):
        sim = orc.make_sim(pkg, pp, s)
        _plate_prescription(sim)
        return sim

    sims = run_slabs(pkg, mk, parts, p, 100, host_exchange(pkg, parts))
    nW = int(p.nContactWildcards)
    states = [s.download_state() for s in sims]
    cnts = [s.contacts() for s in sims]
    Ws = [np.stack([s.wildcard(w) for w in range(nW)], 1) for s in sims]
    parts2, seeds = pkg.decomp.migrate_neighbours_in_process(parts, states, cnts, Ws, edges, halo, _state_x(pkg, p))
    plate_old = int(parts[0]["counts"]["nOwners"]) - 1   # replicated owners sit behind the clumps: the plate is the last one
    z_now = pkg.model.decode_positions(states[0]["voxelID"], states[0]["locX"], states[0]["locY"], states[0]["locZ"], p.nvXp2, p.nvYp2,
                                       p.voxelSize, p.l)[plate_old, 2]
    z_start = pkg.model.decode_positions(parts[0]["arrays"]["voxelID"], parts[0]["arrays"]["locX"], parts[0]["arrays"]["locY"],
                                         parts[0]["arrays"]["locZ"], p.nvXp2, p.nvYp2, p.voxelSize, p.l)[plate_old, 2]
    assert z_now - z_start > 1e-4  # 100 steps at 0.3 m/s
    for pt in parts2:
        a = pt["arrays"]
        plate = int(pt["counts"]["nOwners"]) - 1
        z_new = pkg.model.decode_positions(a["voxelID"], a["locX"], a["locY"], a["locZ"], p.nvXp2, p.nvYp2, p.voxelSize, p.l)[plate, 2]
        assert abs(z_new - z_now) < 1e-12 and a["vZ"][plate] == states[0]["vZ"][plate_old] != 0.0
    sims2 = []
    for pt, sd in zip(parts2, seeds):
        s = mk(p, pt["scene"])
        s.seed_contacts(*sd)
        sims2.append(s)
    ex, ex2 = host_exchange(pkg, parts), host_exchange(pkg, parts2)
    for _ in range(20):
        ex(sims), ex2(sims2)
        for s in sims + sims2:
            s.step(1)
    X, V = gather_positions(pkg, parts2, sims2, p, sc.nOwnerClumps)
    X0, V0 = gather_positions(pkg, parts, sims, p, sc.nOwnerClumps)
    assert np.abs(X - X0).max() < 1e-8 and np.abs(V - V0).max() < 1e-4


# ---- the decomposition inside the library (deme_decomp_*: what a C++ host calls) against the numpy statement above ----------------
def _scene_arrays(sc, name, dtype, n):
    import ctypes as C
    ptr = getattr(sc, name)
    if not ptr or not n:
        return np.zeros(0, dtype)
    return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(np.ctypeslib.as_ctypes_type(dtype))), shape=(int(n),)).copy()


@pytest.mark.parametrize("n_ranks", [2, 3, 5])
def test_library_decomposition_equals_the_numpy_statement(pkg, n_ranks):
    """deme_decomp_create (csrc/deme_decomp.inc; pure host code, runs without a device) cuts the same scene at the same boundaries into
    the same slabs as decomp.decompose: owners, ghosts, sphere lists, renumbered references, exchange lists, per-owner arrays."""
    b, p, sc, _ = build_global(pkg)
    halo = 0.03
    # (the library reads the clumps' coordinates from the ENCODED positions of the scene -- what the kernels see)
    x = pkg.model.decode_positions(b.arrays["voxelID"], b.arrays["locX"], b.arrays["locY"], b.arrays["locZ"], p.nvXp2, p.nvYp2, p.voxelSize,
                                   p.l)[:, 0] + float(p.LBFX)
    ref = pkg.decomp.decompose(b.arrays, b.counts, x, n_ranks, halo=halo)
    plan, parts = pkg.decomp.decompose_lib(p, sc, n_ranks, halo, axis=0, edges=ref[0]["all_edges"])
    assert plan.axis == 0 and plan.halo == halo and plan.n_slabs == n_ranks
    for r in range(n_ranks):
        a, q = ref[r], parts[r]
        assert q["n_own"] == a["n_own"]
        for k in ("global_ids", "ghost_left_g", "ghost_right_g", "owner_global", "sphere_global", "send_left", "send_right", "recv_left",
                  "recv_right"):
            assert np.array_equal(np.asarray(q[k], np.int64), np.asarray(a[k], np.int64)), (r, k)
        s1, s0 = q["scene"], a["scene"]
        for k in ("nOwners", "nOwnerClumps", "nSpheres", "nAnal", "nTri", "nMat", "nComp", "nMassProps"):
            assert getattr(s1, k) == getattr(s0, k), (r, k)
        for k, dt in pkg.abi.SCENE_DTYPES.items():
            n = {"ownerClumpBody": s0.nSpheres, "clumpComponentOffset": s0.nSpheres, "sphereMaterialOffset": s0.nSpheres,
                 "objOwner": s0.nAnal, "ownerMesh": s0.nTri, "ownerGhost": s0.nOwners}.get(k)
            if n is None and k in pkg.decomp._OWNER_KEYS:
                n = s0.nOwners
            if n is None:
                continue  # tables shared with the global scene
            assert np.array_equal(_scene_arrays(s1, k, dt, n), _scene_arrays(s0, k, dt, n)), (r, k)


def test_library_decomposition_picks_axis_halo_and_bin_aligned_cuts(pkg):
    """defaults of a C++ caller: axis -1 = the longest side of the clumps' bounding box, halo 0 = four clump reaches, equal-count cuts
    snapped to bin faces; every clump owned once, every ghost within the halo of a face, counts balanced to within one bin layer"""
    b, p, sc, x = build_global(pkg)  # aspect (2, 1, 0.5): x is the long side
    plan, parts = pkg.decomp.decompose_lib(p, sc, 4, halo=0.0, axis=-1, snap=True)
    assert plan.axis == 0
    reach = max(np.hypot(np.hypot(b.arrays["CDRelPosX"], b.arrays["CDRelPosY"]), b.arrays["CDRelPosZ"]) + b.arrays["Radii"])
    assert abs(plan.halo - 4.0 * float(reach)) < 1e-6 * reach
    inner = plan.edges[1:-1]
    k = (inner - float(p.LBFX)) / float(p.binSize)
    assert np.abs(k - np.rint(k)).max() < 1e-9, "cuts must lie on bin faces"
    assert np.isinf(plan.edges[0]) and np.isinf(plan.edges[-1])
    own = np.sort(np.concatenate([q["global_ids"] for q in parts]))
    assert np.array_equal(own, np.arange(sc.nOwnerClumps))
    xq = pkg.model.decode_positions(b.arrays["voxelID"], b.arrays["locX"], b.arrays["locY"], b.arrays["locZ"], p.nvXp2, p.nvYp2, p.voxelSize,
                                    p.l)[:sc.nOwnerClumps, 0] + float(p.LBFX)
    for r, q in enumerate(parts):
        lo, hi = q["edges"]
        assert ((xq[q["global_ids"]] >= lo) & (xq[q["global_ids"]] < hi)).all()
        assert (xq[q["ghost_left_g"]] >= lo - plan.halo).all() and (xq[q["ghost_left_g"]] < lo).all()
        assert (xq[q["ghost_right_g"]] < hi + plan.halo).all() and (xq[q["ghost_right_g"]] >= hi).all()
    n = np.array([q["n_own"] for q in parts])
    assert n.min() > 0.6 * n.mean(), n
    # a rotated bed: z is the long side
    b2 = pkg.model.packed_bed(1200, seed=4, cd_freq=0, spacing_mult=2.5, aspect=(0.5, 0.5, 3.0))
    p2, sc2 = b2.Initialize()
    plan2, _ = pkg.decomp.decompose_lib(p2, sc2, 2, halo=0.0, axis=-1)
    assert plan2.axis == 2


def test_library_decomposition_refusals(pkg):
    b, p, sc, x = build_global(pkg)
    with pytest.raises(pkg.abi.DemeError, match="thinner than the halo"):
        pkg.decomp.decompose_lib(p, sc, 12, halo=0.2, axis=0)
    own = sc._keep["ownerClumpBody"]
    own[[0, len(own) - 1]] = own[[len(own) - 1, 0]]
    with pytest.raises(pkg.abi.DemeError, match="clump-major"):
        pkg.decomp.decompose_lib(p, sc, 2, halo=0.03, axis=0)


def test_multi_refuses_absent_devices(pkg):
    """DEMSolver(device ids) -> deme_multi_create: an id that is not among the visible devices is refused with a message that names it
    (the reference throws in GpuManager.cpp:64-68); here, without a GPU, every id is absent"""
    n = pkg.abi.device_count()
    with pytest.raises(pkg.abi.DemeError, match=f"device id {n + 3} is not present"):
        pkg.abi.Multi(devices=(n + 3,))
    with pytest.raises(pkg.abi.DemeError, match="at least one device"):
        pkg.abi.Multi(devices=())


def test_library_decomposition_in_the_engines_order(pkg):
    """DEME_DECOMP_SPATIAL_ORDER (what deme_multi_build asks for): the same slabs -- owners, ghosts, boundaries -- with every slab's own
    clumps numbered along the engine's own order instead of by global id; ghosts follow their owner slab's order, so the exchange lists
    still name the same clumps in the same order on both sides and the sender's list ascends.  On a bed handed over in random order
    the owner tiles of a slab (128 consecutive clumps) become compact."""
    b = pkg.model.packed_bed(6000, seed=5, cd_freq=0, spacing_mult=3.0, init_vz=-1.0, aspect=(2.0, 1.0, 0.25), order="random")
    p, sc = b.Initialize()
    plan0, ref = pkg.decomp.decompose_lib(p, sc, 3, 0.03, axis=0)
    plan1, got = pkg.decomp.decompose_lib(p, sc, 3, 0.03, axis=0, spatial_order=True)
    assert np.array_equal(plan0.edges, plan1.edges)
    X = pkg.model.decode_positions(b.arrays["voxelID"], b.arrays["locX"], b.arrays["locY"], b.arrays["locZ"], p.nvXp2, p.nvYp2, p.voxelSize, p.l)

    def tile_extent(gids):  # mean bounding-box diagonal of the 128-clump tiles
        ext = []
        for k in range(0, len(gids) - 127, 128):
            q = X[gids[k:k + 128]]
            ext.append(np.linalg.norm(q.max(0) - q.min(0)))
        return float(np.mean(ext))
    for r, (a, q) in enumerate(zip(ref, got)):
        for k in ("global_ids", "ghost_left_g", "ghost_right_g"):
            assert np.array_equal(np.sort(a[k]), np.sort(q[k])), (r, k)  # the same clumps ...
        assert not np.array_equal(a["global_ids"], q["global_ids"])      # ... in another order
        assert tile_extent(q["global_ids"]) < 0.5 * tile_extent(a["global_ids"]), (tile_extent(q["global_ids"]), tile_extent(a["global_ids"]))
        assert (np.diff(q["send_left"].astype(np.int64)) > 0).all() and (np.diff(q["send_right"].astype(np.int64)) > 0).all()
        if r + 1 < len(got):
            nb = got[r + 1]
            assert np.array_equal(q["global_ids"][q["send_right"]], nb["ghost_left_g"])
            assert np.array_equal(nb["global_ids"][nb["send_left"]], q["ghost_right_g"])
        sc1 = q["scene"]
        own = _scene_arrays(sc1, "ownerClumpBody", np.uint32, sc1.nSpheres)
        assert (np.diff(own.astype(np.int64)) >= 0).all()  # spheres stay clump-major in the slab's numbering
        vid = _scene_arrays(sc1, "voxelID", np.uint64, sc1.nOwners)
        assert np.array_equal(vid, b.arrays["voxelID"][q["owner_global"]])


@pytest.mark.parametrize("seed,n_slabs,axis,order,spatial", [(11, 2, 0, "random", True), (12, 4, 1, "lattice", False), (13, 7, -1, "morton", True),
                                                              (14, 3, 2, "random", False), (15, 5, -1, "random", True)])
def test_library_plans_keep_their_invariants_on_random_beds(pkg, seed, n_slabs, axis, order, spatial):
    """Properties every plan of deme_decomp_create has to have, whatever the bed, the axis, the slab count and the numbering: every
    clump owned by exactly one slab, inside that slab's range; the ghosts of a slab are exactly the neighbours' own clumps within the
    halo of the shared face; what a slab sends is what its neighbour receives, in the same order, and the sender's list ascends;
    replicated owners are on every slab; spheres stay clump-major; boundaries ascend and sit on bin faces."""
    rng = np.random.default_rng(seed)
    aspect = tuple(float(v) for v in rng.permutation([2.0, 1.0, 0.5]))
    b = pkg.model.packed_bed(int(rng.integers(1500, 4000)), seed=seed, cd_freq=0, spacing_mult=float(rng.uniform(2.2, 3.2)), init_vz=-0.5,
                             aspect=aspect, order=order)
    p, sc = b.Initialize()
    nc, no = int(sc.nOwnerClumps), int(sc.nOwners)
    halo = float(rng.uniform(0.02, 0.04))
    try:
        plan, parts = pkg.decomp.decompose_lib(p, sc, n_slabs, halo, axis=axis, snap=True, spatial_order=spatial)
    except pkg.abi.DemeError as e:  # (a bed too short along the chosen axis for that many slabs of at least a halo: a refusal, with the reason)
        assert "halo" in str(e) or "thinner" in str(e) or "slab" in str(e), str(e)
        return
    ax = plan.axis
    assert 0 <= ax <= 2 and (axis < 0 or ax == axis)
    X = pkg.model.decode_positions(b.arrays["voxelID"], b.arrays["locX"], b.arrays["locY"], b.arrays["locZ"], p.nvXp2, p.nvYp2, p.voxelSize, p.l)
    lbf = (float(p.LBFX), float(p.LBFY), float(p.LBFZ))[ax]
    x = X[:nc, ax] + lbf
    e = np.asarray(plan.edges)
    assert len(e) == n_slabs + 1 and np.isneginf(e[0]) and np.isposinf(e[-1]) and (np.diff(e[1:-1]) > 0).all()
    inner = (e[1:-1] - lbf) / float(p.binSize)
    assert np.abs(inner - np.rint(inner)).max() < 1e-9
    owned = np.concatenate([q["global_ids"] for q in parts])
    assert len(owned) == nc and np.array_equal(np.sort(owned), np.arange(nc))
    for r, q in enumerate(parts):
        own = np.asarray(q["global_ids"], np.int64)
        assert ((x[own] >= e[r]) & (x[own] < e[r + 1])).all()
        og = np.asarray(q["owner_global"], np.int64)
        n_c = q["n_own"] + len(q["ghost_left_g"]) + len(q["ghost_right_g"])
        assert np.array_equal(np.sort(og[n_c:]), np.arange(nc, no))  # the replicated owners, all of them
        if r > 0:
            want = np.asarray(parts[r - 1]["global_ids"], np.int64)
            want = want[x[want] >= e[r] - halo]
            assert np.array_equal(np.sort(q["ghost_left_g"]), np.sort(want))
            assert np.array_equal(np.asarray(parts[r - 1]["global_ids"])[parts[r - 1]["send_right"]], q["ghost_left_g"])
        if r + 1 < n_slabs:
            want = np.asarray(parts[r + 1]["global_ids"], np.int64)
            want = want[x[want] < e[r + 1] + halo]
            assert np.array_equal(np.sort(q["ghost_right_g"]), np.sort(want))
            assert np.array_equal(np.asarray(parts[r + 1]["global_ids"])[parts[r + 1]["send_left"]], q["ghost_right_g"])
        for k in ("send_left", "send_right"):
            assert (np.diff(np.asarray(q[k], np.int64)) > 0).all()
        s1 = q["scene"]
        sph_owner = _scene_arrays(s1, "ownerClumpBody", np.uint32, s1.nSpheres).astype(np.int64)
        assert (np.diff(sph_owner) >= 0).all() and (sph_owner < s1.nOwners).all()
        # a sphere of the slab is the global sphere it names: same owner (through the owners' global ids), same component
        g_owner = np.asarray(b.arrays["ownerClumpBody"], np.int64)[np.asarray(q["sphere_global"], np.int64)]
        assert np.array_equal(og[sph_owner], g_owner)
