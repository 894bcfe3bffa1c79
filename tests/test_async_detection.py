"""deme_set_async_detection: part 1 of each contact detection (incidences, sorts, bin sweep, key sort) runs D steps ahead of the swap
on its own stream, from a copy of the owner records, beside the force / integration kernels of those D steps -- what the reference's
kT does beside dT (kT.cpp:100-216, dT.cpp:1955-2038).  The list it builds has margins for K + D steps, so it holds every pair the
lock-step list holds plus near-pairs whose contributions are zero: in the exact arithmetic mode the trajectory is the lock-step one,
which the oracle pins."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KEYS = ("voxelID", "locX", "locY", "locZ", "oriQw", "oriQx", "oriQy", "oriQz", "vX", "vY", "vZ", "omgBarX", "omgBarY", "omgBarZ")


def _bed(pkg, n, K):
    b = pkg.model.packed_bed(n, seed=12, cd_freq=K, spacing_mult=3.0, init_vz=-1.0, aspect=(1.0, 1.0, 0.5))
    # Margins are sized from each owner's speed at the time of the detection plus a safety velocity (the reference's default adds
    # 3 m/s, API.h:1484).  Without it a clump that is struck between two detections can reach a partner its margin did not
    # cover -- in the lock-step loop as well, for up to K steps; the asynchronous list is K + D steps old when it retires.
    b.SetExpandSafetyAdder(1.0)
    p, sc = b.Initialize()
    return b, p, sc


@pytest.mark.parametrize("n,K,D,settle,steps", [(3000, 20, 6, 4000, 1203), (20000, 40, 10, 6000, 1611)])
def test_async_detection_trajectory_is_the_lock_step_one(pkg, orc, n, K, D, settle, steps):
    """a bed collapsing onto the floor (contacts forming and breaking all the time): `settle` steps lock-step in both contexts, then
    `steps` more with the asynchronous detection in one of them"""
    b, p, sc = _bed(pkg, n, K)
    lock, asyn = pkg.Context(0), pkg.Context(0)
    for c in (lock, asyn):
        c.set_arith_mode("exact")
        c.set_params(p), c.upload_scene(sc)
    one = orc.make_sim(pkg, p, sc)
    lock.step(settle), asyn.step(settle)
    asyn.set_async_detection(D)
    lock.step(steps), asyn.step(steps), one.step(settle + steps)
    s_lock, s_asyn, s_one = lock.download_state(), asyn.download_state(), one.download_state()
    for k in KEYS:
        assert np.array_equal(s_asyn[k], s_lock[k]), k   # (array_equal: -0.0 == 0.0)
        assert np.array_equal(s_asyn[k], s_one[k]), k    # and the oracle's lock-step run
    # the asynchronous lists were really used (built from older positions with wider margins they differ from the lock-step ones),
    # and every pair that has ever touched is in both with the same history
    wl = {q: w for q, w in zip(zip(*[x.tolist() for x in lock.contacts()[:3]]), lock.wildcard(3).tolist())}
    wa = {q: w for q, w in zip(zip(*[x.tolist() for x in asyn.contacts()[:3]]), asyn.wildcard(3).tolist())}
    assert set(wl) != set(wa) and len(wa) > len(wl) > n // 10
    common = [q for q in wl if q in wa]
    assert sum(1 for q in common if wl[q] > 0.0) > n // 40 and all(wl[q] == wa[q] for q in common)
    assert all(wl[q] == 0.0 for q in wl if q not in wa) and all(wa[q] == 0.0 for q in wa if q not in wl)


def test_async_detection_falls_back_when_it_cannot_run(pkg):
    """calls shorter than the lead, or K <= D: the lock-step detection serves (same results, no error)"""
    b, p, sc = _bed(pkg, 2000, 8)
    a, c = pkg.Context(0), pkg.Context(0)
    for x in (a, c):
        x.set_arith_mode("exact")
        x.set_params(p), x.upload_scene(sc)
    c.set_async_detection(12)  # K = 8 <= D
    a.step(50), c.step(50)
    assert np.array_equal(a.download_state()["locZ"], c.download_state()["locZ"])
    c.set_async_detection(3)
    for _ in range(40):  # one step per call: never 3 steps left in a call
        a.step(1), c.step(1)
    assert np.array_equal(a.download_state()["locZ"], c.download_state()["locZ"])


def test_async_detection_with_a_mesh_in_the_bed(pkg, orc):
    """a bed falling onto a fixed, wavy mesh plate: part 1 of the asynchronous detection carries the triangle half as well
    (triangle prep, (bin, triangle) incidences, their sort, the sphere-triangle sweep) -- same statement, same comparison"""
    from tests.test_mesh import mesh_bed
    K, D, settle, steps = 20, 6, 1500, 803
    b = mesh_bed(pkg, 1500, cd_freq=K)
    b.SetExpandSafetyAdder(1.0)
    p, sc = b.Initialize()
    lock, asyn = pkg.Context(0), pkg.Context(0)
    for c in (lock, asyn):
        c.set_arith_mode("exact")
        c.set_params(p), c.upload_scene(sc)
    one = orc.make_sim(pkg, p, sc)
    lock.step(settle), asyn.step(settle)
    asyn.set_async_detection(D)
    asyn.set_timing(1)
    lock.step(steps), asyn.step(steps), one.step(settle + steps)
    assert asyn.kernel_time_ms("detect_async_part1")[1] >= steps // K - 1  # (the asynchronous path really served)
    s_lock, s_asyn, s_one = lock.download_state(), asyn.download_state(), one.download_state()
    for k in KEYS:
        assert np.array_equal(s_asyn[k], s_lock[k]), k
        assert np.array_equal(s_asyn[k], s_one[k]), k
    ta = asyn.contacts()[2]
    assert (ta == 2).sum() > 20, "sphere-triangle contacts in the asynchronous list"  # (DEME_SPHERE_MESH_CONTACT)


def test_async_detection_in_a_pair_of_slabs(pkg):
    """two x-slabs stepped through the library's halo loop (deme_halo_group_step): every slab takes its owner snapshot inside a
    step, once that step's ghosts are in place, and runs part 1 beside the next D steps -- exchanges included.  Against the same
    slabs with the lock-step detection: identical states and histories (exact arithmetic), while the lists differ (the
    asynchronous ones hold more near-pairs)."""
    from tests.test_decomp import GKEYS, build_global
    K, D, steps = 20, 6, 407
    b = pkg.model.packed_bed(3000, seed=6, cd_freq=K, spacing_mult=2.5, init_vz=-0.4, aspect=(2.0, 1.0, 0.5))
    b.SetExpandSafetyAdder(1.0)
    p, sc = b.Initialize()
    x = np.concatenate([bb.xyz for bb in b.batches])[:, 0]
    parts = pkg.decomp.decompose(b.arrays, b.counts, x, 2, halo=0.035)

    def run(lead):
        ctxs = []
        for pt in parts:
            c = pkg.Context(0)
            c.set_arith_mode("exact")
            c.set_params(p), c.upload_scene(pt["scene"])
            c.set_timing(1)
            if lead:
                c.set_async_detection(lead)
            ctxs.append(c)
        g = pkg.abi.HaloGroup(rank=0, world=1, device=0)
        for i, (c, pt) in enumerate(zip(ctxs, parts)):
            g.attach(c, pt, left=ctxs[i - 1] if i > 0 else None, right=ctxs[i + 1] if i + 1 < len(ctxs) else None)
        g.step(steps)
        g.sync()
        return g, ctxs

    g0, lock = run(0)
    g1, asyn = run(D)
    for a, c in zip(lock, asyn):
        assert c.kernel_time_ms("detect_async_part1")[1] >= steps // K - 1 and a.kernel_time_ms("detect_async_part1")[1] == 0
        sa, sc_ = a.download_state(), c.download_state()
        for k in GKEYS:
            assert np.array_equal(sa[k], sc_[k]), k
        wl = {q: w for q, w in zip(zip(*[x_.tolist() for x_ in a.contacts()[:3]]), a.wildcard(3).tolist())}
        wa = {q: w for q, w in zip(zip(*[x_.tolist() for x_ in c.contacts()[:3]]), c.wildcard(3).tolist())}
        common = [q for q in wl if q in wa]
        assert len(common) > 500 and all(wl[q] == wa[q] for q in common)
        assert all(wl[q] == 0.0 for q in wl if q not in wa) and all(wa[q] == 0.0 for q in wa if q not in wl)
    g0.close(), g1.close()
