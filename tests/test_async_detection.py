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
