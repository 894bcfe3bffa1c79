"""Adaptive controllers (DEM/kT.cpp:43-98, DEM/dT.cpp:2276-2299): the bin-size hill climb and the
update-frequency tuner run on device timers and must not change the physics -- the contact set does not depend on the bin
size, and a larger K only adds non-touching pairs to the list."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

STATE_KEYS = ("voxelID", "locX", "locY", "locZ", "oriQw", "oriQx", "oriQy", "oriQz", "vX", "vY", "vZ", "omgBarX", "omgBarY",
              "omgBarZ")


def _ctx(pkg, b):
    p, sc = b.Initialize()
    ctx = pkg.Context(0)
    ctx.set_params(p), ctx.upload_scene(sc)
    return ctx, p, sc


def test_adaptive_bin_size_keeps_the_trajectory(pkg, orc):
    """detection every step, bin size adjusted every 5 detections: lists and state stay bit-identical to the oracle, which
    keeps the initial bin size throughout"""
    b = pkg.model.packed_bed(1500, seed=4, cd_freq=0, spacing_mult=2.6, init_vz=-0.5)
    ctx, p, sc = _ctx(pkg, b)
    sim = orc.make_sim(pkg, p, sc)
    ctx.set_adaptive(bin_size=True, bin_observe=5, bin_max_rate=0.2, bin_acc=0.5)
    b0 = ctx.adaptive_state()[0]
    sizes = set()
    for chunk in range(6):
        ctx.step(25), sim.step(25)
        sizes.add(ctx.adaptive_state()[0])
        a, bb, t, _ = ctx.contacts()
        oa, ob, ot, _ = sim.contacts()
        assert np.array_equal(a, oa) and np.array_equal(bb, ob) and np.array_equal(t, ot), chunk
        gs, os_ = ctx.download_state(), sim.download_state()
        for k in STATE_KEYS:
            assert np.array_equal(gs[k], os_[k]), (chunk, k)
    size, K, n_bin, n_freq = ctx.adaptive_state()
    assert n_bin >= 20 and n_freq == 0 and K == 0
    assert len(sizes) >= 2
    assert 1e-3 * b0 < size < 1e3 * b0  # (a detection of 1500 clumps takes microseconds: the climb is a timing-noise walk here)
    assert ctx.counts().nContacts > 100


def test_adaptive_update_frequency_keeps_the_trajectory(pkg, orc):
    """K tuned on the measured time per step: K moves, the state stays bit-identical to an oracle run at fixed K (a larger K
    only lists more non-touching pairs)"""
    b = pkg.model.packed_bed(1500, seed=5, cd_freq=4, spacing_mult=2.6, init_vz=-0.5)
    b.SetExpandSafetyAdder(3.0)  # margin for 3 m/s on top of each owner's own speed: no pair can be missed at any K <= 12
    ctx, p, sc = _ctx(pkg, b)
    sim = orc.make_sim(pkg, p, sc)
    ctx.set_adaptive(update_freq=True, max_update_freq=12, freq_observe=2)
    Ks = set()
    for chunk in range(8):
        ctx.step(40), sim.step(40)
        Ks.add(ctx.adaptive_state()[1])
        gs, os_ = ctx.download_state(), sim.download_state()
        for k in STATE_KEYS:
            assert np.array_equal(gs[k], os_[k]), (chunk, k)
    size, K, n_bin, n_freq = ctx.adaptive_state()
    assert n_freq >= 5 and n_bin == 0 and 1 <= K <= 12 and Ks != {4}
    assert ctx.counts().nDetections != sim.counts().nDetections  # it really ran on a different schedule


def test_adaptive_safety_override_shrinks_overfull_bins(pkg):
    """bins holding more than binUpperSafety * errOutBinSphNum spheres force the size down whatever the timing says"""
    b = pkg.model.packed_bed(1500, seed=6, cd_freq=0, spacing_mult=2.4, init_vz=-0.3)
    b.SetInitBinSizeAsMultipleOfSmallestSphere(10.0)
    ctx, p, sc = _ctx(pkg, b)
    ctx.step(1)
    full = int(ctx.counts().maxSpheresInBin)
    p.errOutBinSphNum = 2 * full  # allowance just above what the bins hold; safety threshold (0.25 x) far below
    ctx.set_params(p)
    ctx.set_adaptive(bin_size=True, bin_observe=2, bin_max_rate=0.1, bin_acc=1.0)
    b0 = ctx.adaptive_state()[0]
    ctx.step(40)
    size, _, n_bin, _ = ctx.adaptive_state()
    # (once the bins are below the threshold the climb is free again and may wander back up to it -- never beyond: the
    # override re-engages at ~0.8 x the initial size)
    assert n_bin >= 10 and size < 0.95 * b0
    assert int(ctx.counts().maxSpheresInBin) < full
