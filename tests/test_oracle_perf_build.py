"""The oracle's TIMED build (oracle/libdeme_oracle_perf.so: -O3, ORC_PERF -- incidence list, sorts, history map and per-owner
accumulation spread over the OpenMP team; what bench.py's cpu_baseline runs) against the parity build the rest of the suite
uses: same lists, same history map, states equal to fp32 contraction (the timed build lets the compiler fuse multiply-adds)."""
import numpy as np


def test_perf_build_agrees_with_the_parity_build(pkg, orc):
    b = pkg.model.packed_bed(1500, seed=3, cd_freq=7, spacing_mult=2.4, init_vz=-0.5)
    p, sc = b.Initialize()
    ref = orc.make_sim(pkg, p, sc)
    orc.set_variant(True)
    try:
        fast = orc.make_sim(pkg, p, sc)
    finally:
        orc.set_variant(False)
    assert fast.L is not ref.L
    ref.step(1), fast.step(1)
    for x, y in zip(ref.contacts(), fast.contacts()):  # first list: built from identical states
        assert np.array_equal(x, y)
    assert np.array_equal(ref.bin_incidence()[0], fast.bin_incidence()[0]) and np.array_equal(ref.bin_incidence()[1], fast.bin_incidence()[1])
    ref.step(40), fast.step(40)
    a, c = ref.contacts(), fast.contacts()
    assert len(a[0]) > 2000 and all(np.array_equal(x, y) for x, y in zip(a[:3], c[:3]))
    assert np.array_equal(a[3], c[3])  # the history map: sequential merge vs per-key binary search
    r, f = ref.download_state(), fast.download_state()
    X = pkg.model.decode_positions(r["voxelID"], r["locX"], r["locY"], r["locZ"], p.nvXp2, p.nvYp2, p.voxelSize, p.l)
    Y = pkg.model.decode_positions(f["voxelID"], f["locX"], f["locY"], f["locZ"], p.nvXp2, p.nvYp2, p.voxelSize, p.l)
    assert np.abs(X - Y).max() < 1e-8
    assert max(np.abs(r[k] - f[k]).max() for k in ("vX", "vY", "vZ")) < 1e-3
