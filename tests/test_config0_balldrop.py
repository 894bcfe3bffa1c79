"""BASELINE configs[0]: a BallDrop-like scene -- ~1e4 single-sphere clumps from 11 templates
(r = 1.25 ... 1.75 mm, rho = 2500), E 7e7, nu 0.24, CoR 0.9, mu 0.3, Crr 0, box 0.2 x 0.2 x 2 "top_open", h 2e-6, plus a
spherical projectile mesh dropped into the bed (DEMdemo_BallDrop.cpp:53-150; seeded HCP + jitter instead of
std::random_device / PD sampling).  The CPU test is the plumbing run on the oracle; the GPU test is parity."""
import os

import numpy as np
import pytest


def balldrop(pkg, n_target=10000):
    # the reference's projectile: data/mesh/sphere.obj (a unit icosphere, 162 vertices / 320 facets) scaled to 12 mm
    v, f = pkg.io.read_obj(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_data", "sphere.obj"))
    return pkg.model.balldrop_like(n_target, seed=12345, projectile=(v, f))


def test_config0_runs_on_the_oracle(pkg, orc):
    b = balldrop(pkg)
    p, sc = b.Initialize()
    assert 8000 < int(sc.nOwnerClumps) < 13000 and int(sc.nTri) == 320 and len(b.templates) == 11
    sim = orc.make_sim(pkg, p, sc)
    sim.step(150)
    a, bb, t, _ = sim.contacts()
    assert (t == 1).sum() > 3000 and (t == 2).sum() > 3 and (t >= 11).sum() > 300  # sphere-sphere, projectile, floor
    st = sim.download_state()
    n = int(sc.nOwnerClumps)
    assert np.isfinite(st["vZ"]).all() and st["vZ"][n + 1] < -1.0  # the projectile (last owner) is still coming down
    assert abs(sim.inspect("clump_mass") - b.arrays["MassProperties"][b.arrays["inertiaPropOffsets"][:n]].sum()) < 1e-6


@pytest.mark.gpu
def test_config0_gpu_matches_oracle(pkg, orc):
    b = balldrop(pkg)
    p, sc = b.Initialize()
    ctx = pkg.Context(0)
    ctx.set_params(p), ctx.upload_scene(sc)
    sim = orc.make_sim(pkg, p, sc)
    ctx.step(150), sim.step(150)
    ga, oa = ctx.contacts(), sim.contacts()
    assert len(ga[0]) == len(oa[0]) and all(np.array_equal(x, y) for x, y in zip(ga[:3], oa[:3]))
    gs, os_ = ctx.download_state(), sim.download_state()
    X = pkg.model.decode_positions(gs["voxelID"], gs["locX"], gs["locY"], gs["locZ"], p.nvXp2, p.nvYp2, p.voxelSize, p.l)
    Y = pkg.model.decode_positions(os_["voxelID"], os_["locX"], os_["locY"], os_["locZ"], p.nvXp2, p.nvYp2, p.voxelSize, p.l)
    assert np.abs(X - Y).max() == 0.0  # bit-identical trajectories
    for q in ("clump_max_z", "clump_mass", "max_absv"):
        assert abs(ctx.inspect(q) - sim.inspect(q)) <= 2e-5 * abs(sim.inspect(q))


def test_reference_projectile_mesh_file(pkg):
    """tests/golden/ref_data/sphere.obj is the reference's data/mesh/sphere.obj (a DATA file): `f a//n b//n c//n` corners; the
    a twice-subdivided icosahedron on the unit sphere, outward-facing facets"""
    v, f = pkg.io.read_obj(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_data", "sphere.obj"))
    assert v.shape == (162, 3) and f.shape == (320, 3)
    assert np.allclose(np.linalg.norm(v, axis=1), 1.0, atol=2e-6)
    # a closed 2-manifold: every edge belongs to exactly two facets, V - E + F = 2; 12 vertices of valence 5, the rest 6
    e = np.sort(np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]]), axis=1)
    ue, cnt = np.unique(e, axis=0, return_counts=True)
    assert (cnt == 2).all() and len(v) - len(ue) + len(f) == 2
    val = np.bincount(ue.reshape(-1), minlength=len(v))
    assert sorted(np.unique(val).tolist()) == [5, 6] and int((val == 5).sum()) == 12
    n = np.cross(v[f[:, 1]] - v[f[:, 0]], v[f[:, 2]] - v[f[:, 0]])
    c = v[f].mean(axis=1)
    assert (np.einsum("ij,ij->i", n, c) > 0).all()
