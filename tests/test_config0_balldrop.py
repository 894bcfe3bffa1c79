"""BASELINE configs[0]: a BallDrop-like scene -- ~1e4 single-sphere clumps from 11 templates
(r = 1.25 ... 1.75 mm, rho = 2500), E 7e7, nu 0.24, CoR 0.9, mu 0.3, Crr 0, box 0.2 x 0.2 x 2 "top_open", h 2e-6, plus a
spherical projectile mesh dropped into the bed (DEMdemo_BallDrop.cpp:53-150; seeded HCP + jitter instead of
std::random_device / PD sampling).  The CPU test is the plumbing run on the oracle; the GPU test is parity."""
import math
import os

import numpy as np
import pytest


def icosphere(radius, subdiv=2):
    t = (1.0 + 5.0 ** 0.5) / 2.0
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t), (t, 0, -1), (t, 0, 1),
         (-t, 0, -1), (-t, 0, 1)]
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6), (7, 1, 8),
         (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    v = [np.array(x, float) / np.linalg.norm(x) for x in v]
    for _ in range(subdiv):
        cache, nf = {}, []

        def mid(a, b):
            key = (min(a, b), max(a, b))
            if key not in cache:
                m = v[a] + v[b]
                v.append(m / np.linalg.norm(m))
                cache[key] = len(v) - 1
            return cache[key]
        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        f = nf
    return (np.array(v) * radius).astype(np.float32), np.array(f, np.int64)


def balldrop(pkg, n_target=10000):
    b = pkg.model.SceneBuilder()
    mat = b.LoadMaterial({"E": 7e7, "nu": 0.24, "CoR": 0.9, "mu": 0.3, "Crr": 0.0})
    b.InstructBoxDomainDimension((-0.1, 0.1), (-0.1, 0.1), (0.0, 2.0))
    b.InstructBoxDomainBoundingBC("top_open", mat)
    radii = [0.00125 + 0.00005 * i for i in range(11)]
    tmpls = [b.LoadSphereType(2500.0 * 4.0 / 3.0 * math.pi * r ** 3, r, mat) for r in radii]
    sep = 0.0033  # denser than the largest diameter (3.5 mm): the bed starts with contacts, like a poured bed
    side = int(round((n_target / 6) ** 0.5))
    pts = pkg.model.hcp_points([-side * sep / 2, -side * sep / 2, 0.0017], [side * sep / 2, side * sep / 2, 0.0017 + 7 * sep], sep)
    pts = pts[:n_target]  # z-major ordering: the lowest layers
    rng = np.random.default_rng(12345)
    pts = pts + ((rng.random(pts.shape) * 2 - 1) * 0.02 * sep).astype(np.float32)
    pick = rng.integers(0, len(tmpls), len(pts))
    batch = b.AddClumps([tmpls[i] for i in pick], pts)
    batch.SetVel(np.tile(np.array([0, 0, -0.2], np.float32), (len(pts), 1)))
    # the reference's projectile: data/mesh/sphere.obj (a unit icosphere, 162 vertices / 320 facets) scaled to 12 mm
    v, f = pkg.io.read_obj(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_data", "sphere.obj"))
    proj = b.AddMeshObject(v * np.float32(0.012), f, mat)
    under = (np.abs(pts[:, 0]) < 0.008) & (np.abs(pts[:, 1]) < 0.008)  # the top layer may be partial: look under the projectile
    proj.SetInitPos((0.0, 0.0, float(pts[under, 2].max()) + 0.0015 + 0.012 + 0.0001))  # ~0.1 mm above the spheres below it
    proj.SetMass(2.6e3 * 4.0 / 3.0 * math.pi * 0.012 ** 3)
    proj.SetMOI((2.7e-7, 2.7e-7, 2.7e-7))
    proj.SetFamily(2)
    proj.vel = (0.0, 0.0, -2.0)
    b.SetInitTimeStep(2e-6)
    b.SetGravitationalAcceleration((0, 0, -9.81))
    b.SetCDUpdateFreq(0)
    b.SetInitBinSizeAsMultipleOfSmallestSphere(4.0)
    b.SetMaxVelocity(15.0)
    b.SetErrorOutVelocity(1e3)
    return b


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
