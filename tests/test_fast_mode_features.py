"""The default (fast) arithmetic mode on the features the bit-exact suite covers in the exact mode: cylinders, a free owner with
hundreds of contacts (the separate heavy-owner reduction in the world frame), persistent contacts, family prescriptions, accelerations
added by the script, mixed clump templates, a metre-scale scene, a user force model.  Both modes run in the same library; the fast
one must stay within a tolerance tied to the scene's scale after N steps and keep the same contact list (same decision code)."""
import os

import numpy as np
import pytest

from tests.test_gpu_parity import persistent_scene, raft_scene
from tests.test_prescription import _scene as prescribed_scene

pytestmark = pytest.mark.gpu


def _pos(pkg, p, st):
    return pkg.model.decode_positions(st["voxelID"], st["locX"], st["locY"], st["locZ"], p.nvXp2, p.nvYp2, p.voxelSize, p.l)


def _pair(pkg, b, settle=0):
    """two contexts with the scene; both run `settle` steps in the exact mode (identical state and contact history), then the
    second one switches to the fast mode"""
    p, sc = b.Initialize()
    out = []
    for _ in range(2):
        c = pkg.Context(0)
        c.set_arith_mode("exact")
        c.set_params(p), c.upload_scene(sc)
        b.compile_into(c)
        if settle:
            c.step(settle)
        out.append(c)
    out[1].set_arith_mode("fast")
    return out[0], out[1], p, sc


def _compare(pkg, p, exact, fast, length, speed, same_list=True):
    """positions within 2e-4 of the scene's length scale (a sphere radius), velocities within 2e-2 of its speed scale: these
    scenes are violent on purpose (clumps thrown by a post they start inside of, an exploding lattice), so the bounds are those of a
    feature check -- a wrong formula shows as O(1) -- not the stated tolerance of tests/test_fast_mode.py"""
    se, sf = exact.download_state(), fast.download_state()
    dx = np.abs(_pos(pkg, p, se) - _pos(pkg, p, sf)).max()
    dv = max(np.abs(se[k] - sf[k]).max() for k in ("vX", "vY", "vZ"))
    assert np.isfinite(sf["vX"]).all() and dx < 2e-4 * length and dv < 2e-2 * speed, (dx, dv)
    if same_list:
        ce, cf = exact.contacts(), fast.contacts()
        assert all(np.array_equal(x, y) for x, y in zip(ce[:3], cf[:3]))
    return dx, dv


def test_fast_mode_cylinders(pkg):
    b = pkg.model.packed_bed(1500, seed=19, cd_freq=0, spacing_mult=3.0, init_vz=-0.3, aspect=(1.0, 1.0, 0.6))
    lo, hi = b.user_box_min, b.user_box_max
    cx, cy = float(lo[0] + hi[0]) / 2, float(lo[1] + hi[1]) / 2
    b.AddExternalObject().AddCylinder((cx, cy, 0.0), (0, 0, 1), 0.42 * float(hi[0] - lo[0]), 0, normal_inward=True)
    b.AddExternalObject().AddCylinder((cx, cy, 0.0), (0, 0, 1), 0.012, 0, normal_inward=False)
    e, f, p, sc = _pair(pkg, b, settle=4000)  # down to the floor and against the post
    e.step(100), f.step(100)
    _compare(pkg, p, e, f, 0.004, 0.5, same_list=False)
    assert (e.contacts()[2] == 13).sum() > 5


def test_fast_mode_free_owner_with_hundreds_of_contacts(pkg):
    b, n_raft = raft_scene(pkg)
    e, f, p, sc = _pair(pkg, b)
    e.step(100), f.step(100)
    _compare(pkg, p, e, f, 0.001, 0.02)
    raft = int(sc.nOwnerClumps) - 1
    se, sf = e.download_state(), f.download_state()
    assert se["vZ"][raft] != np.float32(-0.02) and abs(se["vZ"][raft] - sf["vZ"][raft]) < 2e-5  # the raft felt its ~280 contacts


def test_fast_mode_persistent_contacts_and_added_accelerations(pkg):
    b = persistent_scene(pkg)
    e, f, p, sc = _pair(pkg, b)
    rng = np.random.default_rng(3)
    acc = rng.uniform(-50, 50, (40, 3)).astype(np.float32)
    ang = rng.uniform(-500, 500, (40, 3)).astype(np.float32)
    for c in (e, f):
        c.step(40)
        c.mark_persistent_contacts(1, 1)  # either owner of family 1
        c.add_owner_acc(100, acc, ang)
        c.step(40)
    assert e.num_persistent_contacts() == f.num_persistent_contacts() > 0
    # an exploding start (spacing 2.6: overlaps at t = 0): compare loosely, the point is that the paths run and agree in kind
    _compare(pkg, p, e, f, 0.05, 50.0, same_list=False)


def test_fast_mode_prescribed_families(pkg):
    b = prescribed_scene(pkg)[0]
    e, f, p, sc = _pair(pkg, b)
    e.step(120), f.step(120)
    _compare(pkg, p, e, f, 0.004, 1.0, same_list=False)


def test_fast_mode_metre_scale(pkg):
    b = pkg.model.packed_bed(1200, seed=12, cd_freq=0, scale=0.5, spacing_mult=3.0, jitter=0.05, init_vz=-3.0, h=2e-4, E=1e7,
                             bin_multiple=5.0)
    e, f, p, sc = _pair(pkg, b, settle=8000)
    e.step(100), f.step(100)
    assert len(e.contacts()[0]) > 300
    _compare(pkg, p, e, f, 0.4, 3.0, same_list=False)


def test_fast_mode_mixed_templates_from_the_reference_data(pkg):
    ref = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_data")
    b = pkg.model.SceneBuilder()
    m = b.LoadMaterial({"E": 1e8, "nu": 0.3, "CoR": 0.5, "mu": 0.3, "Crr": 0.02})
    b.InstructBoxDomainDimension((0.0, 0.4), (0.0, 0.4), (0.0, 0.5))
    b.InstructBoxDomainBoundingBC("top_open", m)
    kinds = []
    for name, mass in (("3_clump.csv", 2.6e3 * 5.5886717), ("6_clump.csv", 2.6e3 * 8.0), ("ellipsoid_2_1_1.csv", 2.6e3 * 8.4)):
        d = np.loadtxt(os.path.join(ref, name), delimiter=",", skiprows=1)
        t = b.LoadClumpType(mass, (mass * 0.5, mass * 0.5, mass * 0.6), d[:, 3], d[:, :3], m)
        t.Scale(0.006)
        kinds.append(t)
    rng = np.random.default_rng(9)
    g = np.stack(np.meshgrid(np.arange(9), np.arange(9), np.arange(6), indexing="ij"), -1).reshape(-1, 3)
    xyz = (0.05 + 0.035 * g + rng.uniform(-2e-3, 2e-3, g.shape)).astype(np.float32)
    batch = b.AddClumps([kinds[i % 3] for i in range(len(xyz))], xyz)
    q = rng.standard_normal((len(xyz), 4))
    batch.SetOriQ((q / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float32))
    batch.SetVel(np.tile(np.array([0, 0, -1.5], np.float32), (len(xyz), 1)))
    b.SetInitTimeStep(5e-6), b.SetGravitationalAcceleration((0, 0, -9.81)), b.SetCDUpdateFreq(10)
    b.SetExpandSafetyAdder(0.5), b.SetInitBinSizeAsMultipleOfSmallestSphere(4.0)
    e, f, p, sc = _pair(pkg, b, settle=12000)
    e.step(100), f.step(100)
    assert len(e.contacts()[0]) > 50
    _compare(pkg, p, e, f, 0.006, 1.5, same_list=False)
