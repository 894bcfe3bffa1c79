"""CPU tests: the C-ABI library loads and exports every symbol include/deme_hip.h declares (no compute
calls without a GPU), ctypes layouts match the header, and the host-side model builder reproduces the
reference's sizing rules."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(pkg):
    lib = pkg.abi.load_library()
    names = pkg.abi.exported_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/deme_hip.h but not exported"
    assert b"gfx950" in lib.deme_version()


def test_no_device_is_a_loud_error(pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(pkg.abi.DemeError):
        pkg.Context(0)


def test_struct_layouts_match_header(pkg):
    # sizes computed by a C compiler from the real header
    import subprocess
    import tempfile
    src = ('#include <stdio.h>\n#include "deme_hip.h"\nint main(){printf("%zu %zu %zu %zu %zu\\n",sizeof(DemeParams),'
           'sizeof(DemeScene),sizeof(DemeOwnerState),sizeof(DemeCounts),sizeof(DemeAdaptive));}')
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "s.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "s.c"), "-o",
                               os.path.join(d, "s")])
        out = subprocess.check_output([os.path.join(d, "s")]).split()
    got = [C.sizeof(pkg.DemeParams), C.sizeof(pkg.DemeScene), C.sizeof(pkg.DemeOwnerState), C.sizeof(pkg.DemeCounts),
           C.sizeof(pkg.abi.DemeAdaptive)]
    assert [int(x) for x in out] == got


def test_product_never_touches_the_oracle(pkg):
    for dp, _, files in os.walk(os.path.join(ROOT, "dem-engine_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".hpp", "Makefile")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "libdeme_oracle" not in txt and "/root/reference" not in txt, f
                assert not re.search(r"^\s*(from|import)\s+oracle", txt, re.M), f
                assert not re.search(r"#include\s*[\"<][^\">]*oracle", txt), f


def test_voxel_bit_split_and_bins(pkg):
    b = pkg.SceneBuilder()
    m = b.LoadMaterial({"E": 1e8, "nu": 0.3, "CoR": 0.5})
    b.InstructBoxDomainDimension(0.2, 0.2, 2.0)  # BallDrop box (DEMdemo_BallDrop.cpp)
    b.InstructBoxDomainBoundingBC("top_open", m)
    t = b.LoadSphereType(1e-3, 0.00125, m)
    b.AddClumps(t, [[0, 0, 0]])
    p, sc = b.Initialize()
    assert p.nvXp2 + p.nvYp2 + p.nvZp2 == 64
    assert (p.nvXp2, p.nvYp2, p.nvZp2) == (20, 21, 23)  # z is 10x longer: 3 more bits; the left-over bit goes to y
    assert p.voxelSize == 65536 * p.l
    # the voxel grid covers the (20% enlarged) target box in every direction
    for n, size in ((p.nvXp2, 0.24), (p.nvYp2, 0.24), (p.nvZp2, 2.4)):
        assert p.voxelSize * 2 ** n >= size * 0.999999
    # default initial bin size: the reference's target of ~1e6 bins (API.h:1403-1412, loop of APIPrivate.cpp:525-541 from 8 radii)
    assert 0.67e6 <= p.nbX * p.nbY * p.nbZ <= 1.5e6
    assert p.nbX == int(p.voxelSize * 2 ** p.nvXp2 / p.binSize) + 1
    b.SetInitBinSizeAsMultipleOfSmallestSphere(8)
    p8, _ = b.Initialize()
    assert p8.binSize == float(np.float32(8) * np.float32(0.00125))  # a float product stored in a double, like the reference's
    assert sc.nAnal == 5 and sc.nOwners == 2
    # wall planes sit on the USER box faces, normals inward
    a = b.arrays
    assert a["objRelPosZ"][0] == np.float32(-1.0) and a["objRotZ"][0] == 1.0
    assert a["familyID"][1] == 255 and a["familyFlags"][255] == 1


def test_position_encoding_matches_oracle(pkg, orc):
    b = pkg.model.packed_bed(500, seed=3)
    p, sc = b.Initialize()
    xyz = np.concatenate([bb.xyz for bb in b.batches]).astype(np.float32)
    lbf = np.array([p.LBFX, p.LBFY, p.LBFZ], np.float32)
    sh = (xyz - lbf).astype(np.float32).astype(np.float64)
    vid, sx, sy, sz = orc.encode("orc", np.ascontiguousarray(sh[:, 0]), np.ascontiguousarray(sh[:, 1]),
                                 np.ascontiguousarray(sh[:, 2]), p.nvXp2, p.nvYp2, p.voxelSize, p.l)
    n = len(xyz)
    assert (b.arrays["voxelID"][:n] == vid).all() and (b.arrays["locX"][:n] == sx).all()
    assert (b.arrays["locY"][:n] == sy).all() and (b.arrays["locZ"][:n] == sz).all()


def test_three_sphere_template_and_pair_matrix(pkg):
    b = pkg.SceneBuilder()
    m0 = b.LoadMaterial({"E": 1e8, "nu": 0.3, "CoR": 0.6, "mu": 0.2, "Crr": 0.0})
    m1 = b.LoadMaterial({"E": 1e9, "nu": 0.3, "CoR": 0.8, "mu": 0.4, "Crr": 0.1})
    b.SetMaterialPropertyPair("mu", m0, m1, 0.5)
    t = b.LoadThreeSphereClump(0.005, 2.6e3, m0)
    b.AddClumps(t, [[0, 0, 0]])
    p, sc = b.Initialize()
    a = b.arrays
    assert np.allclose(a["Radii"], 0.004) and sc.nSpheres == 3 and sc.nComp == 3
    assert a["MassProperties"][0] == pytest.approx(2.6e3 * 5.5886717 * 0.005 ** 3, rel=1e-6)
    assert a["moiX"][0] == pytest.approx(2.928 * 2.6e3 * 0.005 ** 5, rel=1e-6)
    CoR = a["CoR"].reshape(2, 2)
    mu = a["mu"].reshape(2, 2)
    assert CoR[0, 1] == pytest.approx(0.7) and CoR[0, 0] == np.float32(0.6)  # off-diagonal default = mean
    assert mu[0, 1] == mu[1, 0] == np.float32(0.5)  # explicit pair override
    assert p.nContactWildcards == 4


def test_hcp_sampler_spacing(pkg):
    pts = pkg.model.hcp_points([0, 0, 0], [0.1, 0.1, 0.1], 0.015)
    d = np.linalg.norm(pts[None, :60] - pts[:60, None], axis=2)
    d[d == 0] = 1
    assert d.min() == pytest.approx(0.015, rel=1e-4)


def test_spatial_sort_keeps_the_scene_and_localises_neighbours(pkg):
    b = pkg.model.packed_bed(4000, seed=3, order="random")
    xyz0 = b.batches[0].xyz.copy()
    q0 = b.batches[0].oriq.copy()
    b.SortClumpsSpatially()
    xyz1 = b.batches[0].xyz
    # the same clumps (positions and orientations travel together), in a different order
    i0, i1 = np.lexsort(xyz0.T), np.lexsort(xyz1.T)
    assert np.array_equal(xyz0[i0], xyz1[i1]) and np.array_equal(q0[i0], b.batches[0].oriq[i1])
    # consecutive clumps are now close in space (random order: ~1/3 of the box apart)
    d0 = np.linalg.norm(np.diff(xyz0, axis=0), axis=1).mean()
    d1 = np.linalg.norm(np.diff(xyz1, axis=0), axis=1).mean()
    assert d1 < 0.25 * d0


@pytest.mark.gpu
def test_resort_clumps_in_a_running_simulation(pkg):
    """ResortClumps: a bed handed over in random order is renumbered along a Z-order curve mid-run; state, contact history and
    persistent marks follow, the physics carries on as in a twin that keeps its numbering (to fp32 summation order), and
    neighbours in space become neighbours in memory"""
    def make():
        b = pkg.model.packed_bed(4000, seed=3, cd_freq=5, spacing_mult=2.5, init_vz=-0.4, order="random")
        b.SetExpandSafetyAdder(1.0)
        p, sc = b.Initialize()
        ctx = pkg.Context(0)
        ctx.set_params(p), ctx.upload_scene(sc)
        return b, p, sc, ctx

    (b, p, sc, ctx), (b2, p2, sc2, twin) = make(), make()
    ctx.step(400), twin.step(400)
    ctx.mark_persistent_contacts(), twin.mark_persistent_contacts()
    n_marked = ctx.num_persistent_contacts()
    own_old = np.asarray(b.arrays["ownerClumpBody"], np.int64).copy()

    def spread(builder, c):
        a, bb, t, _ = c.contacts()
        own = np.asarray(builder.arrays["ownerClumpBody"], np.int64)
        ss = t == 1
        return float(np.abs(own[a[ss]] - own[bb[ss]]).mean()), int(ss.sum())

    before, n_ss = spread(b, ctx)
    pnew, scnew, new_of_old = b.ResortClumps(ctx, 400 * p.h)
    assert sorted(new_of_old[:int(sc.nOwnerClumps)].tolist()) == list(range(int(sc.nOwnerClumps)))
    assert ctx.num_persistent_contacts() == n_marked > 100
    ctx.step(25), twin.step(25)
    after, n_ss2 = spread(b, ctx)
    assert n_ss > 1000 and after < 0.2 * before, (before, after)  # owner-id distance of contacting clumps
    gs, ts = ctx.download_state(), twin.download_state()
    n = int(sc.nOwnerClumps)
    X = pkg.model.decode_positions(gs["voxelID"], gs["locX"], gs["locY"], gs["locZ"], p.nvXp2, p.nvYp2, p.voxelSize, p.l)
    Xt = pkg.model.decode_positions(ts["voxelID"], ts["locX"], ts["locY"], ts["locZ"], p.nvXp2, p.nvYp2, p.voxelSize, p.l)
    assert np.abs(X[new_of_old[:n]] - Xt[:n]).max() < 1e-9
    assert np.abs(gs["vZ"][new_of_old[:n]] - ts["vZ"][:n]).max() < 1e-5
    assert int(ctx.counts().nContacts) == int(twin.counts().nContacts)


def test_clump_outside_the_box_is_warned_about(pkg):
    """the reference's courtesy check at initialisation (dT.cpp:739-744, 887-893): a warning naming one such clump"""
    b = pkg.model.packed_bed(50, seed=1)
    b.batches[0].xyz[3] = (0.0, 0.0, 99.0)
    with pytest.warns(UserWarning, match="out of the box domain"):
        b.Initialize()
