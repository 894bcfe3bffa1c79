"""CPU tests: the oracle restatement against the golden vectors generated from the
reference's own helpers/fragments (tests/golden/make_golden.py, oracle/_ref).
Integer / decision outputs must be bit-exact; fp outputs are compared bit-exactly
too because both sides ran on the same host libm with contraction off."""
import numpy as np
import pytest


def same(a, b):
    a = np.asarray(a)
    b = np.asarray(b)
    assert a.shape == b.shape
    if a.dtype.kind == "f":
        ok = (a == b) | (np.isnan(a) & np.isnan(b))
        assert ok.all(), f"{int((~ok).sum())} of {a.size} differ; max abs {np.nanmax(np.abs(a - b))}"
    else:
        assert (a == b).all()


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_g1_codec(orc, golden, tag):
    nvx, nvy, voxel, l = golden[f"g1{tag}_cfg"]
    nvx, nvy = int(nvx), int(nvy)
    vid, sx, sy, sz = orc.encode("orc", golden[f"g1{tag}_X"], golden[f"g1{tag}_Y"], golden[f"g1{tag}_Z"], nvx, nvy,
                                 voxel, l)
    same(vid, golden[f"g1{tag}_vid"]), same(sx, golden[f"g1{tag}_sx"]), same(sy, golden[f"g1{tag}_sy"])
    same(sz, golden[f"g1{tag}_sz"])
    dX, dY, dZ = orc.decode("orc", vid, sx, sy, sz, nvx, nvy, voxel, l)
    same(dX, golden[f"g1{tag}_dX"]), same(dY, golden[f"g1{tag}_dY"]), same(dZ, golden[f"g1{tag}_dZ"])
    # round-trip property: decode(encode(x)) <= x and within one sub-voxel
    assert (dX <= golden[f"g1{tag}_X"]).all() and (golden[f"g1{tag}_X"] - dX < 1.0001 * l).all()


def test_g2_spheres_overlap_and_bin(orc, golden):
    t, CP, nrm, d = orc.spheres_overlap("orc", golden["g2_A"], golden["g2_rA"], golden["g2_B"], golden["g2_rB"])
    same(t, golden["g2_type"]), same(CP, golden["g2_CP"]), same(nrm, golden["g2_nrm"]), same(d, golden["g2_depth"])
    assert 0 < int(t.sum()) < len(t)  # both outcomes exercised
    bs, nbx, nby = golden["g2_bincfg"]
    b = orc.point_bin("orc", np.ascontiguousarray(CP[:, 0]), np.ascontiguousarray(CP[:, 1]),
                      np.ascontiguousarray(CP[:, 2]), float(bs), int(nbx), int(nby))
    same(b, golden["g2_bin"])


def test_mask_pair(orc, golden):
    same(orc.mask_pair("orc", golden["gm_i"], golden["gm_j"]), golden["gm_out"])
    # symmetric and within the table
    assert golden["gm_out"].max() < 32896


def test_rotation_and_hamilton(orc, golden):
    q = golden["gq_q"]
    same(orc.rotate("orc", golden["gq_v"], q), golden["gq_rot"])
    same(orc.rotate_d("orc", golden["gq_vd"], q), golden["gq_rotd"])
    same(orc.hamilton("orc", q, golden["gh_q2"]), golden["gh_out"])


def test_g5_force_models(orc, golden):
    args = (golden["g5_depth"], golden["g5_fin"], golden["g5_mu"], golden["g5_crr"], golden["g5_hist"])
    h, o = orc.force("orc", 0, *args)
    same(o, golden["g5_out_full"]), same(h, golden["g5_hist_full"])
    _, o2 = orc.force("orc", 1, *args)
    same(o2, golden["g5_out_frictionless"])
    E, G = orc.mat_proxy("orc", golden["g5_fin"][:, 34], golden["g5_fin"][:, 35], golden["g5_fin"][:, 36],
                         golden["g5_fin"][:, 37])
    same(E, golden["g5_Eeff"]), same(G, golden["g5_Geff"])
    # non-contacts clear the history
    neg = golden["g5_depth"] <= 0
    assert neg.any() and (h[neg] == 0).all() and (o[neg] == 0).all()


def test_g8_sphere_entity(orc, golden):
    t, CP, nrm, d = orc.sphere_entity("orc", golden["g8_A"], golden["g8_radA"], golden["g8_typeB"], golden["g8_B"],
                                      golden["g8_dirB"], golden["g8_size1"], golden["g8_nsign"], golden["g8_beta"])
    same(t, golden["g8_type"]), same(d, golden["g8_depth"]), same(CP, golden["g8_CP"]), same(nrm, golden["g8_nrm"])
    assert set(np.unique(t)) >= {0, 11, 13}


def test_reference_anchor_values(orc):
    """Outputs of the reference itself recorded in SURVEY App. E (probe run)."""
    vid, sx, sy, sz = orc.encode("orc", np.array([1.2345]), np.array([2.5]), np.array([0.75]), 21, 21, 1e-3,
                                 1e-3 / 65536)
    assert int(vid[0]) == 3298540126209234 and (int(sx[0]), int(sy[0]), int(sz[0])) == (32767, 0, 0)
    X, _, _ = orc.decode("orc", vid, sx, sy, sz, 21, 21, 1e-3, 1e-3 / 65536)
    assert abs(X[0] - 1.234499985) < 1e-9
    t, CP, nrm, d = orc.spheres_overlap("orc", np.zeros((1, 3)), np.ones(1), np.array([[1.5, 0, 0]]), np.ones(1))
    assert t[0] == 1 and d[0] == 0.5 and CP[0, 0] == 0.75 and nrm[0, 0] == -1.0
    fin = np.zeros((1, 39), np.float32)
    fin[0, 0:3] = (0, 0, 1)
    fin[0, 3:5] = 1e-3
    fin[0, 5:7] = 5e-3
    fin[0, 7] = fin[0, 11] = 1.0
    fin[0, 15:18] = (0, 0, -5e-3)
    fin[0, 18:21] = (0, 0, 5e-3)
    fin[0, 23], fin[0, 26] = -0.05, 0.05
    fin[0, 33] = 5e-6
    fin[0, 34], fin[0, 35], fin[0, 36], fin[0, 37], fin[0, 38] = 1e8, 0.3, 1e8, 0.3, 0.6
    _, o = orc.force("orc", 0, np.array([1e-4]), fin, np.array([0.2], np.float32), np.zeros(1, np.float32),
                     np.zeros((1, 4), np.float32))
    assert abs(o[0, 2] - 3.8166) < 1e-3


def test_ref_build_agrees_when_present(orc, golden):
    """Where oracle/_ref exists (survey container and the prebuilt file on the GPU box),
    re-run the reference build on fresh seeded inputs against the restatement."""
    if not orc.ref_available():
        pytest.skip("oracle/_ref/libdeme_ref.so not present")
    rng = np.random.default_rng(99)
    n = 5000
    A = rng.random((n, 3))
    rA = rng.uniform(1e-3, 1e-2, n)
    rB = rng.uniform(1e-3, 1e-2, n)
    B = A + (rng.standard_normal((n, 3)) * 6e-3)
    for x, y in zip(orc.spheres_overlap("orc", A, rA, B, rB), orc.spheres_overlap("ref", A, rA, B, rB)):
        same(x, y)
    q = rng.standard_normal((n, 4)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    v = rng.standard_normal((n, 3)).astype(np.float32)
    same(orc.rotate("orc", v, q), orc.rotate("ref", v, q))


def test_g7_triangle_box_overlap(orc, golden):
    """check_TriangleBoxOverlap (DEMTriangleBoxIntersect.cu:295), bit-exact verdicts incl. face-plane grazing cases."""
    hit = orc.tri_box(golden["g7_center"], golden["g7_half"], golden["g7_A"], golden["g7_B"], golden["g7_C"])
    assert 0.15 < golden["g7_hit"].mean() < 0.85
    assert (hit == golden["g7_hit"]).all()


def test_g7b_triangle_bbox_bins(orc, golden):
    """boundingBoxIntersectBin (DEMHelperKernels.cuh:528-565): integer bin ranges, bit-exact."""
    lo, hi = orc.tri_bbox(golden["g7b_A"], golden["g7b_B"], golden["g7b_C"], float(golden["g7b_binSize"]),
                          [int(x) for x in golden["g7b_nb"]])
    assert (lo == golden["g7b_L"]).all() and (hi == golden["g7b_U"]).all()
    assert (hi > lo).any()


def test_g9_integrator_velocity_pass_on(orc, golden):
    """IntegrationVelPassOn{ForwardEuler,CenteredDiff,ExtendedTaylor}.cu, compiled from the reference when the fixture was
    made: the velocity the position update uses, bit for bit"""
    import ctypes as C
    L = orc.lib()
    ov, vu = np.ascontiguousarray(golden["g9_old_v"]), np.ascontiguousarray(golden["g9_v_update"])
    for scheme in (0, 1, 2):
        out = np.zeros_like(ov)
        L.orc_el_vel_pass_on(C.c_size_t(len(ov)), C.c_int(scheme), C.c_void_p(ov.ctypes.data), C.c_void_p(vu.ctypes.data),
                             C.c_void_p(out.ctypes.data))
        assert np.array_equal(out, golden[f"g9_v_scheme{scheme}"]), scheme
    assert not np.array_equal(golden["g9_v_scheme1"], golden["g9_v_scheme2"])


def test_g3_pair_enumeration_covers_what_the_cyclic_pairing_covers(golden):
    """recoverCntPair (DEMHelperKernels.cuh) enumerates the n(n-1)/2 pairs of a bin; k_sweep spreads the same pairs by cyclic
    pairing instead (entry k tests k+1 ... k+floor((n-1)/2) mod n, plus k+n/2 for k < n/2 when n is even): same set, each once"""
    ind, n, gi, gj = (golden[k] for k in ("g3_ind", "g3_n", "g3_i", "g3_j"))
    for nn in np.unique(n)[:: max(1, len(np.unique(n)) // 40)]:
        sel = n == nn
        ref = set(zip(np.minimum(gi[sel], gj[sel]).tolist(), np.maximum(gi[sel], gj[sel]).tolist()))
        nn = int(nn)
        half = (nn - 1) // 2
        mine = []
        for k in range(nn):
            for m in range(1, half + 1):
                mine.append((k, (k + m) % nn))
            if nn % 2 == 0 and k < nn // 2:
                mine.append((k, k + nn // 2))
        canon = [(min(a, b), max(a, b)) for a, b in mine]
        assert len(canon) == nn * (nn - 1) // 2 == len(set(canon))
        full = {(a, b) for a in range(nn) for b in range(a + 1, nn)}
        assert set(canon) == full and ref <= full
