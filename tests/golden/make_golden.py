"""Generate tests/golden/elementwise.npz from the REFERENCE's own helpers.

Runs only where /root/reference exists (this container): it drives
oracle/_ref/libdeme_ref.so -- the reference's __host__ __device__ functions and
force-model fragments compiled from where they lie (oracle/Makefile) -- on
seeded inputs and stores inputs + outputs.  The fixtures are data; no reference
text is stored.  G-numbers follow SURVEY section 8c.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import orc  # noqa: E402


def unit(v):
    return v / np.linalg.norm(v, axis=1, keepdims=True)


def force_inputs(rng, n):
    fin = np.zeros((n, orc.FORCE_NF), np.float32)
    fin[:, 0:3] = unit(rng.standard_normal((n, 3))).astype(np.float32)          # B2A
    fin[:, 3:5] = rng.uniform(1e-4, 5e-3, (n, 2)).astype(np.float32)            # masses
    fin[:, 5:7] = rng.uniform(1e-3, 8e-3, (n, 2)).astype(np.float32)            # radii
    big = rng.random(n) < 0.15                                                    # wall-like partner
    fin[big, 4] = 1e6
    fin[big, 6] = 1e15
    fin[:, 7:11] = unit(rng.standard_normal((n, 4))).astype(np.float32)         # AOriQ wxyz
    fin[:, 11:15] = unit(rng.standard_normal((n, 4))).astype(np.float32)        # BOriQ
    fin[:, 15:21] = rng.uniform(-6e-3, 6e-3, (n, 6)).astype(np.float32)         # locCPA, locCPB
    fin[:, 21:27] = rng.uniform(-1.0, 1.0, (n, 6)).astype(np.float32)           # lin vel
    fin[:, 27:33] = rng.uniform(-30.0, 30.0, (n, 6)).astype(np.float32)         # rot vel
    fin[:, 33] = np.float32(5e-6)                                                # ts
    fin[:, 34] = rng.choice([1e8, 7e7, 1e9], n).astype(np.float32)              # E_A
    fin[:, 35] = rng.uniform(0.2, 0.4, n).astype(np.float32)
    fin[:, 36] = rng.choice([1e8, 7e7, 1e9], n).astype(np.float32)
    fin[:, 37] = rng.uniform(0.2, 0.4, n).astype(np.float32)
    fin[:, 38] = rng.choice([0.0, 0.3, 0.6, 0.9], n).astype(np.float32)         # CoR (incl. the <tiny branch)
    depth = rng.uniform(-1e-5, 3e-4, n)                                          # some non-contacts
    mu = rng.choice([0.0, 0.2, 0.5], n).astype(np.float32)
    crr = rng.choice([0.0, 0.0, 0.05], n).astype(np.float32)
    hist = np.zeros((n, 4), np.float32)
    hist[:, 0:3] = rng.uniform(-2e-5, 2e-5, (n, 3)).astype(np.float32)
    hist[:, 3] = rng.uniform(0, 2e-3, n).astype(np.float32)
    hist[rng.random(n) < 0.3] = 0
    return depth, fin, mu, crr, hist


def main():
    if not orc.ref_available():
        orc.build(force=True)
    assert orc.ref_available(), "oracle/_ref/libdeme_ref.so missing (needs /root/reference)"
    rng = np.random.default_rng(20240927)
    out = {}

    # G1 codec round trips, three bit splits
    for tag, (nvx, nvy, voxel, l) in {"a": (21, 21, 1e-3, 1e-3 / 65536), "b": (22, 21, 1.28e-7, 1.28e-7 / 65536),
                                      "c": (16, 24, 3.3e-5, 3.3e-5 / 65536)}.items():
        n = 2000
        span = np.array([voxel * (2 ** nvx), voxel * (2 ** nvy), voxel * (2 ** (64 - nvx - nvy))])
        span = np.minimum(span, 50.0)
        P = rng.random((n, 3)) * span * 0.999
        X, Y, Z = (np.ascontiguousarray(P[:, k]) for k in range(3))
        vid, sx, sy, sz = orc.encode("ref", X, Y, Z, nvx, nvy, voxel, l)
        dX, dY, dZ = orc.decode("ref", vid, sx, sy, sz, nvx, nvy, voxel, l)
        out.update({f"g1{tag}_cfg": np.array([nvx, nvy, voxel, l]), f"g1{tag}_X": X, f"g1{tag}_Y": Y, f"g1{tag}_Z": Z,
                    f"g1{tag}_vid": vid, f"g1{tag}_sx": sx, f"g1{tag}_sy": sy, f"g1{tag}_sz": sz,
                    f"g1{tag}_dX": dX, f"g1{tag}_dY": dY, f"g1{tag}_dZ": dZ})

    # G2 checkSpheresOverlap<double,float>: random pairs incl. grazing ones
    n = 10000
    A = rng.random((n, 3)) * 0.5
    rA = rng.uniform(1e-3, 8e-3, n).astype(np.float32).astype(np.float64)
    rB = rng.uniform(1e-3, 8e-3, n).astype(np.float32).astype(np.float64)
    dirn = unit(rng.standard_normal((n, 3)))
    gap = rng.uniform(-0.3, 0.3, n) * (rA + rB)
    graze = rng.random(n) < 0.3
    gap[graze] = rng.uniform(-1e-9, 1e-9, int(graze.sum()))
    exact = rng.random(n) < 0.02
    gap[exact] = 0.0
    B = A + dirn * (rA + rB + gap)[:, None]
    t, CP, nrm, d = orc.spheres_overlap("ref", A, rA, B, rB)
    out.update(g2_A=A, g2_rA=rA, g2_B=B, g2_rB=rB, g2_type=t, g2_CP=CP, g2_nrm=nrm, g2_depth=d)
    # contact-point bin (getPointBinID)
    bs, nbx, nby = 0.016, 40, 40
    out.update(g2_bincfg=np.array([bs, nbx, nby]),
               g2_bin=orc.point_bin("ref", np.ascontiguousarray(CP[:, 0]), np.ascontiguousarray(CP[:, 1]),
                                    np.ascontiguousarray(CP[:, 2]), bs, nbx, nby))

    # G3 recoverCntPair, n in [2, 512] (enumeration order only; kept for completeness)
    inds, cnts = [], []
    for c in list(range(2, 40)) + [64, 100, 255, 256, 511, 512]:
        k = c * (c - 1) // 2
        sel = np.unique(np.concatenate([np.arange(min(k, 50)), rng.integers(0, k, 50), [k - 1]]))
        inds.append(sel)
        cnts.append(np.full(len(sel), c))
    inds = np.concatenate(inds).astype(np.uint32)
    cnts = np.concatenate(cnts).astype(np.uint32)
    oi = np.zeros_like(inds)
    oj = np.zeros_like(inds)
    import ctypes as C
    orc.ref().ref_recover_pair(C.c_size_t(len(inds)), C.c_void_p(inds.ctypes.data), C.c_void_p(cnts.ctypes.data),
                               C.c_void_p(oi.ctypes.data), C.c_void_p(oj.ctypes.data))
    out.update(g3_ind=inds, g3_n=cnts, g3_i=oi, g3_j=oj)

    # family mask index
    fi = rng.integers(0, 256, 3000).astype(np.uint32)
    fj = rng.integers(0, 256, 3000).astype(np.uint32)
    out.update(gm_i=fi, gm_j=fj, gm_out=orc.mask_pair("ref", fi, fj))

    # quaternion rotation (float and double vectors), Hamilton product
    q = unit(rng.standard_normal((4000, 4))).astype(np.float32)
    v = rng.uniform(-0.01, 0.01, (4000, 3)).astype(np.float32)
    out.update(gq_q=q, gq_v=v, gq_rot=orc.rotate("ref", v, q),
               gq_vd=v.astype(np.float64) * 1.0000001, gq_rotd=orc.rotate_d("ref", v.astype(np.float64) * 1.0000001, q))
    q2 = rng.standard_normal((4000, 4)).astype(np.float32)
    out.update(gh_q2=q2, gh_out=orc.hamilton("ref", q, q2))

    # G5 force models
    depth, fin, mu, crr, hist = force_inputs(rng, 3000)
    h_full, o_full = orc.force("ref", 0, depth, fin, mu, crr, hist)
    _, o_fl = orc.force("ref", 1, depth, fin, mu, crr, hist)
    out.update(g5_depth=depth, g5_fin=fin, g5_mu=mu, g5_crr=crr, g5_hist=hist, g5_hist_full=h_full, g5_out_full=o_full,
               g5_out_frictionless=o_fl)
    E, G = orc.mat_proxy("ref", fin[:, 34], fin[:, 35], fin[:, 36], fin[:, 37])
    out.update(g5_Eeff=E, g5_Geff=G)

    # G8 checkSphereEntityOverlap plane / cylinder
    n = 4000
    A = rng.uniform(-0.3, 0.3, (n, 3))
    radA = rng.uniform(1e-3, 8e-3, n).astype(np.float32)
    typeB = rng.choice([0, 2], n).astype(np.uint8)
    Bp = rng.uniform(-0.3, 0.3, (n, 3))
    dirB = unit(rng.standard_normal((n, 3))).astype(np.float32)
    axis_aligned = rng.random(n) < 0.5
    dirB[axis_aligned] = np.eye(3, dtype=np.float32)[rng.integers(0, 3, int(axis_aligned.sum()))]
    size1 = rng.uniform(0.05, 0.4, n).astype(np.float32)
    nsign = rng.choice([1.0, -1.0], n).astype(np.float32)
    beta = rng.choice([0.0, 1e-4, 5e-4], n).astype(np.float32)
    # place a share of the spheres near the surface so both outcomes occur
    near = rng.random(n) < 0.6
    pl = near & (typeB == 0)
    A[pl] = Bp[pl] + dirB[pl].astype(np.float64) * (radA[pl] * rng.uniform(0.5, 1.5, int(pl.sum())))[:, None] + \
        np.cross(dirB[pl].astype(np.float64), rng.standard_normal((int(pl.sum()), 3))) * 0.1
    t, CP, nrm, d = orc.sphere_entity("ref", A, radA, typeB, Bp, dirB, size1, nsign, beta)
    out.update(g8_A=A, g8_radA=radA, g8_typeB=typeB, g8_B=Bp, g8_dirB=dirB, g8_size1=size1, g8_nsign=nsign,
               g8_beta=beta, g8_type=t, g8_CP=CP, g8_nrm=nrm, g8_depth=d)

    # G7 check_TriangleBoxOverlap: triangles around a cubic bin, many of them grazing a face / edge / corner
    n = 6000
    half = rng.choice([0.004, 0.0075, 0.02], n).astype(np.float32)
    center = (rng.integers(0, 40, (n, 3)) * 2 + 1).astype(np.float32) * half[:, None]
    base = center + rng.uniform(-2.2, 2.2, (n, 3)).astype(np.float32) * half[:, None]
    ext = rng.choice([0.3, 1.0, 4.0], n).astype(np.float32)[:, None] * half[:, None]
    tA = base
    tB = base + rng.uniform(-1, 1, (n, 3)).astype(np.float32) * ext
    tC = base + rng.uniform(-1, 1, (n, 3)).astype(np.float32) * ext
    flat = rng.random(n) < 0.3  # axis-parallel triangles lying exactly in a box face plane or just off it
    ax = rng.integers(0, 3, n)
    off = rng.choice([-1.0, 1.0], n).astype(np.float32) * half * rng.choice([1.0, 1.0 + 1e-6, 1.0 - 1e-6], n).astype(np.float32)
    for k in range(3):
        sel = flat & (ax == k)
        for t in (tA, tB, tC):
            t[sel, k] = center[sel, k] + off[sel]
    hit = orc.tri_box(center, half, tA, tB, tC, which="ref")
    out.update(g7_center=center, g7_half=half, g7_A=tA, g7_B=tB, g7_C=tC, g7_hit=hit)

    # G7b boundingBoxIntersectBin: bin ranges of triangle bounding boxes, incl. vertices outside the world and on bin faces
    n = 4000
    bs = 0.0125
    nb = (40, 24, 16)
    bA = rng.uniform(-0.05, 0.55, (n, 3)).astype(np.float32)
    bB = bA + rng.uniform(-0.03, 0.03, (n, 3)).astype(np.float32)
    bC = bA + rng.uniform(-0.03, 0.03, (n, 3)).astype(np.float32)
    onface = rng.random(n) < 0.4
    bA[onface] = (rng.integers(0, 16, (int(onface.sum()), 3)) * bs).astype(np.float32)
    lo, hi = orc.tri_bbox(bA, bB, bC, bs, nb, which="ref")
    out.update(g7b_A=bA, g7b_B=bB, g7b_C=bC, g7b_binSize=np.float64(bs), g7b_nb=np.array(nb, np.uint32), g7b_L=lo, g7b_U=hi)

    # G9 integrator velocity pass-on fragments
    ov = rng.uniform(-2, 2, (1000, 3)).astype(np.float32)
    vu = rng.uniform(-1e-3, 1e-3, (1000, 3)).astype(np.float32)
    for scheme in (0, 1, 2):
        vo = np.zeros_like(ov)
        orc.ref().ref_vel_pass_on(C.c_size_t(len(ov)), C.c_int(scheme), C.c_void_p(ov.ctypes.data),
                                  C.c_void_p(vu.ctypes.data), C.c_void_p(vo.ctypes.data))
        out[f"g9_v_scheme{scheme}"] = vo
    out.update(g9_old_v=ov, g9_v_update=vu)

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "elementwise.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB,", len(out), "arrays")


if __name__ == "__main__":
    main()
