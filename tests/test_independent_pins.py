"""Independent checks of the three pieces of the oracle that cannot be pinned by building reference code here (SURVEY 8c
G4, G6 and the integrateVelPos body): the reference's `__global__` bodies carry placeholders its run-time compiler fills
in, and DEMCollisionKernels.cu needs device-only round-up intrinsics -- so these restatements were written by reading.
The checks below are NOT reference outputs.  They are second formulations that share no code and no arithmetic style with
the restatement: exact rational arithmetic (fractions.Fraction) for the bin-touch sets and for one integration step, a
50-digit closed form (mpmath) for the closest point on a triangle.  What they establish: the restated formulas compute the
geometric / kinematic quantity the reference's source describes, to within the rounding the restatement's fp types allow,
with the exceptions (a boundary within rounding of a bin face; the round-up branch of snap_to_face) counted and explained.
"""
import ctypes as C
from fractions import Fraction as Fr

import numpy as np
import pytest


def _bin_range(orc, pos, rad, bin_size, nb):
    pos = np.ascontiguousarray(pos, np.float64)
    rad = np.ascontiguousarray(rad, np.float64)
    lo, hi = np.zeros(len(pos), np.uint32), np.zeros(len(pos), np.uint32)
    L = orc.lib()
    L.orc_el_bin_range(C.c_size_t(len(pos)), pos.ctypes.data_as(C.c_void_p), rad.ctypes.data_as(C.c_void_p), C.c_double(bin_size),
                       C.c_uint32(nb), lo.ctypes.data_as(C.c_void_p), hi.ctypes.data_as(C.c_void_p))
    return lo, hi


def test_g4_bin_touch_sets_against_exact_interval_arithmetic(orc):
    """DEMBinSphereKernels.cu:51-74: along one axis an (inflated) sphere registers in bins floor((x - r) / s) .. floor((x + r) / s),
    clamped to [0, nb - 1].  Second formulation: the same interval with exact rationals built from the very doubles the restatement
    receives.  The fp64 quotients may land on the other side of an integer only when the exact quotient is within rounding of it:
    such cases are counted, must be rare, and must be off by exactly one bin on the side in question."""
    rng = np.random.default_rng(20240928)
    n = 20000
    s = 0.0123456789
    nb = 700
    # centres inside the grid (the kinematic frame starts at the domain's corner; a clump outside the domain is an error state
    # of the solver, not an input of this formula); extents may poke out at both ends: the clamps
    pos = rng.uniform(0.0, nb * s * (1 - 1e-9), n)
    rad = rng.uniform(0.2 * s, 2.5 * s, n)
    # spheres straddling / kissing bin faces: centre or an extreme point within a few ulp of a face
    k = rng.integers(1, nb - 1, n // 4)
    kiss = np.nextafter(k * s, np.where(rng.random(n // 4) < 0.5, -1.0, 1.0) * np.inf)
    pos[: n // 4] = kiss
    pos[n // 4: n // 2] = kiss + rad[n // 4: n // 2]        # lower extreme point on a face
    lo, hi = _bin_range(orc, pos, rad, s, nb)
    S = Fr(s)
    near = 0
    for i in range(n):
        x, r = Fr(float(pos[i])), Fr(float(rad[i]))
        ql, qh = (x - r) / S, (x + r) / S
        elo = ql.__floor__() if ql > 0 else 0
        ehi = qh.__floor__() if qh < nb else nb - 1
        ehi = max(ehi, 0)
        ok_lo, ok_hi = int(lo[i]) == elo, int(hi[i]) == ehi
        if not (ok_lo and ok_hi):
            near += 1
            assert i < n // 2, i  # never among the randomly placed spheres: only the constructed face-kissing ones
            # only where the exact quotient sits within 4 ulp of an integer, and then by one bin
            for got, want, q in ((int(lo[i]), elo, ql), (int(hi[i]), ehi, qh)):
                if got != want:
                    assert abs(got - want) == 1, (i, got, want)
                    assert abs(q - round(q)) < Fr(4, 2 ** 52) * max(abs(q), 1), (i, float(q))
    # half of the sample was put within an ulp of a bin face on purpose: there the three roundings of x/s, r/s and their sum or
    # difference decide the side, and a few per cent of those cases land in the neighbouring bin (by one, checked above)
    assert 0 < near < n // 10, near


def _closest_point_on_triangle(mp, A, B, C, P):
    """Ericson, Real-Time Collision Detection 5.1.5, in 50-digit arithmetic; returns (point, region) with region 0 = face"""
    dot = lambda u, v: sum(a * b for a, b in zip(u, v))  # noqa: E731
    sub = lambda u, v: [a - b for a, b in zip(u, v)]  # noqa: E731
    add = lambda u, v, t: [a + t * b for a, b in zip(u, v)]  # noqa: E731
    ab, ac, ap = sub(B, A), sub(C, A), sub(P, A)
    d1, d2 = dot(ab, ap), dot(ac, ap)
    if d1 <= 0 and d2 <= 0:
        return A, 1
    bp = sub(P, B)
    d3, d4 = dot(ab, bp), dot(ac, bp)
    if d3 >= 0 and d4 <= d3:
        return B, 2
    vc = d1 * d4 - d3 * d2
    if vc <= 0 and d1 >= 0 and d3 <= 0:
        return add(A, ab, d1 / (d1 - d3)), 4
    cp = sub(P, C)
    d5, d6 = dot(ab, cp), dot(ac, cp)
    if d6 >= 0 and d5 <= d6:
        return C, 3
    vb = d5 * d2 - d1 * d6
    if vb <= 0 and d2 >= 0 and d6 <= 0:
        return add(A, ac, d2 / (d2 - d6)), 5
    va = d3 * d6 - d5 * d4
    if va <= 0 and (d4 - d3) >= 0 and (d5 - d6) >= 0:
        return add(B, sub(C, B), (d4 - d3) / ((d4 - d3) + (d5 - d6))), 6
    den = 1 / (va + vb + vc)
    return add(add(A, ab, vb * den), ac, vc * den), 0


@pytest.mark.parametrize("directional", [False, True])
def test_g6_triangle_sphere_against_high_precision_closed_form(orc, directional):
    """DEMCollisionKernels.cu:16-236 (snap_to_face, triangle_sphere_CD[_directional]) restated in oracle/deme_oracle.cpp, fp64
    instantiation, against the closest point on the triangle in 50-digit arithmetic: 1500 seeded cases spread over the face, the
    three edge and the three vertex regions, touching and not touching.  The reference's device code rounds two intermediates of
    snap_to_face UP (__drcp_ru / __dmul_ru); the restatement does the same through the floating-point environment.  That moves the
    snapped point by ~1 ulp along an edge, far inside the tolerances here."""
    mp = pytest.importorskip("mpmath")
    mp.mp.dps = 50
    rng = np.random.default_rng(7 if directional else 8)
    n = 1500
    A = rng.uniform(-1, 1, (n, 3))
    B = A + rng.uniform(-1, 1, (n, 3))
    Cc = A + rng.uniform(-1, 1, (n, 3))
    # sphere centres aimed at all Voronoi regions: barycentric coordinates partly outside [0, 1], plus an offset along the normal
    u, v = rng.uniform(-0.8, 1.6, n), rng.uniform(-0.8, 1.6, n)
    nrm = np.cross(B - A, Cc - A)
    area2 = np.linalg.norm(nrm, axis=1, keepdims=True)
    nrm /= area2
    off = rng.uniform(-0.6, 0.6, (n, 1))
    P = A + u[:, None] * (B - A) + v[:, None] * (Cc - A) + off * nrm
    r = rng.uniform(0.05, 0.7, n)
    hit, nr, depth, pt = orc.tri_sphere(A, B, Cc, P, r, directional=directional)
    regions = np.zeros(7, int)
    checked = 0
    for i in range(n):
        if area2[i, 0] < (0.05 if directional else 1e-3):
            continue  # slivers: the two formulations may pick different (equally close) points
        a, b, c, p = ([mp.mpf(float(x)) for x in V[i]] for V in (A, B, Cc, P))
        q, reg = _closest_point_on_triangle(mp, a, b, c, p)
        regions[reg] += 1
        fn = [mp.mpf(float(x)) for x in nrm[i]]
        h = sum((pi - ai) * ni for pi, ai, ni in zip(p, a, fn))
        rr = mp.mpf(float(r[i]))
        if reg == 0:
            e_depth, e_n = h - rr, fn
            e_hit = (e_depth < 0) if directional else (abs(h) < rr)
        else:
            d = [pi - qi for pi, qi in zip(p, q)]
            dist = mp.sqrt(sum(x * x for x in d))
            e_depth, e_n = dist - rr, [x / dist for x in d]
            e_hit = (e_depth < 0 and h < rr) if directional else (e_depth < 0 and abs(h) < rr)
        scale = float(max(np.abs(A[i]).max(), np.abs(B[i]).max(), np.abs(Cc[i]).max(), 1.0))
        # Precision the restatement carries, by the reference's own choice of types: the directional flavour is evaluated in fp32
        # (triangle_sphere_CD_directional<float3, float>); the other in fp64 except the face normal, which the reference
        # normalises with the single-precision reciprocal square root (normalize() in CUDAMathHelpers.cuh) -- so a face-region
        # depth and every normal are fp32-accurate, edge / vertex depths and all contact points fp64-accurate
        # (fp32 barycentric arithmetic cancels: a few 1e-5 of the triangle's size on badly shaped triangles)
        tol_pt = (2e-4 if directional else 1e-12) * scale
        tol_depth = (2e-4 if directional else (3e-7 if reg == 0 else 1e-12)) * scale
        tol_n = 2e-3 if directional else 3e-7
        # decisions: identical unless the case sits within that precision of the touching limit
        if abs(e_depth) > 10 * tol_depth and abs(abs(h) - rr) > 10 * tol_depth:
            assert bool(hit[i]) == bool(e_hit), (i, reg, float(e_depth), float(h))
        assert abs(mp.mpf(float(depth[i])) - e_depth) < tol_depth, (i, reg)
        assert max(abs(mp.mpf(float(pt[i, k])) - q[k]) for k in range(3)) < tol_pt, (i, reg)
        assert max(abs(mp.mpf(float(nr[i, k])) - e_n[k]) for k in range(3)) < tol_n, (i, reg)
        checked += 1
    assert checked >= (900 if directional else 1000) and (regions >= 50).all(), (checked, regions)  # every region is represented


def _exact_step(p, st, i, g, scheme):
    """one explicit step of owner i with no contacts, in exact rationals (square root in 50 digits): what integrateVelPos
    (DEMIntegrationKernels.cu:100-236) describes -- v += g h; x += v_used h; q <- normalise(q * (1, h/2 omega))"""
    import mpmath as mp
    h = Fr(float(p.h))
    v0 = [Fr(float(st[k][i])) for k in ("vX", "vY", "vZ")]
    w0 = [Fr(float(st[k][i])) for k in ("omgBarX", "omgBarY", "omgBarZ")]
    upd = [Fr(float(np.float32(gk))) * h for gk in g]
    v1 = [a + b for a, b in zip(v0, upd)]
    used = {0: v0, 1: v1, 2: [a + b / 2 for a, b in zip(v0, upd)]}[scheme]
    x1 = [a + b * h for a, b in zip(st["_X"][i], used)]
    qw, qx, qy, qz = (Fr(float(st[k][i])) for k in ("oriQw", "oriQx", "oriQy", "oriQz"))
    hw = [c * h / 2 for c in w0]  # contact-free: omega does not change
    nw = qw - qx * hw[0] - qy * hw[1] - qz * hw[2]
    nx = qw * hw[0] + qx + qy * hw[2] - qz * hw[1]
    ny = qw * hw[1] - qx * hw[2] + qy + qz * hw[0]
    nz = qw * hw[2] + qx * hw[1] - qy * hw[0] + qz
    ln = mp.sqrt(mp.mpf(nw.numerator) / nw.denominator * mp.mpf(nw.numerator) / nw.denominator + sum(
        (mp.mpf(c.numerator) / c.denominator) ** 2 for c in (nx, ny, nz)))
    q1 = [float(mp.mpf(c.numerator) / c.denominator / ln) for c in (nw, nx, ny, nz)]
    return [float(c) for c in v1], [float(c) for c in x1], q1


@pytest.mark.parametrize("scheme", [0, 1, 2])
def test_integrator_body_against_exact_rational_step(pkg, orc, scheme):
    """One contact-free step of 200 free clumps (gravity, random velocities, spins, orientations) by the oracle against the same
    update in exact rational arithmetic, for the three velocity pass-on schemes (forward Euler / centred difference / extended
    Taylor).  Bounds: velocities to 1 fp32 ulp; positions to one sub-voxel unit l (the codec truncates) plus the fp32 rounding of v times h;
    quaternions to 4 fp32 ulp."""
    pytest.importorskip("mpmath")
    import mpmath as mp
    mp.mp.dps = 50
    b = pkg.model.packed_bed(200, seed=5, cd_freq=0, spacing_mult=6.0)  # far apart: no contacts
    b.SetIntegrator(["FORWARD_EULER", "CENTERED_DIFFERENCE", "EXTENDED_TAYLOR"][scheme])
    p, sc = b.Initialize()
    n = int(sc.nOwnerClumps)
    rng = np.random.default_rng(11)
    sim = orc.make_sim(pkg, p, sc)
    st = sim.download_state()
    for k in ("vX", "vY", "vZ"):
        st[k][:n] = rng.uniform(-2, 2, n).astype(np.float32)
    for k in ("omgBarX", "omgBarY", "omgBarZ"):
        st[k][:n] = rng.uniform(-50, 50, n).astype(np.float32)
    sim.upload_state({k: st[k] for k in st if not k.startswith(("a", "alpha"))})
    st = sim.download_state()
    X0 = pkg.model.decode_positions(st["voxelID"], st["locX"], st["locY"], st["locZ"], p.nvXp2, p.nvYp2, p.voxelSize, p.l)
    st["_X"] = [[Fr(float(c)) for c in row] for row in X0]
    sim.step(1)
    assert int(sim.counts().nContacts) == 0
    s1 = sim.download_state()
    X1 = pkg.model.decode_positions(s1["voxelID"], s1["locX"], s1["locY"], s1["locZ"], p.nvXp2, p.nvYp2, p.voxelSize, p.l)
    g = (p.Gx, p.Gy, p.Gz)
    for i in range(n):
        v, x, q = _exact_step(p, st, i, g, scheme)
        for k, name in enumerate(("vX", "vY", "vZ")):
            assert abs(float(s1[name][i]) - v[k]) <= 1.2e-7 * max(abs(v[k]), 1e-3), (i, name)
        for k in range(3):
            # one sub-voxel unit (the codec truncates) + the fp32 rounding of the velocity that is multiplied by h + fp64 rounding
            assert abs(X1[i, k] - x[k]) <= p.l + 3e-7 * abs(v[k]) * p.h + 1e-15 * max(abs(x[k]), 1.0), (i, k, X1[i, k] - x[k])
        for k, name in enumerate(("oriQw", "oriQx", "oriQy", "oriQz")):
            assert abs(float(s1[name][i]) - q[k]) <= 5e-7, (i, name)
