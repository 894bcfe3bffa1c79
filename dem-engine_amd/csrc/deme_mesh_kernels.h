// deme_mesh_kernels.h -- sphere-triangle contact detection kernels (gfx950).
//
// Replaces kernel/DEMBinTriangleKernels.cu (makeTriangleSandwich, getNumberOfBinsEachTriangleTouches,
// populateBinTriangleTouchingPairs) and kernel/DEMContactKernels_SphereTriangle.cu
// (getNumberOfSphTriContactsEachBin + populateTriSphContactsEachBin, single sweep), plus the host-side
// merge of triangle-active and sphere-active bins (algorithms/DEMCubContactDetection.cu:400-414), which
// becomes a binary search over the sphere incidence keys on the device.
#pragma once
#include "deme_kernels.h"
#include "deme_mesh.h"

namespace deme_dev {

__device__ inline void tri_bin_bounds(const DevParams& p, const float* v1, const float* v2, const float* v3, int L[3], int U[3]) {
    // boundingBoxIntersectBin, DEMHelperKernels.cuh:528-565
    const int nbm[3] = {(int)p.nbX - 1, (int)p.nbY - 1, (int)p.nbZ - 1};
    for (int d = 0; d < 3; d++) {
        const float mn = fminf(v1[d], fminf(v2[d], v3[d])), mx = fmaxf(v1[d], fmaxf(v2[d], v3[d]));
        const float lo = (float)(mn - 0.001 * p.binSize), hi = (float)(mx + 0.001 * p.binSize);
        const float ql = (float)(lo / p.binSize), qh = (float)(hi / p.binSize);
        const float cl = fminf(fmaxf(ql, 0.f), (float)nbm[d]), ch = fminf(fmaxf(qh, 0.f), (float)nbm[d]);
        L[d] = min(L[d], (int)cl);
        U[d] = max(U[d], (int)ch);
    }
}

// per triangle: world-frame sandwich triangles, merged bin bounds, number of bins passing the SAT test
__global__ __launch_bounds__(256) void k_tri_prep(const DevParams p, uint32_t nTri, const TriRec* __restrict__ tris,
                                                  const OwnerRec* __restrict__ owners, TriWorld* __restrict__ tw,
                                                  int4* __restrict__ triLo, int4* __restrict__ triHi,
                                                  uint32_t* __restrict__ counts) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t == 0)
        counts[nTri] = 0;
    if (t >= nTri)
        return;
    const TriRec tr = tris[t];
    const OwnerRec o = load_owner(owners, tr.owner);
    const v3<float> p1{tr.n1[0], tr.n1[1], tr.n1[2]}, p2{tr.n2[0], tr.n2[1], tr.n2[2]}, p3{tr.n3[0], tr.n3[1], tr.n3[2]};
    const v3<float> inc = tri_incenter(p1, p2, p3);
    const v3<float> n = vnormalize(vcross(vsub(p2, p1), vsub(p3, p1)));
    const v3<float> nn{-n.x, -n.y, -n.z};
    const float beta = o.margin;
    const v3<float> loc[6] = {sandwich_vertex(p1, inc, vsub(p2, p1), n, beta),  sandwich_vertex(p2, inc, vsub(p3, p2), n, beta),
                              sandwich_vertex(p3, inc, vsub(p1, p3), n, beta),  sandwich_vertex(p1, inc, vsub(p2, p1), nn, beta),
                              sandwich_vertex(p3, inc, vsub(p1, p3), nn, beta), sandwich_vertex(p2, inc, vsub(p3, p2), nn, beta)};
    const d3 op = decode_pos(o.voxelID, o.locX, o.locY, o.locZ, p);
    const RotM m = rot_coeffs(o.qw, o.qx, o.qy, o.qz);
    TriWorld w;
    float* dst[6] = {w.a1, w.a2, w.a3, w.b1, w.b2, w.b3};
    for (int k = 0; k < 6; k++) {
        const f3 r = rot_apply(m, mk3(loc[k].x, loc[k].y, loc[k].z));
        dst[k][0] = (float)(op.x + r.x), dst[k][1] = (float)(op.y + r.y), dst[k][2] = (float)(op.z + r.z);
    }
    w.owner = tr.owner;
    w.family = fam_of(o.family);
    tw[t] = w;
    int L[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, U[3] = {-1, -1, -1};
    tri_bin_bounds(p, w.a1, w.a2, w.a3, L, U);
    tri_bin_bounds(p, w.b1, w.b2, w.b3, L, U);
    triLo[t] = make_int4(L[0], L[1], L[2], 0);
    triHi[t] = make_int4(U[0], U[1], U[2], 0);
    const float bh = (float)(p.binSize / 2. + 0.001 * p.binSize);
    uint32_t cnt = 0;
    for (int i = L[0]; i <= U[0]; i++)
        for (int j = L[1]; j <= U[1]; j++)
            for (int k = L[2]; k <= U[2]; k++) {
                const float bc[3] = {(float)(p.binSize * i + p.binSize / 2.), (float)(p.binSize * j + p.binSize / 2.),
                                     (float)(p.binSize * k + p.binSize / 2.)};
                if (tri_box_overlap(bc, bh, w.a1, w.a2, w.a3) || tri_box_overlap(bc, bh, w.b1, w.b2, w.b3))
                    cnt++;
            }
    counts[t] = cnt;
}

__global__ __launch_bounds__(256) void k_tri_fill(const DevParams p, uint32_t nTri, const TriWorld* __restrict__ tw,
                                                  const int4* __restrict__ triLo, const int4* __restrict__ triHi,
                                                  const uint32_t* __restrict__ offsets, uint32_t* __restrict__ keys,
                                                  uint32_t* __restrict__ vals, uint64_t cap) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nTri)
        return;
    const TriWorld w = tw[t];
    const int4 L = triLo[t], U = triHi[t];
    const float bh = (float)(p.binSize / 2. + 0.001 * p.binSize);
    uint64_t off = offsets[t];
    for (int i = L.x; i <= U.x; i++)
        for (int j = L.y; j <= U.y; j++)
            for (int k = L.z; k <= U.z; k++) {
                const float bc[3] = {(float)(p.binSize * i + p.binSize / 2.), (float)(p.binSize * j + p.binSize / 2.),
                                     (float)(p.binSize * k + p.binSize / 2.)};
                if (tri_box_overlap(bc, bh, w.a1, w.a2, w.a3) || tri_box_overlap(bc, bh, w.b1, w.b2, w.b3)) {
                    if (off < cap) {
                        keys[off] = (uint32_t)i + (uint32_t)j * p.nbX + (uint32_t)k * p.nbX * p.nbY;
                        vals[off] = t;
                    }
                    off++;
                }
            }
}

#define TRI_WBUF 256u
// all lanes of one wavefront: append wbuf[0..n) to the global key list (LDS operations of one wavefront complete in order)
__device__ inline void tri_flush(const uint64_t* wbuf, uint32_t n, uint32_t lane, const KeyArena& ar) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    unsigned long long base = 0;
    if (lane == 0)
        base = arena_reserve(ar, arena_seg(ar), (unsigned long long)n);
    base = __shfl(base, 0);
    for (uint32_t i = lane; i < n; i += 64u)
        arena_store(ar, arena_seg(ar), base + i, wbuf[i]);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// one thread per (bin, triangle) incidence: every sphere registered in that bin (binary search over the
// bin-sorted sphere incidence list) is tested against both sandwich triangles
__global__ __launch_bounds__(256) void k_tri_sweep(const DevParams p, uint32_t TP, const uint32_t* __restrict__ triKeys,
                                                   const uint32_t* __restrict__ triIds, const TriWorld* __restrict__ tw,
                                                   uint32_t P, const uint32_t* __restrict__ sphKeys,
                                                   const uint32_t* __restrict__ sphIds, const GeoRec* __restrict__ geo,
                                                   const uint16_t* __restrict__ sphFam, const KeyArena ar) {
    __shared__ uint64_t wbufAll[4][TRI_WBUF];
    uint64_t* wbuf = wbufAll[threadIdx.x >> 6];
    uint32_t nBuf = 0;
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u;
    const bool valid = e < TP;
    uint32_t bin = 0, t = 0, lo = 0, hi = 0;
    TriWorld w;
    if (valid) {
        bin = triKeys[e];
        t = triIds[e];
        w = tw[t];
        uint32_t a = 0, b = P;
        while (a < b) {
            const uint32_t mid = a + ((b - a) >> 1);
            if (sphKeys[mid] < bin)
                a = mid + 1;
            else
                b = mid;
        }
        lo = a;
        b = P;
        while (a < b) {
            const uint32_t mid = a + ((b - a) >> 1);
            if (sphKeys[mid] <= bin)
                a = mid + 1;
            else
                b = mid;
        }
        hi = a;
    }
    const v3<float> A1{w.a1[0], w.a1[1], w.a1[2]}, A2{w.a2[0], w.a2[1], w.a2[2]}, A3{w.a3[0], w.a3[1], w.a3[2]};
    const v3<float> B1{w.b1[0], w.b1[1], w.b1[2]}, B2{w.b2[0], w.b2[1], w.b2[2]}, B3{w.b3[0], w.b3[1], w.b3[2]};
    for (uint32_t x = lo;; x++) {
        const bool act = valid && x < hi;
        if (!__any(act))
            break;
        bool hit = false;
        uint64_t key = 0;
        if (act) {
            const uint32_t sp = sphIds[x];
            const GeoRec g = geo[sp];
            bool ok = g.owner != w.owner;  // (the caller's owner number of the sphere; a mesh owner has the same number inside and out)
            float am = 0.f;
            if (ok && (p.hasGhosts & 2u))
                ok = !ghost_of(sphFam[sp]);  // (its own rank lists a ghost sphere's mesh contacts)
            if (ok && !p.familyTrivial) {
                const uint32_t fS = fam_of(sphFam[sp]);
                ok = p.familyMasks[mask_pair(fS, w.family)] == 0;
                const float ea = p.familyExtra[fS], eb = p.familyExtra[w.family];
                am = (ea < eb) ? ea : eb;
            }
            if (ok) {
                const v3<float> sph{(float)g.x, (float)g.y, (float)g.z};
                v3<float> cp, nr;
                float depth;
                bool inA = tri_sphere_cd<float, true>(A1, A2, A3, sph, g.r, nr, depth, cp);
                inA = inA && (-depth > am);
                bool inB = tri_sphere_cd<float, true>(B1, B2, B3, sph, g.r, nr, depth, cp);
                inB = inB && (-depth > am);
                if (inA || inB) {
                    snap_to_face<float, double>(A1, A2, A3, sph, cp);
                    hit = point_bin((double)cp.x, (double)cp.y, (double)cp.z, p) == bin;
                    key = make_key(DEME_KEY_CLASS_SM, sp, t);
                }
            }
        }
        // hits are collected in a wavefront-private LDS buffer and appended to the global list with ONE reservation per
        // flush: a reservation per loop iteration made this kernel atomic-bound (same-address atomics ~12 ns each)
        const unsigned long long m = __ballot(hit);
        if (m) {
            if (hit)
                wbuf[nBuf + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = key;
            nBuf += (uint32_t)__popcll(m);  // wave-uniform
        }
        if (nBuf > TRI_WBUF - 64u) {
            tri_flush(wbuf, nBuf, lane, ar);
            nBuf = 0;
        }
    }
    if (nBuf)
        tri_flush(wbuf, nBuf, lane, ar);
}

__global__ __launch_bounds__(256) void k_pack_tris(uint32_t nTri, TriRec* tris, const float* n1, const float* n2, const float* n3) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nTri)
        return;
    for (int k = 0; k < 3; k++) {
        tris[t].n1[k] = n1[3 * t + k];
        tris[t].n2[k] = n2[3 * t + k];
        tris[t].n3[k] = n3[3 * t + k];
    }
}

__global__ __launch_bounds__(256) void k_family_material_tris(uint32_t n, TriRec* __restrict__ tris, const OwnerRec* __restrict__ owners,
                                                              uint32_t family, uint32_t material) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n && (owners[tris[t].owner].family & 0xFFu) == family)
        tris[t].mat = material;
}

}  // namespace deme_dev
