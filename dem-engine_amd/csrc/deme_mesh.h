// deme_mesh.h -- sphere-triangle geometry (gfx950).
//
// Replaces kernel/DEMCollisionKernels.cu (snap_to_face, triangle_sphere_CD[_directional]),
// kernel/DEMTriangleBoxIntersect.cu (triangle/AABB separating-axis test) and the sandwich construction of
// kernel/DEMBinTriangleKernels.cu:7-53.  Self-contained (included by deme_force.h, also under hipRTC).
// Arithmetic types follow the reference instantiations: <float3,float> in detection, <double3,double> in the
// force kernel, <float3,double> for the contact-point snap of the bin sweep.
#pragma once
#include "deme_device.h"

namespace deme_dev {

struct __attribute__((aligned(16))) TriRec {  // owner-local nodes (relPosNode1..3), owner, material
    float n1[3], n2[3], n3[3];
    uint32_t owner;
    uint32_t mat;
    uint32_t pad;
};
static_assert(sizeof(TriRec) == 48, "TriRec is 48 bytes");

struct __attribute__((aligned(16))) TriWorld {  // the two margin-offset ("sandwich") triangles, world frame
    float a1[3], a2[3], a3[3], b1[3], b2[3], b3[3];
    uint32_t owner;
    uint32_t family;
};
static_assert(sizeof(TriWorld) == 80, "TriWorld is 80 bytes");

template <typename T>
struct v3 {
    T x, y, z;
};
template <typename T>
__device__ inline v3<T> vsub(v3<T> a, v3<T> b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
template <typename T>
__device__ inline v3<T> vadd(v3<T> a, v3<T> b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
template <typename T>
__device__ inline v3<T> vscale(T s, v3<T> a) { return {s * a.x, s * a.y, s * a.z}; }
template <typename T>
__device__ inline T vdot(v3<T> a, v3<T> b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <typename T>
__device__ inline v3<T> vcross(v3<T> a, v3<T> b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
// normalize(): v * rsqrtf(dot(v,v)) for float3 AND double3 (CUDAMathHelpers.cuh:1090,1402); rsqrtf in its
// host form 1.0f / sqrtf(x) (CUDAMathHelpers.cuh:66-68) so that the CPU oracle can reproduce it exactly
template <typename T>
__device__ inline v3<T> vnormalize(v3<T> v) {
    const T il = (T)(1.0f / sqrtf((float)vdot(v, v)));
    return {v.x * il, v.y * il, v.z * il};
}
__device__ inline float vlen(v3<float> v) { return sqrtf(vdot(v, v)); }
__device__ inline double vlen(v3<double> v) { return sqrt(vdot(v, v)); }

// __drcp_ru / __dmul_ru (DEMCollisionKernels.cu:76-78): HIP has no directed-rounding variants, so they are
// emulated with one exact fma residual (valid for finite, normal operands)
__device__ inline double rcp_ru(double x) {
    double r = 1.0 / x;
    const double e = fma(-x, r, 1.0);
    if ((e > 0.0 && x > 0.0) || (e < 0.0 && x < 0.0))
        r = __longlong_as_double(__double_as_longlong(r) + ((r > 0.0) ? 1 : -1));  // nextafter(r, +inf)
    return r;
}
__device__ inline double mul_ru(double a, double b) {
    double p = a * b;
    const double e = fma(a, b, -p);
    if (e > 0.0) {
        if (p == 0.0)
            p = __longlong_as_double(1);  // smallest subnormal
        else
            p = __longlong_as_double(__double_as_longlong(p) + ((p > 0.0) ? 1 : -1));
    }
    return p;
}

// DEMCollisionKernels.cu:16-82.  VT: point component type; ST: barycentric scalar type (the reference's T2)
template <typename VT, typename ST>
__device__ inline bool snap_to_face(v3<VT> A, v3<VT> B, v3<VT> C, v3<VT> P, v3<VT>& res) {
    const v3<VT> AB = vsub(B, A), AC = vsub(C, A), AP = vsub(P, A);
    const ST d1 = vdot(AB, AP), d2 = vdot(AC, AP);
    if (d1 <= 0 && d2 <= 0) {
        res = A;
        return true;
    }
    const v3<VT> BP = vsub(P, B);
    const ST d3 = vdot(AB, BP), d4 = vdot(AC, BP);
    if (d3 >= 0 && d4 <= d3) {
        res = B;
        return true;
    }
    const ST vc = d1 * d4 - d3 * d2;
    if (vc <= 0 && d1 >= 0 && d3 <= 0) {
        const ST v = d1 / (d1 - d3);
        res = vadd(A, vscale((VT)v, AB));
        return true;
    }
    const v3<VT> CP = vsub(P, C);
    const ST d5 = vdot(AB, CP), d6 = vdot(AC, CP);
    if (d6 >= 0 && d5 <= d6) {
        res = C;
        return true;
    }
    const ST vb = d5 * d2 - d1 * d6;
    if (vb <= 0 && d2 >= 0 && d6 <= 0) {
        const ST w = d2 / (d2 - d6);
        res = vadd(A, vscale((VT)w, AC));
        return true;
    }
    const ST va = d3 * d6 - d5 * d4;
    if (va <= 0 && (d4 - d3) >= 0 && (d5 - d6) >= 0) {
        const ST w = (d4 - d3) / ((d4 - d3) + (d5 - d6));
        res = vadd(B, vscale((VT)w, vsub(C, B)));
        return true;
    }
    const ST denom = (ST)rcp_ru((double)(va + vb + vc));
    const ST v = (ST)mul_ru((double)vb, (double)denom);
    const ST w = (ST)mul_ru((double)vc, (double)denom);
    res = vadd(vadd(A, vscale((VT)v, AB)), vscale((VT)w, AC));
    return false;
}

// DEMCollisionKernels.cu:99-159 and :177-236
template <typename T, bool DIRECTIONAL>
__device__ inline bool tri_sphere_cd(v3<T> A, v3<T> B, v3<T> C, v3<T> sp, T radius, v3<T>& normal, T& depth, v3<T>& pt1) {
    const v3<T> face_n = vnormalize(vcross(vsub(B, A), vsub(C, A)));
    const T h = vdot(vsub(sp, A), face_n);
    v3<T> faceLoc;
    bool in_contact;
    if (!snap_to_face<T, T>(A, B, C, sp, faceLoc)) {
        depth = h - radius;
        normal = face_n;
        pt1 = faceLoc;
        if (DIRECTIONAL)
            in_contact = !(depth >= 0.);
        else
            in_contact = !(h >= radius || h <= -radius);
    } else {
        normal = vsub(sp, faceLoc);
        const T dist = vlen(normal);
        depth = dist - radius;
        normal = vscale((T)(1.0 / dist), normal);
        pt1 = faceLoc;
        if (DIRECTIONAL)
            in_contact = !(depth >= 0. || h >= radius);
        else
            in_contact = !(depth >= 0. || h >= radius || h <= -radius);
    }
    return in_contact;
}

// DEMTriangleBoxIntersect.cu:176-374 (Akenine-Moller separating-axis test, fp32)
__device__ inline bool axis_sep(float pa, float pb, float rad) {
    const float mn = (pa < pb) ? pa : pb, mx = (pa < pb) ? pb : pa;
    return mn > rad || mx < -rad;
}
__device__ inline bool tri_box_overlap(const float bc[3], float bh, const float* vA, const float* vB, const float* vC) {
    const float v0[3] = {vA[0] - bc[0], vA[1] - bc[1], vA[2] - bc[2]};
    const float v1[3] = {vB[0] - bc[0], vB[1] - bc[1], vB[2] - bc[2]};
    const float v2[3] = {vC[0] - bc[0], vC[1] - bc[1], vC[2] - bc[2]};
    const float e0[3] = {v1[0] - v0[0], v1[1] - v0[1], v1[2] - v0[2]};
    const float e1[3] = {v2[0] - v1[0], v2[1] - v1[1], v2[2] - v1[2]};
    const float e2[3] = {v0[0] - v2[0], v0[1] - v2[1], v0[2] - v2[2]};
    float fex, fey, fez;
    fex = fabsf(e0[0]), fey = fabsf(e0[1]), fez = fabsf(e0[2]);
    if (axis_sep(e0[2] * v0[1] - e0[1] * v0[2], e0[2] * v2[1] - e0[1] * v2[2], fez * bh + fey * bh)) return false;
    if (axis_sep(-e0[2] * v0[0] + e0[0] * v0[2], -e0[2] * v2[0] + e0[0] * v2[2], fez * bh + fex * bh)) return false;
    if (axis_sep(e0[1] * v1[0] - e0[0] * v1[1], e0[1] * v2[0] - e0[0] * v2[1], fey * bh + fex * bh)) return false;
    fex = fabsf(e1[0]), fey = fabsf(e1[1]), fez = fabsf(e1[2]);
    if (axis_sep(e1[2] * v0[1] - e1[1] * v0[2], e1[2] * v2[1] - e1[1] * v2[2], fez * bh + fey * bh)) return false;
    if (axis_sep(-e1[2] * v0[0] + e1[0] * v0[2], -e1[2] * v2[0] + e1[0] * v2[2], fez * bh + fex * bh)) return false;
    if (axis_sep(e1[1] * v0[0] - e1[0] * v0[1], e1[1] * v1[0] - e1[0] * v1[1], fey * bh + fex * bh)) return false;
    fex = fabsf(e2[0]), fey = fabsf(e2[1]), fez = fabsf(e2[2]);
    if (axis_sep(e2[2] * v0[1] - e2[1] * v0[2], e2[2] * v1[1] - e2[1] * v1[2], fez * bh + fey * bh)) return false;
    if (axis_sep(-e2[2] * v0[0] + e2[0] * v0[2], -e2[2] * v1[0] + e2[0] * v1[2], fez * bh + fex * bh)) return false;
    if (axis_sep(e2[1] * v1[0] - e2[0] * v1[1], e2[1] * v2[0] - e2[0] * v2[1], fey * bh + fex * bh)) return false;
    for (int d = 0; d < 3; d++) {
        float mn = v0[d], mx = v0[d];
        if (v1[d] < mn) mn = v1[d];
        if (v1[d] > mx) mx = v1[d];
        if (v2[d] < mn) mn = v2[d];
        if (v2[d] > mx) mx = v2[d];
        if (mn > bh || mx < -bh)
            return false;
    }
    const float nr[3] = {e0[1] * e1[2] - e0[2] * e1[1], e0[2] * e1[0] - e0[0] * e1[2], e0[0] * e1[1] - e0[1] * e1[0]};
    float vmin[3], vmax[3];
    for (int q = 0; q < 3; q++) {
        const float v = v0[q];
        if (nr[q] > 0.0f) {
            vmin[q] = -bh - v;
            vmax[q] = bh - v;
        } else {
            vmin[q] = bh - v;
            vmax[q] = -bh - v;
        }
    }
    if (nr[0] * vmin[0] + nr[1] * vmin[1] + nr[2] * vmin[2] > 0.0f)
        return false;
    return nr[0] * vmax[0] + nr[1] * vmax[1] + nr[2] * vmax[2] >= 0.0f;
}

// DEMBinTriangleKernels.cu:7-20 (sandwichVertex), DEMHelperKernels.cuh:215-225 (triangleIncenter)
__device__ inline v3<float> sandwich_vertex(v3<float> vertex, v3<float> incenter, v3<float> side, v3<float> normal, float beta) {
    const v3<float> ev = vnormalize(vsub(vertex, incenter));
    const v3<float> nev{-ev.x, -ev.y, -ev.z};
    const float cos_half = vdot(nev, side) / vlen(side);
    const float enlarge = (float)(beta / sqrt(1. - cos_half * cos_half));
    vertex = vadd(vertex, vscale(enlarge, ev));
    vertex = vadd(vertex, vscale(beta, normal));
    return vertex;
}
__device__ inline v3<float> tri_incenter(v3<float> p1, v3<float> p2, v3<float> p3) {
    const float a = vlen(vsub(p2, p3)), b = vlen(vsub(p1, p3)), c = vlen(vsub(p1, p2));
    return {(a * p1.x + b * p2.x + c * p3.x) / (a + b + c), (a * p1.y + b * p2.y + c * p3.y) / (a + b + c),
            (a * p1.z + b * p2.z + c * p3.z) / (a + b + c)};
}

}  // namespace deme_dev
