// deme_force_fast.h -- the contact-force kernel of the FAST arithmetic mode (the default; deme_set_arith_mode).
//
// Same physics as deme_force.h (kernel/DEMCalcForceKernels.cu:44-267 with FullHertzianForceModel.cu /
// FrictionlessHertzianForceModel.cu), re-associated so that the per-owner work is done once per owner per step by the
// integrator instead of once per contact end here:
//   * the two owner records of a contact are fetched by the whole wavefront, four lanes per 64-byte record (one load
//     instruction touches 16 full records instead of 64 scattered 16-byte pieces: the vector L1 serves one 64-byte request per
//     cycle, and this kernel is bound by that request rate and by memory latency, not by arithmetic), and transposed to one
//     record per lane through LDS;
//   * the owner-to-owner offset is the EXACT integer difference of the two encoded positions times l (one rounding), not the
//     difference of two decoded fp64 positions;
//   * everything is evaluated in the WORLD frame: the velocity of the contact point is v + w x r, the torque r x F; the
//     per-contact contributions are the world-frame force and torque about the owner's centre, and the integrator turns
//     their per-owner sums into a = F / m and alpha = R^T tau / I once per owner (the reference rotates every contact's
//     force into the body frame and divides per contact: ForceInKernelReductionStrat.cu:2-33);
//   * the overlap is formed without an fp64 square root: depth = ((rA + rB)^2 - d^2) / ((rA + rB) + |d|) with the
//     numerator -- the only place where cancellation happens -- in fp64;
//   * physics-only divisions and square roots use the 1-ulp hardware forms.
// Decisions (is this pair in contact?) are the sign of the fp64 numerator: the same predicate as the reference's fp64
// comparison.  Results differ from the bit-exact mode by fp32 rounding only; tests/test_fast_mode.py states the tolerance.
#pragma once
#include "deme_force.h"

// this header's arithmetic is free to contract a * b + c into one FMA (the translation unit is compiled with
// -ffp-contract=off for the decision code and the bit-exact mode)
#pragma clang fp contract(fast)

namespace deme_dev {

// contracted twins of the small vector helpers (the ones of deme_device.h keep their no-contraction semantics when inlined)
__device__ inline RotM frot_coeffs(float w, float x, float y, float z) {
    RotM m;
    m.xx = 2.0f * (w * w + x * x) - 1.0f;
    m.xy = 2.0f * (x * y - w * z);
    m.xz = 2.0f * (x * z + w * y);
    m.yx = 2.0f * (x * y + w * z);
    m.yy = 2.0f * (w * w + y * y) - 1.0f;
    m.yz = 2.0f * (y * z - w * x);
    m.zx = 2.0f * (x * z - w * y);
    m.zy = 2.0f * (y * z + w * x);
    m.zz = 2.0f * (w * w + z * z) - 1.0f;
    return m;
}
__device__ inline f3 frot_apply(const RotM& m, f3 v) {
    return mk3(m.xx * v.x + m.xy * v.y + m.xz * v.z, m.yx * v.x + m.yy * v.y + m.yz * v.z, m.zx * v.x + m.zy * v.y + m.zz * v.z);
}
__device__ inline f3 fcross(f3 a, f3 b) { return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
__device__ inline float fdot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ inline f3 fsub(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ inline f3 fadd(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ inline f3 fscale(float s, f3 a) { return mk3(s * a.x, s * a.y, s * a.z); }
__device__ inline f3 faxpy(float s, f3 a, f3 b) { return mk3(s * a.x + b.x, s * a.y + b.y, s * a.z + b.z); }  // s a + b

// LDS stride of a staged OwnerRec in 16-byte units: 80 bytes, so that the 64 lanes' own-record reads (ds_read_b128 at lane * 80)
// fall on distinct banks (a 64-byte stride would put every fourth lane on the same ones)
#define DEME_REC_LDS_STRIDE 5

// streamed-once data (gather records, history, contribution records) can bypass the caches' retention so that the owner records
// several contacts share stay resident: non-temporal loads and stores for the streams (measured: force pass -2.3 %, the integrator that
// gathers the B-side records afterwards -2 %)
#ifndef DEME_FAST_NT
#define DEME_FAST_NT 1
#endif
#ifndef DEME_REC24
#define DEME_REC24 1  // 1: the tile sums and the crossing contacts' records are 24 bytes (six floats as three 8-byte pieces) instead of
                      // two 16-byte pieces with two unused floats: -22 MB written by the force pass and -22 MB read by the integrator at
                      // 1e6 clumps (the integrator moves its bytes at the copy rate; the force pass answers to bytes with ~0.3)
#endif
typedef unsigned int nt_u4 __attribute__((ext_vector_type(4)));
typedef unsigned int nt_u2 __attribute__((ext_vector_type(2)));
template <typename T>
__device__ inline T stream_load(const T* p) {
#if DEME_FAST_NT
    static_assert(sizeof(T) == 16 || sizeof(T) == 8 || sizeof(T) == 4, "16-, 8- or 4-byte records");
    T out;
    if constexpr (sizeof(T) == 16) {
        const nt_u4 v = __builtin_nontemporal_load(reinterpret_cast<const nt_u4*>(p));
        __builtin_memcpy(&out, &v, 16);
    } else if constexpr (sizeof(T) == 4) {
        const unsigned int v = __builtin_nontemporal_load(reinterpret_cast<const unsigned int*>(p));
        __builtin_memcpy(&out, &v, 4);
    } else {
        const nt_u2 v = __builtin_nontemporal_load(reinterpret_cast<const nt_u2*>(p));
        __builtin_memcpy(&out, &v, 8);
    }
    return out;
#else
    return *p;
#endif
}
template <typename T>
__device__ inline void stream_store(T* p, T v) {
#if DEME_FAST_NT
    if constexpr (sizeof(T) == 16) {
        nt_u4 w;
        __builtin_memcpy(&w, &v, 16);
        __builtin_nontemporal_store(w, reinterpret_cast<nt_u4*>(p));
    } else if constexpr (sizeof(T) == 4) {
        unsigned int w;
        __builtin_memcpy(&w, &v, 4);
        __builtin_nontemporal_store(w, reinterpret_cast<unsigned int*>(p));
    } else {
        nt_u2 w;
        __builtin_memcpy(&w, &v, 8);
        __builtin_nontemporal_store(w, reinterpret_cast<nt_u2*>(p));
    }
#else
    *p = v;
#endif
}
__device__ inline float frcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ inline float fsqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ inline float frsq(float x) { return __builtin_amdgcn_rsqf(x); }

// voxel coordinates of an encoded position (IDChopper, DEMHelperKernels.cuh:92-114) as sub-voxel counts: v * 2^16 + loc
__device__ inline void pos_units(const OwnerRec& r, const DevParams& p, int64_t& ux, int64_t& uy, int64_t& uz) {
    const uint64_t id = r.voxelID;
    ux = (int64_t)(((id & (((uint64_t)1 << p.nvXp2) - 1)) << 16) | r.locX);
    uy = (int64_t)((((id >> p.nvXp2) & (((uint64_t)1 << p.nvYp2) - 1)) << 16) | r.locY);
    uz = (int64_t)(((id >> (p.nvXp2 + p.nvYp2)) << 16) | r.locZ);
}

// Cooperative fetch of the 2 x 64 owner records a wavefront needs (A's and B's owner of each lane's contact): four lanes per
// record, all eight load instructions issued before anything waits, then two transposes through the wavefront's LDS area.
// Every lane of the wavefront must call; lanes without a contact pass owner 0.
__device__ inline void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // LDS operations of one wavefront complete in order: only the
    __builtin_amdgcn_wave_barrier();                         // compiler has to be kept from reordering
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
#ifndef DEME_FAST_STAGE_A
#define DEME_FAST_STAGE_A 0  // A's owners: ~15 distinct records per wavefront (the list is sorted by A), loaded directly
#endif
__device__ inline void stage_owner_records(const OwnerRec* owners, uint32_t ownerA, uint32_t ownerB, uint4* stage, OwnerRec& OA,
                                           OwnerRec& OB) {
    const uint32_t lane = threadIdx.x & 63u, piece = lane & 3u, sub = lane >> 2;
    uint4 vb[4];
#if DEME_FAST_STAGE_A
    uint4 va[4];
#endif
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t ob = (uint32_t)__shfl((int)ownerB, (int)(16 * k + sub));
        vb[k] = reinterpret_cast<const uint4*>(owners + ob)[piece];
#if DEME_FAST_STAGE_A
        const uint32_t oa = (uint32_t)__shfl((int)ownerA, (int)(16 * k + sub));
        va[k] = reinterpret_cast<const uint4*>(owners + oa)[piece];
#endif
    }
    const uint4* mine = stage + lane * DEME_REC_LDS_STRIDE;
    uint4* q;
#if DEME_FAST_STAGE_A
#pragma unroll
    for (int k = 0; k < 4; k++)
        stage[(16 * k + sub) * DEME_REC_LDS_STRIDE + piece] = va[k];
    wave_lds_fence();
    q = reinterpret_cast<uint4*>(&OA);
    q[0] = mine[0], q[1] = mine[1], q[2] = mine[2], q[3] = mine[3];
    wave_lds_fence();
#else
    OA = load_owner(owners, ownerA);
#endif
#pragma unroll
    for (int k = 0; k < 4; k++)
        stage[(16 * k + sub) * DEME_REC_LDS_STRIDE + piece] = vb[k];
    wave_lds_fence();
    q = reinterpret_cast<uint4*>(&OB);
    q[0] = mine[0], q[1] = mine[1], q[2] = mine[2], q[3] = mine[3];
}

// MODEL 0: full Hertzian (history float4: delta_tan x/y/z, delta_time), 1: frictionless.  One thread per contact of the hot
// classes (sphere-sphere, sphere-analytical).  Outputs: the A side's world-frame force and torque (the caller reduces them
// over the owner's run), the B side's record is stored here.
template <int MODEL>
__device__ inline void forces_fast_body(const DevParams& p, const ForceArgs& a, const uint32_t c, const uint4 ci, const OwnerRec& OA,
                                        const OwnerRec& OB, float4& outA4, float2& outA2) {
    const uint32_t cls = ci.x >> 30;
    const float massA = p.massProps[OA.inertiaOff].x;
    float4 hist = make_float4(0, 0, 0, 0);
    float4* wcp = nullptr;
    if (MODEL == 0) {
        wcp = reinterpret_cast<float4*>(a.wc) + c;
        hist = stream_load(wcp);
    }
    const float4 cA = p.comp[ci.z & 0xFFFFu];
    const uint32_t matA = ci.z >> 16;
    // sphere offsets with the reference's own rounding (no contraction, deme_device.h): an offset that differs in its last bit
    // moves the overlap by ~1e-10 m, i.e. 1e-4 of a typical overlap -- everything after this point is well conditioned
    const RotM RA = rot_coeffs(OA.qw, OA.qx, OA.qy, OA.qz);
    const RotM RB = rot_coeffs(OB.qw, OB.qx, OB.qy, OB.qz);
    const f3 relA = rot_apply(RA, mk3(cA.x, cA.y, cA.z));
    const float rA = cA.w;
    float extraMargin = 0.f;
    if (!p.familyTrivial) {
        const float eA = p.familyExtra[OA.family & 0xFFu], eB = p.familyExtra[OB.family & 0xFFu];
        extraMargin = fmaxf(eA, eB);
    }
    // owner-to-owner offset: the one long difference every later vector hangs on, taken exactly in sub-voxel units
    double dOx, dOy, dOz;
    {
        int64_t uAx, uAy, uAz, uBx, uBy, uBz;
        pos_units(OA, p, uAx, uAy, uAz);
        pos_units(OB, p, uBx, uBy, uBz);
        dOx = (double)(uAx - uBx) * p.l, dOy = (double)(uAy - uBy) * p.l, dOz = (double)(uAz - uBz) * p.l;
    }
    const f3 dO = mk3((float)dOx, (float)dOy, (float)dOz);
    f3 n, rAv, rBv;   // B2A, contact point relative to A's / B's centre
    float depth, rB, massB;
    uint32_t matB;
    bool touching;
    if (cls == DEME_KEY_CLASS_SS) {
        const float4 cB = p.comp[ci.w & 0xFFFFu];
        matB = ci.w >> 16;
        rB = cB.w;
        massB = p.massProps[OB.inertiaOff].x;
        const f3 relB = rot_apply(RB, mk3(cB.x, cB.y, cB.z));
        // centre-to-centre vector in fp64, as the reference forms it (DEMHelperKernels.cuh:292-326)
        const double dx = (dOx + (double)relA.x) - (double)relB.x;
        const double dy = (dOy + (double)relA.y) - (double)relB.y;
        const double dz = (dOz + (double)relA.z) - (double)relB.z;
        const double d2 = dx * dx + dy * dy + dz * dz;
        // (the sum of the radii in fp64: rounded to fp32 it is off by up to 3e-8 of itself -- nothing for equal radii, whose sum is
        // exact, but 5e-10 m between unequal ones, 1e-5 of a typical overlap: found on the polydisperse bed of configs[4])
        const double sumRd = (double)rA + (double)rB;
        const float sumR = (float)sumRd;
        const double num = sumRd * sumRd - d2;  // > 0 iff the spheres overlap; no cancellation left after this
        const float d2f = (float)d2;
        const float inv = frsq(d2f);
        const float dist = d2f * inv;
        n = mk3((float)dx * inv, (float)dy * inv, (float)dz * inv);
        depth = (float)num * frcp(sumR + dist);
        touching = !(depth < -extraMargin);
        // contact point = B's centre + (rB - depth / 2) n  (calcContactPoint), relative to the two owners
        const float s = rB - 0.5f * depth;
        rBv = mk3(relB.x + s * n.x, relB.y + s * n.y, relB.z + s * n.z);
        rAv = fsub(rBv, dO);
    } else {  // sphere-analytical (a few per cent of the list, at the walls): the reference's own arithmetic
        const AnalObj ob = p.anal[ci.w];
        matB = ob.mat;
        rB = 1e15f;  // DEME_HUGE_FLOAT
        massB = ob.mass;
        const f3 relB = rot_apply(RB, mk3(ob.relx, ob.rely, ob.relz));
        const f3 dir = rot_apply(RB, mk3(ob.rotx, ob.roty, ob.rotz));
        d3 PA = decode_pos(OA.voxelID, OA.locX, OA.locY, OA.locZ, p), PB = decode_pos(OB.voxelID, OB.locX, OB.locY, OB.locZ, p);
        PA.x += p.LBFX, PA.y += p.LBFY, PA.z += p.LBFZ;
        PB.x += p.LBFX, PB.y += p.LBFY, PB.z += p.LBFZ;
        const d3 bodyA{PA.x + (double)relA.x, PA.y + (double)relA.y, PA.z + (double)relA.z};
        const d3 bodyB{PB.x + (double)relB.x, PB.y + (double)relB.y, PB.z + (double)relB.z};
        d3 cp;
        double dd;
        sphere_entity(bodyA, rA, ob.type, bodyB, dir, ob.size1, ob.normal, 0.0f, cp, n, dd);
        depth = (float)dd;
        touching = !(dd < -(double)extraMargin);  // the list entry keeps its type; only the grace margin retires it
        rAv = mk3((float)(cp.x - PA.x), (float)(cp.y - PA.y), (float)(cp.z - PA.z));
        rBv = mk3((float)(cp.x - PB.x), (float)(cp.y - PB.y), (float)(cp.z - PB.z));
    }
    f3 force = mk3(0, 0, 0), torque_only = mk3(0, 0, 0);
    if (touching) {
        if (depth > 0.f) {
            const MatPair mp = p.matPair[matA * p.nMat + matB];
            const f3 wA = frot_apply(RA, mk3(OA.wx, OA.wy, OA.wz)), wB = frot_apply(RB, mk3(OB.wx, OB.wy, OB.wz));  // world frame
            const f3 rotVelA = fcross(wA, rAv), rotVelB = fcross(wB, rBv);
            const f3 velB2A = fsub(fadd(mk3(OA.vx, OA.vy, OA.vz), rotVelA), fadd(mk3(OB.vx, OB.vy, OB.vz), rotVelB));
            const float projection = fdot(velB2A, n);
            const float mass_eff = massA * massB * frcp(massA + massB);
            const float sqrt_Rd = fsqrt(depth * (rA * rB) * frcp(rA + rB));
            const float Sn = 2.f * mp.E_cnt * sqrt_Rd;
            const float k_n = 0.6666666666666667f * Sn;
            const float gamma_n = 1.825741858350554f * mp.beta * fsqrt(Sn * mass_eff);
            const float Fn = k_n * depth + gamma_n * projection;
            force = fscale(Fn, n);
            if (MODEL == 0) {
                const f3 vrel_tan = faxpy(-projection, n, velB2A);
                f3 delta_tan = faxpy(p.h, vrel_tan, mk3(hist.x, hist.y, hist.z));
                delta_tan = faxpy(-fdot(delta_tan, n), n, delta_tan);
                hist.w += p.h;
                if (mp.Crr > 0.0f) {  // FullHertzianForceModel.cu:73-100
                    bool roll = true;
                    const float R_eff = fsqrt((rA * rB) * frcp(rA + rB));
                    const float kn_simple = 1.3333333333333333f * mp.E_cnt * fsqrt(R_eff);
                    const float gn_simple = -2.f * fsqrt(1.6666666666666667f * mass_eff * mp.E_cnt) * mp.beta * fsqrt(fsqrt(R_eff));
                    const float d_coeff = gn_simple * frcp(2.f * fsqrt(kn_simple * mass_eff));
                    if (d_coeff < 1.0f) {
                        const float t_collision = 3.1415926535897932f * fsqrt(mass_eff * frcp(kn_simple * (1.f - d_coeff * d_coeff)));
                        if (hist.w <= t_collision)
                            roll = false;
                    }
                    if (roll) {
                        const f3 v_rot = fsub(rotVelB, rotVelA);
                        const float m2 = fdot(v_rot, v_rot);
                        if (m2 > 1e-24f)
                            torque_only = fscale(frsq(m2) * mp.Crr * fabsf(Fn), v_rot);
                    }
                }
                if (mp.mu > 0.0f) {
                    const float kt = 8.f * mp.G_cnt * sqrt_Rd;
                    const float gt = -1.825741858350554f * mp.beta * fsqrt(mass_eff * kt);
                    f3 tf = faxpy(-kt, delta_tan, fscale(-gt, vrel_tan));
                    const float ft2 = fdot(tf, tf);
                    if (ft2 > 1e-24f) {
                        const float ft_max = fabsf(Fn) * mp.mu;  // |force| = |Fn| |n|
                        if (ft2 > ft_max * ft_max) {
                            tf = fscale(ft_max * frsq(ft2), tf);
                            delta_tan = fscale(-frcp(kt), faxpy(gt, vrel_tan, tf));
                        }
                    } else {
                        tf = mk3(0, 0, 0);
                    }
                    force = fadd(force, tf);
                }
                hist.x = delta_tan.x, hist.y = delta_tan.y, hist.z = delta_tan.z;
            }
        } else if (MODEL == 0) {
            hist = make_float4(0, 0, 0, 0);  // in the list, within the margin, not touching: the model resets its history
        }
        const f3 tot = fadd(force, torque_only);
        const f3 tA = fcross(rAv, tot);
        f3 tB = fcross(tot, rBv);  // = r_B x (-F)
        outA4 = make_float4(force.x, force.y, force.z, tA.x);
        outA2 = make_float2(tA.y, tA.z);
        if (ghost_of(OA.family) && shared_of(OB.family))  // a ghost sphere on a replicated free body: the sphere's own rank adds it
            force = mk3(0, 0, 0), tB = mk3(0, 0, 0);
        stream_store(a.conB4 + c, make_float4(-force.x, -force.y, -force.z, tB.x));
        stream_store(a.conB2 + c, make_float2(tB.y, tB.z));
    } else {
        outA4 = make_float4(0, 0, 0, 0);
        outA2 = make_float2(0, 0);
        stream_store(a.conB4 + c, make_float4(0, 0, 0, 0));
        stream_store(a.conB2 + c, make_float2(0, 0));
        hist = make_float4(0, 0, 0, 0);  // _forceModelContactWildcardDestroy_
    }
    if (MODEL == 0)
        stream_store(wcp, hist);
}

// Block structure, halo passes and the in-workgroup reduction of the A side are those of calc_forces_block<MODEL, 0>
// (deme_force.h); sphere-mesh contacts still go through the mesh variant of the general kernel, launched first, whose
// A-side records (world-frame too: ForceArgs::world) are folded into the same sums.
#ifndef DEME_FAST_WAVES
#define DEME_FAST_WAVES 1
#endif
template <int MODEL>
__global__ __launch_bounds__(DEME_FORCE_BLOCK, DEME_FAST_WAVES) void k_forces_fast(const DevParams p, const ForceArgs a) {
    __shared__ uint4 sRec[DEME_FORCE_BLOCK / 64][64 * DEME_REC_LDS_STRIDE];
    __shared__ float4 s4[DEME_FORCE_BLOCK];
    __shared__ float2 s2[DEME_FORCE_BLOCK];
    const uint32_t bid = force_block_id(a);
    if (bid * DEME_FORCE_BLOCK >= a.nContacts)
        return;
    if (a.blockMode && !(a.blockMode[bid] & (1u << a.pass)))
        return;
    const uint32_t c = bid * DEME_FORCE_BLOCK + threadIdx.x;
    const bool valid = c < a.nContacts;
    uint4 ci = make_uint4(0, 0, 0, 0);
    bool mine = false, inPass = valid;
    float4 c4 = make_float4(0, 0, 0, 0);
    float2 c2 = make_float2(0, 0);
    uint32_t s = 0, e = 0;
    if (valid) {
        ci = stream_load(a.info + c);
        if (a.cDefer)
            inPass = a.cDefer[c] == a.pass;
        mine = inPass && ((ci.x >> 30) != DEME_KEY_CLASS_SM);
        s = a.aStart[ci.x & 0x3FFFFFFFu];
        e = a.aStart[(ci.x & 0x3FFFFFFFu) + 1];
    }
    OwnerRec OA, OB;
    stage_owner_records(a.owners, mine ? (ci.x & 0x3FFFFFFFu) : 0u, mine ? ci.y : 0u, sRec[threadIdx.x >> 6], OA, OB);
    if (mine)
        forces_fast_body<MODEL>(p, a, c, ci, OA, OB, c4, c2);
    if (inPass && !mine) {  // sphere-mesh contact: evaluated by the mesh variant
        c4 = a.conA4[c];
        c2 = a.conA2[c];
    }
    s4[threadIdx.x] = c4;
    s2[threadIdx.x] = c2;
    __syncthreads();
    if (!inPass)
        return;
    const uint32_t AOwner = ci.x & 0x3FFFFFFFu;
    if (a_run_in_one_block(s, e)) {
        if (c == s) {
            float ax = 0.f, ay = 0.f, az = 0.f, lx = 0.f, ly = 0.f, lz = 0.f;
            for (uint32_t i = s % DEME_FORCE_BLOCK; i <= (e - 1) % DEME_FORCE_BLOCK; i++) {
                const float4 v4 = s4[i];
                const float2 v2 = s2[i];
                ax += v4.x, ay += v4.y, az += v4.z;
                lx += v4.w, ly += v2.x, lz += v2.y;
            }
            a.aSum[2 * (size_t)AOwner] = make_float4(ax, ay, az, 0.f);
            a.aSum[2 * (size_t)AOwner + 1] = make_float4(lx, ly, lz, 0.f);
        }
    } else if (mine) {
        a.conA4[c] = c4;
        a.conA2[c] = c2;
    }
}

}  // namespace deme_dev

#pragma clang fp contract(off)
