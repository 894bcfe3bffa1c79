// deme_force.h -- the contact-force kernel and the built-in force models.
//
// Replaces kernel/DEMCalcForceKernels.cu:44-267 (calculateContactForces), the in-kernel
// reduction of DEMCustomizablePolicies/ForceInKernelReductionStrat.cu and the two built-in
// model fragments (FullHertzianForceModel.cu, FrictionlessHertzianForceModel.cu).
//
// Hook vocabulary kept from the reference: inside the model body the names
// overlapDepth, B2A, contactPnt, AOwnerPos/BOwnerPos, bodyAPos/bodyBPos, AOwnerMass/BOwnerMass,
// ARadius/BRadius, AOriQ/BOriQ, bodyAMatType/bodyBMatType, ContactType, locCPA/locCPB, force,
// torque_only_force, AOwnerFamily/BOwnerFamily, ts, time, ALinVel/BLinVel, ARotVel/BRotVel,
// AOwner/BOwner, AGeo/BGeo, AOwnerMOI/BOwnerMOI and myContactID are in scope with the
// reference's meaning.  Self-contained so the same text compiles under hipRTC.
#pragma once
#include "deme_device.h"
#include "deme_mesh.h"

namespace deme_dev {

struct ForceArgs {
    const OwnerRec* owners;
    const SphereRec* spheres;
    const uint64_t* keys;  // sorted contact keys (sphere ids; only user models read them, as AGeo/BGeo)
    // per-contact gather record written once per detection (k_contact_owners): x = A's owner | class << 30,
    // y = B's owner, z = A's component | material << 16, w = B's component | material << 16 (sphere) or the
    // analytical-object / triangle index.  One coalesced 16-byte load replaces the key load and two dependent
    // SphereRec gathers, so the owner records are the only gathers left on the critical path.
    const uint4* info;
    float* wc;             // contact wildcards, AoS: wc[c*nW + w]
    // per-contact, per-side contributions to the owners' a and alpha (no atomics: the integrator
    // gathers them per owner in contact order).  side A: conA4 = (ax, ay, az, alx), conA2 = (aly, alz)
    // A side is reduced inside the workgroup whenever an owner's whole A run lies in one 256-contact block (see
    // calc_forces_block): then only aSum[owner] = {ax, ay, az, 0 | alx, aly, alz, 0} is written; the per-contact
    // conA records are written only for runs that straddle a block boundary and for sphere-mesh contacts.
    float4* conA4;
    float2* conA2;
    float4* conB4;
    float2* conB2;
    float4* aSum;            // two float4 per owner
    // halo overlap (slab decomposition): cDefer[c] = 1 for the contacts of an owner run that touches a ghost owner; they
    // are evaluated in pass 1, after the ghost records of this step have arrived; blockMode[b] bit p: block b has pass-p
    // work.  Both null: no split (single GPU, or an un-split step).
    const uint8_t* cDefer;
    const uint32_t* blockMode;
    uint32_t pass;
    const uint32_t* smList;  // indices of the sphere-mesh contacts (built per detection): the mesh variant's work list
    uint32_t nSM;
    const uint32_t* aStart;  // nOwners+1: first contact of each owner's A run
    float* recForce;       // optional per-contact records (3 floats each), may be null
    float* recTorque;
    float* recCPA;
    float* recCPB;
    uint32_t nContacts;
    float timeElapsed;
    // XCD-aware launch order: 0 = hardware order (workgroup b on XCD b % 8); G > 0: every XCD works through runs of G
    // consecutive 256-contact blocks, the eight XCDs side by side in the list (see force_block_id)
    uint32_t xcdGroup;
    // 1: the per-side contributions are the WORLD-frame force and torque about the owner's centre (the fast arithmetic mode:
    // the integrator turns the per-owner sums into a and alpha, see acc_from_world); 0: a and alpha contributions as the
    // reference's ForceInKernelReductionStrat.cu forms them per contact (the bit-exact mode)
    uint32_t world;
    // user-model wildcards beyond the per-contact ones (Models.h:319-360): per-owner arrays, and per-geometry arrays for
    // spheres / triangles / analytical components; null when not declared
    float* ownerWc[8];
    float* geoWcSph[8];
    float* geoWcTri[8];
    float* geoWcAnal[8];
};

// Physics-only arithmetic (force model, per-side contributions).  Contact / bin DECISIONS never go through these.
// DEME_FAST_PHYSICS: 1-ulp hardware reciprocal / square root instead of the correctly rounded sequences
// (~10 instructions each); results stay deterministic, within a few fp32 ulp of the exact form.
#ifndef DEME_FAST_PHYSICS
#define DEME_FAST_PHYSICS 0
#endif
__device__ inline float pdiv(float a, float b) {
#if DEME_FAST_PHYSICS
    return a * __builtin_amdgcn_rcpf(b);
#else
    return a / b;
#endif
}
__device__ inline float psqrt(float x) {
#if DEME_FAST_PHYSICS
    return __builtin_amdgcn_sqrtf(x);
#else
    return sqrtf(x);
#endif
}
__device__ inline float plen3(f3 a) { return psqrt(dot3(a, a)); }
__device__ inline f3 pdiv3(f3 a, float s) {
#if DEME_FAST_PHYSICS
    const float r = __builtin_amdgcn_rcpf(s);
    return mk3(a.x * r, a.y * r, a.z * r);
#else
    return a / s;
#endif
}

// B-side contribution record: 24 B as float4 + float2 in two arrays (a single 32-byte record was measured: no gain)
__device__ inline void conb_store(float4* b4, float2* b2, size_t i, float4 c4, float2 c2) {
    b4[i] = c4;
    b2[i] = c2;
}
__device__ inline void conb_load(const float4* b4, const float2* b2, size_t i, float4& c4, float2& c2) {
    c4 = b4[i];
    c2 = b2[i];
}

struct HertzIn {
    double overlapDepth;
    f3 B2A;
    float AOwnerMass, BOwnerMass, ARadius, BRadius;
    RotM RA, RB;  // rotation coefficients of AOriQ / BOriQ (forward)
    f3 locCPA, locCPB;
    f3 ALinVel, BLinVel, ARotVel, BRotVel;
    float ts;
};

// FullHertzianForceModel.cu:5-135.  E_cnt/G_cnt/beta come from the per-material-pair table
// (same formulas evaluated once on the host: matProxy2ContactParam, the log/sqrt of CoR).
__device__ inline void hertz_full(const HertzIn& in, const MatPair& mp, float& dtx, float& dty, float& dtz, float& dtime,
                                  f3& force, f3& torque_only) {
    if (in.overlapDepth > 0) {
        const f3 rotVelCPA = rot_apply(in.RA, cross3(in.ARotVel, in.locCPA));
        const f3 rotVelCPB = rot_apply(in.RB, cross3(in.BRotVel, in.locCPB));
        const f3 velB2A = (in.ALinVel + rotVelCPA) - (in.BLinVel + rotVelCPB);
        const float projection = dot3(velB2A, in.B2A);
        const f3 vrel_tan = velB2A - projection * in.B2A;
        f3 delta_tan = mk3(dtx, dty, dtz);
        delta_tan = delta_tan + in.ts * vrel_tan;
        const float disp_proj = dot3(delta_tan, in.B2A);
        delta_tan = delta_tan - disp_proj * in.B2A;
        dtime += in.ts;

        const float mass_eff = pdiv(in.AOwnerMass * in.BOwnerMass, in.AOwnerMass + in.BOwnerMass);
        const float sqrt_Rd =
            (float)sqrt(in.overlapDepth * (double)(in.ARadius * in.BRadius) / (double)(in.ARadius + in.BRadius));
        const float Sn = (float)(2. * mp.E_cnt * sqrt_Rd);
        const float beta = mp.beta;
        const float k_n = (float)((2. / 3.) * Sn);
        const float gamma_n = (float)(1.825741858350554 * beta * psqrt(Sn * mass_eff));
        force = force + (float)(k_n * in.overlapDepth + gamma_n * projection) * in.B2A;

        if (mp.Crr > 0.0f) {
            bool roll = true;
            const float R_eff = sqrtf((in.ARadius * in.BRadius) / (in.ARadius + in.BRadius));
            const float kn_simple = (float)((4. / 3.) * mp.E_cnt * sqrtf(R_eff));
            const float gn_simple = -2.f * sqrtf((float)((5. / 3.) * mass_eff * mp.E_cnt)) * beta * powf(R_eff, 0.25f);
            const float d_coeff = gn_simple / (2.f * sqrtf(kn_simple * mass_eff));
            if (d_coeff < 1.0) {
                const float t_collision =
                    (float)(3.1415926535897932385 * sqrtf(mass_eff / (kn_simple * (1.f - d_coeff * d_coeff))));
                if (dtime <= t_collision)
                    roll = false;
            }
            if (roll) {
                const f3 v_rot = rotVelCPB - rotVelCPA;
                const float v_rot_mag = plen3(v_rot);
                if (v_rot_mag > 1e-12)
                    torque_only = pdiv3(v_rot, v_rot_mag) * (mp.Crr * plen3(force));
            }
        }
        if (mp.mu > 0.0f) {
            const float kt = (float)(8. * mp.G_cnt * sqrt_Rd);
            const float gt = (float)(-1.825741858350554 * beta * psqrt(mass_eff * kt));
            f3 tangent_force = (-kt) * delta_tan - gt * vrel_tan;
            const float ft = plen3(tangent_force);
            if (ft > 1e-12) {
                const float ft_max = plen3(force) * mp.mu;
                if (ft > ft_max) {
                    tangent_force = pdiv(ft_max, ft) * tangent_force;
                    delta_tan = pdiv3(tangent_force + gt * vrel_tan, -kt);
                }
            } else {
                tangent_force = mk3(0, 0, 0);
            }
            force = force + tangent_force;
        }
        dtx = delta_tan.x;
        dty = delta_tan.y;
        dtz = delta_tan.z;
    } else {
        dtime = 0;
        dtx = 0;
        dty = 0;
        dtz = 0;
    }
}

// FrictionlessHertzianForceModel.cu:3-42
__device__ inline void hertz_frictionless(const HertzIn& in, const MatPair& mp, f3& force) {
    if (in.overlapDepth > 0) {
        const f3 rotVelCPA = rot_apply(in.RA, cross3(in.ARotVel, in.locCPA));
        const f3 rotVelCPB = rot_apply(in.RB, cross3(in.BRotVel, in.locCPB));
        const f3 velB2A = (in.ALinVel + rotVelCPA) - (in.BLinVel + rotVelCPB);
        const float projection = dot3(velB2A, in.B2A);
        const float mass_eff = (in.AOwnerMass * in.BOwnerMass) / (in.AOwnerMass + in.BOwnerMass);
        const float sqrt_Rd =
            (float)sqrt(in.overlapDepth * (double)(in.ARadius * in.BRadius) / (double)(in.ARadius + in.BRadius));
        const float Sn = (float)(2. * mp.E_cnt * sqrt_Rd);
        const float k_n = (float)((2. / 3.) * Sn);
        const float gamma_n = (float)(1.825741858350554 * mp.beta * sqrtf(Sn * mass_eff));
        force = force + (float)(k_n * in.overlapDepth + gamma_n * projection) * in.B2A;
    }
}

// One side's contribution to its owner's a and alpha: ForceInKernelReductionStrat.cu:2-33 (same arithmetic
// as DEMCollectForceKernels_Compact.cu:13-101), evaluated per contact and stored instead of atomically added.
__device__ inline void side_contribution(f3 F, f3 Ftot, float mass, f3 moi, const RotM& Rinv, f3 locCP, float4& c4,
                                         float2& c2) {
    const f3 myF = rot_apply(Rinv, Ftot);
    const f3 cr = cross3(locCP, myF);
#if DEME_FAST_PHYSICS
    const float rm = __builtin_amdgcn_rcpf(mass);
    c4 = make_float4(F.x * rm, F.y * rm, F.z * rm, cr.x * __builtin_amdgcn_rcpf(moi.x));
    c2 = make_float2(cr.y * __builtin_amdgcn_rcpf(moi.y), cr.z * __builtin_amdgcn_rcpf(moi.z));
#else
    c4 = make_float4(F.x / mass, F.y / mass, F.z / mass, cr.x / moi.x);
    c2 = make_float2(cr.y / moi.y, cr.z / moi.z);
#endif
}

#ifdef DEME_JIT
// Run-time compiled user model (MODEL 2).  The kernel hands every ingredient of the reference's hook
// vocabulary (DEMCalcForceKernels.cu:48-61,236-248; Models.h:219-312) to deme_user_model(), whose body is
// generated around the user's statement block by deme_jit.h.
struct UserModelIO {
    double overlapDepth;
    float3 B2A;
    double3 contactPnt, AOwnerPos, BOwnerPos, bodyAPos, bodyBPos;
    float AOwnerMass, BOwnerMass, ARadius, BRadius;
    float4 AOriQ, BOriQ;
    uint16_t bodyAMatType, bodyBMatType;
    uint8_t ContactType, AOwnerFamily, BOwnerFamily;
    float3 locCPA, locCPB, force, torque_only_force;
    float ts, time;
    float3 ALinVel, BLinVel, ARotVel, BRotVel, AOwnerMOI, BOwnerMOI;
    uint32_t AOwner, BOwner, AGeo, BGeo, myContactID;
    float* wc;  // this contact's wildcard slots
    float* const* ownerWc;  // owner wildcards: arrays indexed by owner id (aliases, as in the reference)
    float* const* geoWcA;   // geometry wildcards of A's kind (always spheres)
    float* const* geoWcB;   // geometry wildcards of B's kind (spheres, triangles or analytical components)
};
__device__ void deme_user_model(UserModelIO& io);
#endif

// One thread per contact.  MODEL: 0 full Hertzian (4 wildcards), 1 frictionless (none), 2 user model (JIT).
// CLS selects the geometry branches that are compiled in: 0 = sphere-sphere and sphere-analytical (the hot
// variant, over the whole list), 1 = sphere-mesh only (over the per-detection list of sphere-mesh contacts, launched
// first and only when triangles are loaded).  Keeping the fp64 triangle code out of the hot
// variant saves ~40 VGPRs (3 -> 4 waves per SIMD; measured 174 -> 148 us at 4.2 M contacts).  Measured and
// rejected: a third variant for analytical contacts (no occupancy gain, one more pass), per-class index lists
// (their same-address atomics cost 1.7 ms per detection), register caps of 96 / 80 VGPRs (spills: 231 / 385 us).
template <int MODEL, int CLS>
__device__ inline void calc_forces_body(const DevParams& p, const ForceArgs& a, const uint32_t myContactID, const uint4 ci,
                                        float4& outA4, float2& outA2) {
    const uint32_t cls = ci.x >> 30;
    uint32_t ContactType = (cls == DEME_KEY_CLASS_SS) ? 1u : 0u;

    HertzIn in;
    in.ts = p.h;
    d3 contactPnt{0, 0, 0}, AOwnerPos, BOwnerPos, bodyAPos, bodyBPos;
    // ---- A: always a sphere (DEMCalcForceKernels.cu:63-94)
    SphereRec sA;
    sA.owner = ci.x & 0x3FFFFFFFu, sA.comp = (uint16_t)(ci.z & 0xFFFFu), sA.mat = (uint16_t)(ci.z >> 16);
    const uint32_t AOwner = sA.owner;
    const OwnerRec oA = load_owner(a.owners, AOwner);
    const float4 mpA = p.massProps[oA.inertiaOff];
    in.AOwnerMass = mpA.x;
    in.ALinVel = mk3(oA.vx, oA.vy, oA.vz);
    in.ARotVel = mk3(oA.wx, oA.wy, oA.wz);
    AOwnerPos = decode_pos(oA.voxelID, oA.locX, oA.locY, oA.locZ, p);
    AOwnerPos.x += p.LBFX;
    AOwnerPos.y += p.LBFY;
    AOwnerPos.z += p.LBFZ;
    in.RA = rot_coeffs(oA.qw, oA.qx, oA.qy, oA.qz);
    {
        const float4 c = p.comp[sA.comp];
        const f3 rel = rot_apply(in.RA, mk3(c.x, c.y, c.z));
        bodyAPos = {AOwnerPos.x + (double)rel.x, AOwnerPos.y + (double)rel.y, AOwnerPos.z + (double)rel.z};
        in.ARadius = c.w;
    }
    const uint32_t AOwnerFamily = fam_of(oA.family);
    float extraMarginSize = p.familyTrivial ? 0.f : p.familyExtra[AOwnerFamily];
    const uint32_t bodyAMatType = sA.mat;
    uint32_t bodyBMatType = 0;
    const uint32_t BOwner = ci.y;
    const uint32_t BIdx = ci.w;  // analytical-object or triangle index for the non-sphere classes
    float4 mpB;
    OwnerRec oB;
    if (CLS == 0 && cls == DEME_KEY_CLASS_SS) {  // DEMCalcForceKernels.cu:97-136
        SphereRec sB;
        sB.owner = BOwner, sB.comp = (uint16_t)(ci.w & 0xFFFFu), sB.mat = (uint16_t)(ci.w >> 16);
        oB = load_owner(a.owners, BOwner);
        mpB = p.massProps[oB.inertiaOff];
        in.BOwnerMass = mpB.x;
        BOwnerPos = decode_pos(oB.voxelID, oB.locX, oB.locY, oB.locZ, p);
        BOwnerPos.x += p.LBFX;
        BOwnerPos.y += p.LBFY;
        BOwnerPos.z += p.LBFZ;
        in.RB = rot_coeffs(oB.qw, oB.qx, oB.qy, oB.qz);
        const float4 c = p.comp[sB.comp];
        const f3 rel = rot_apply(in.RB, mk3(c.x, c.y, c.z));
        bodyBPos = {BOwnerPos.x + (double)rel.x, BOwnerPos.y + (double)rel.y, BOwnerPos.z + (double)rel.z};
        in.BRadius = c.w;
        bodyBMatType = sB.mat;
        if (!p.familyTrivial) {
            const float eB = p.familyExtra[fam_of(oB.family)];
            extraMarginSize = (extraMarginSize > eB) ? extraMarginSize : eB;
        }
        spheres_overlap(bodyAPos.x, bodyAPos.y, bodyAPos.z, (double)in.ARadius, bodyBPos.x, bodyBPos.y, bodyBPos.z,
                        (double)in.BRadius, contactPnt, in.B2A, in.overlapDepth);
        if (in.overlapDepth < -extraMarginSize)
            ContactType = 0u;
    } else if (CLS == 0) {  // DEMCalcForceKernels.cu:184-232
        const AnalObj ob = p.anal[BIdx];
        oB = load_owner(a.owners, BOwner);
        mpB = p.massProps[oB.inertiaOff];
        bodyBMatType = ob.mat;
        in.BOwnerMass = ob.mass;
        in.BRadius = 1e15f;  // DEME_HUGE_FLOAT
        BOwnerPos = decode_pos(oB.voxelID, oB.locX, oB.locY, oB.locZ, p);
        BOwnerPos.x += p.LBFX;
        BOwnerPos.y += p.LBFY;
        BOwnerPos.z += p.LBFZ;
        in.RB = rot_coeffs(oB.qw, oB.qx, oB.qy, oB.qz);
        const f3 rel = rot_apply(in.RB, mk3(ob.relx, ob.rely, ob.relz));
        bodyBPos = {BOwnerPos.x + (double)rel.x, BOwnerPos.y + (double)rel.y, BOwnerPos.z + (double)rel.z};
        if (!p.familyTrivial) {
            const float eB = p.familyExtra[fam_of(oB.family)];
            extraMarginSize = (extraMarginSize > eB) ? extraMarginSize : eB;
        }
        const f3 rot = rot_apply(in.RB, mk3(ob.rotx, ob.roty, ob.rotz));
        ContactType = sphere_entity(bodyAPos, in.ARadius, ob.type, bodyBPos, rot, ob.size1, ob.normal, 0.0f, contactPnt,
                                    in.B2A, in.overlapDepth);
        // the list entry keeps its type even when the exact test says "not touching"; only the
        // grace-margin test below can retire it (DEMCalcForceKernels.cu:228-231)
        ContactType = (ob.type == 0) ? 11u : 13u;
        if (in.overlapDepth < -extraMarginSize)
            ContactType = 0u;
    } else {  // sphere-mesh, DEMCalcForceKernels.cu:138-183: exact test in fp64 against the ORIGINAL triangle
        const TriRec tr = reinterpret_cast<const TriRec*>(p.tris)[BIdx];
        oB = load_owner(a.owners, BOwner);
        mpB = p.massProps[oB.inertiaOff];
        in.BOwnerMass = mpB.x;
        in.BRadius = 1e15f;  // DEME_HUGE_FLOAT
        bodyBMatType = tr.mat;
        if (!p.familyTrivial) {
            const float eB = p.familyExtra[fam_of(oB.family)];
            extraMarginSize = (extraMarginSize > eB) ? extraMarginSize : eB;
        }
        BOwnerPos = decode_pos(oB.voxelID, oB.locX, oB.locY, oB.locZ, p);
        BOwnerPos.x += p.LBFX;
        BOwnerPos.y += p.LBFY;
        BOwnerPos.z += p.LBFZ;
        in.RB = rot_coeffs(oB.qw, oB.qx, oB.qy, oB.qz);
        v3<double> nd[3];
        const float* src[3] = {tr.n1, tr.n2, tr.n3};
        for (int k = 0; k < 3; k++) {
            const d3 r = rot_apply_d(in.RB, d3{(double)src[k][0], (double)src[k][1], (double)src[k][2]});
            nd[k] = {BOwnerPos.x + r.x, BOwnerPos.y + r.y, BOwnerPos.z + r.z};
        }
        bodyBPos = {(nd[0].x + nd[1].x + nd[2].x) / 3., (nd[0].y + nd[1].y + nd[2].y) / 3., (nd[0].z + nd[1].z + nd[2].z) / 3.};
        v3<double> cn, cpt;
        double depth;
        const bool in_contact = tri_sphere_cd<double, false>(nd[0], nd[1], nd[2], v3<double>{bodyAPos.x, bodyAPos.y, bodyAPos.z},
                                                             (double)in.ARadius, cn, depth, cpt);
        in.B2A = mk3((float)cn.x, (float)cn.y, (float)cn.z);
        contactPnt = {cpt.x, cpt.y, cpt.z};
        ContactType = 2u;
        // the extra margin only counts on the positive side of the facet (DEMCalcForceKernels.cu:176-181)
        if ((depth > extraMarginSize) || (!in_contact && depth < 0.))
            ContactType = 0u;
        in.overlapDepth = -depth;
    }
    in.BLinVel = mk3(oB.vx, oB.vy, oB.vz);
    in.BRotVel = mk3(oB.wx, oB.wy, oB.wz);

    // contact wildcards (contact history): _forceModelContactWildcardAcq_
    float4 hist = make_float4(0, 0, 0, 0);
    float4* wcp = nullptr;
    if (MODEL == 0) {
        wcp = reinterpret_cast<float4*>(a.wc) + myContactID;
        hist = *wcp;  // delta_tan_x, delta_tan_y, delta_tan_z, delta_time (std::set order, Models.h:363-378)
    }
    if (ContactType != 0u) {
        f3 force = mk3(0, 0, 0), torque_only_force = mk3(0, 0, 0);
        // rotation by the conjugate quaternion: its nine coefficients are, bit for bit, the transposed forward ones
        // ((-x)(-y) = xy, w(-z) = -(wz), a - (-b) = a + b are all exact), so they are not recomputed
        const RotM RAinv = rot_transpose(in.RA);
        const RotM RBinv = rot_transpose(in.RB);
        in.locCPA = rot_apply(RAinv, mk3((float)(contactPnt.x - AOwnerPos.x), (float)(contactPnt.y - AOwnerPos.y),
                                         (float)(contactPnt.z - AOwnerPos.z)));
        in.locCPB = rot_apply(RBinv, mk3((float)(contactPnt.x - BOwnerPos.x), (float)(contactPnt.y - BOwnerPos.y),
                                         (float)(contactPnt.z - BOwnerPos.z)));
        const MatPair mp = p.matPair[bodyAMatType * p.nMat + bodyBMatType];
        if (MODEL == 0)
            hertz_full(in, mp, hist.x, hist.y, hist.z, hist.w, force, torque_only_force);
        else if (MODEL == 1)
            hertz_frictionless(in, mp, force);
#ifdef DEME_JIT
        else {
            UserModelIO io;
            io.overlapDepth = in.overlapDepth;
            io.B2A = make_float3(in.B2A.x, in.B2A.y, in.B2A.z);
            io.contactPnt = make_double3(contactPnt.x, contactPnt.y, contactPnt.z);
            io.AOwnerPos = make_double3(AOwnerPos.x, AOwnerPos.y, AOwnerPos.z);
            io.BOwnerPos = make_double3(BOwnerPos.x, BOwnerPos.y, BOwnerPos.z);
            io.bodyAPos = make_double3(bodyAPos.x, bodyAPos.y, bodyAPos.z);
            io.bodyBPos = make_double3(bodyBPos.x, bodyBPos.y, bodyBPos.z);
            io.AOwnerMass = in.AOwnerMass, io.BOwnerMass = in.BOwnerMass, io.ARadius = in.ARadius, io.BRadius = in.BRadius;
            io.AOriQ = make_float4(oA.qx, oA.qy, oA.qz, oA.qw);  // float4 is (x, y, z, w)
            io.BOriQ = make_float4(oB.qx, oB.qy, oB.qz, oB.qw);
            io.bodyAMatType = (uint16_t)bodyAMatType, io.bodyBMatType = (uint16_t)bodyBMatType;
            io.ContactType = (uint8_t)ContactType, io.AOwnerFamily = (uint8_t)AOwnerFamily, io.BOwnerFamily = (uint8_t)fam_of(oB.family);
            io.locCPA = make_float3(in.locCPA.x, in.locCPA.y, in.locCPA.z);
            io.locCPB = make_float3(in.locCPB.x, in.locCPB.y, in.locCPB.z);
            io.force = make_float3(0, 0, 0), io.torque_only_force = make_float3(0, 0, 0);
            io.ts = p.h, io.time = a.timeElapsed;
            io.ALinVel = make_float3(in.ALinVel.x, in.ALinVel.y, in.ALinVel.z);
            io.BLinVel = make_float3(in.BLinVel.x, in.BLinVel.y, in.BLinVel.z);
            io.ARotVel = make_float3(in.ARotVel.x, in.ARotVel.y, in.ARotVel.z);
            io.BRotVel = make_float3(in.BRotVel.x, in.BRotVel.y, in.BRotVel.z);
            io.AOwnerMOI = make_float3(mpA.y, mpA.z, mpA.w), io.BOwnerMOI = make_float3(mpB.y, mpB.z, mpB.w);
            // ids in the user vocabulary are the CALLER's (the engine may keep owners and spheres in an order of its own:
            // deme_order.inc); the user's owner / sphere wildcard arrays are indexed by them and stay in the caller's order
            io.AOwner = p.o2e ? p.o2e[AOwner] : AOwner, io.BOwner = p.o2e ? p.o2e[BOwner] : BOwner, io.myContactID = myContactID;
            {
                const uint64_t key = a.keys[myContactID];
                const uint32_t ga = key_a(key), gb = key_b(key);
                io.AGeo = p.s2e ? p.s2e[ga] : ga, io.BGeo = (p.s2e && cls == DEME_KEY_CLASS_SS) ? p.s2e[gb] : gb;
            }
            io.wc = a.wc + (size_t)myContactID * p.nW;
            io.ownerWc = a.ownerWc;
            io.geoWcA = a.geoWcSph;
            io.geoWcB = (cls == DEME_KEY_CLASS_SS) ? a.geoWcSph : (cls == DEME_KEY_CLASS_SM) ? a.geoWcTri : a.geoWcAnal;
            deme_user_model(io);
            force = mk3(io.force.x, io.force.y, io.force.z);
            torque_only_force = mk3(io.torque_only_force.x, io.torque_only_force.y, io.torque_only_force.z);
        }
#endif
        if (a.recForce) {  // _contactInfoWrite_ (ContactInfoWriteBack.cu)
            float* r = a.recForce + 3ull * myContactID;
            r[0] = force.x, r[1] = force.y, r[2] = force.z;
            r = a.recTorque + 3ull * myContactID;
            r[0] = torque_only_force.x, r[1] = torque_only_force.y, r[2] = torque_only_force.z;
            r = a.recCPA + 3ull * myContactID;
            r[0] = in.locCPA.x, r[1] = in.locCPA.y, r[2] = in.locCPA.z;
            r = a.recCPB + 3ull * myContactID;
            r[0] = in.locCPB.x, r[1] = in.locCPB.y, r[2] = in.locCPB.z;
        }
        // _forceCollectInPlaceStrat_ -> per-side contributions
        f3 tot = force + torque_only_force;
        float4 c4;
        float2 c2;
        f3 nF = mk3(-force.x, -force.y, -force.z);
        if (ghost_of(oA.family) && shared_of(oB.family))  // a ghost sphere on a replicated free body: the sphere's own rank adds
            nF = mk3(0, 0, 0), tot = mk3(0, 0, 0);         // this contact to the body's sum (A's side is a ghost's: never used)
        if (a.world) {  // world-frame force and torque (R locCP) x F_tot per side
            const f3 tA = cross3(rot_apply(in.RA, in.locCPA), tot);
            outA4 = make_float4(force.x, force.y, force.z, tA.x);
            outA2 = make_float2(tA.y, tA.z);
            const f3 tB = cross3(tot, rot_apply(in.RB, in.locCPB));
            conb_store(a.conB4, a.conB2, myContactID, make_float4(nF.x, nF.y, nF.z, tB.x), make_float2(tB.y, tB.z));
        } else {
            side_contribution(force, tot, in.AOwnerMass, mk3(mpA.y, mpA.z, mpA.w), RAinv, in.locCPA, c4, c2);
            outA4 = c4;
            outA2 = c2;
            side_contribution(nF, -1.f * tot, in.BOwnerMass, mk3(mpB.y, mpB.z, mpB.w), RBinv, in.locCPB, c4, c2);
            conb_store(a.conB4, a.conB2, myContactID, c4, c2);
        }
    } else {
        outA4 = make_float4(0, 0, 0, 0);
        outA2 = make_float2(0, 0);
        conb_store(a.conB4, a.conB2, myContactID, make_float4(0, 0, 0, 0), make_float2(0, 0));
        hist = make_float4(0, 0, 0, 0);  // _forceModelContactWildcardDestroy_
        if (MODEL == 2)
            for (uint32_t w = 0; w < p.nW; w++)
                a.wc[(size_t)myContactID * p.nW + w] = 0.f;
        if (a.recForce) {
            for (int k = 0; k < 3; k++) {
                a.recForce[3ull * myContactID + k] = 0.f;
                a.recTorque[3ull * myContactID + k] = 0.f;
                a.recCPA[3ull * myContactID + k] = 0.f;
                a.recCPB[3ull * myContactID + k] = 0.f;
            }
        }
    }
    if (MODEL == 0)
        *wcp = hist;  // _forceModelContactWildcardWrite_ (skipping unchanged all-zero histories was measured: slower)
}

// an owner's A run [s, e) is reduced in-workgroup iff it lies inside one block of DEME_FORCE_BLOCK contacts
#ifndef DEME_FORCE_BLOCK
#define DEME_FORCE_BLOCK 256
#endif
__host__ __device__ inline bool a_run_in_one_block(uint32_t s, uint32_t e) {
    return s < e && (s / DEME_FORCE_BLOCK) == ((e - 1) / DEME_FORCE_BLOCK);
}

// Workgroup = DEME_FORCE_BLOCK consecutive contacts.  The list is sorted by A's owner, so an owner's A-side
// contributions sit in consecutive lanes: they go to LDS and the lane of the run's first contact sums them in
// list order (the order the integrator and the oracle use) and writes ONE 32-byte record per owner instead of
// 24 bytes per contact (-0.2 GB of HBM traffic per step at 4.3 M contacts, half here, half in the integrator).
// Runs that straddle a block boundary (~1 owner in 60) keep the per-contact records.  The mesh-only variant
// (CLS 1) is launched BEFORE the hot variant and leaves its A-side records in conA; the hot variant folds them
// into the same in-order sum.
// Which 256-contact block a workgroup takes.  The hardware deals consecutive workgroups round-robin to the 8 XCDs, each with
// its own L2: with the identity map the blocks that share B-owner records (list neighbours: the same and the adjacent lattice
// rows) sit on eight different L2s and every XCD fetches its own copy.  With G > 0, workgroup b (XCD b % 8, the (b / 8)-th on
// it) takes block ((j / G) * 8 + xcd) * G + j % G: each XCD sweeps G consecutive blocks, the eight runs adjacent in the list,
// so neighbours meet in one L2 while all XCDs still work on the same region of the bed.  The grid is rounded up to a multiple
// of 8 G by the host; workgroups mapped beyond the list exit.
__device__ inline uint32_t force_block_id(const ForceArgs& a) {
    const uint32_t b = blockIdx.x, G = a.xcdGroup;
    if (G == 0)
        return b;
    const uint32_t xcd = b & 7u, j = b >> 3;
    return ((j / G) * 8u + xcd) * G + (j % G);
}

template <int MODEL, int CLS>
__device__ inline void calc_forces_block(const DevParams& p, const ForceArgs& a) {
    const uint32_t bid = (CLS == 0) ? force_block_id(a) : blockIdx.x;
    if (CLS == 0 && bid * DEME_FORCE_BLOCK >= a.nContacts)
        return;
    if (CLS == 0 && a.blockMode && !(a.blockMode[bid] & (1u << a.pass)))
        return;  // nothing of this pass in the block (workgroup-uniform)
    uint32_t c = bid * DEME_FORCE_BLOCK + threadIdx.x;
    if (CLS == 1) {  // mesh variant: one thread per sphere-mesh contact, through the per-detection index list
        if (c >= a.nSM)
            return;
        c = a.smList[c];
    }
    const bool valid = c < a.nContacts;
    uint4 ci = make_uint4(0, 0, 0, 0);
    bool mine = false;
    float4 c4 = make_float4(0, 0, 0, 0);
    float2 c2 = make_float2(0, 0);
    uint32_t s = 0, e = 0;
    bool inPass = valid;
    if (valid) {
        ci = a.info[c];
        if (CLS == 0 && a.cDefer)
            inPass = a.cDefer[c] == a.pass;
        mine = inPass && ((CLS == 1) == ((ci.x >> 30) == DEME_KEY_CLASS_SM));
        if (CLS == 0) {  // issued before the force evaluation so that their latency is hidden behind it
            s = a.aStart[ci.x & 0x3FFFFFFFu];
            e = a.aStart[(ci.x & 0x3FFFFFFFu) + 1];
        }
        if (mine)
            calc_forces_body<MODEL, CLS>(p, a, c, ci, c4, c2);
    }
    if (CLS == 1) {
        if (mine) {
            a.conA4[c] = c4;
            a.conA2[c] = c2;
        }
        return;
    }
    __shared__ float4 s4[DEME_FORCE_BLOCK];
    __shared__ float2 s2[DEME_FORCE_BLOCK];
    if (inPass && !mine) {  // sphere-mesh contact: evaluated by the mesh variant
        c4 = a.conA4[c];
        c2 = a.conA2[c];
    }
    s4[threadIdx.x] = c4;
    s2[threadIdx.x] = c2;
    __syncthreads();
    if (!inPass)
        return;  // beyond the list, or a contact of the other pass (whole owner runs belong to one pass)
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

template <int MODEL, int CLS>
__global__ __launch_bounds__(DEME_FORCE_BLOCK) void k_calc_forces(const DevParams p, const ForceArgs a) {
    calc_forces_block<MODEL, CLS>(p, a);
}
}  // namespace deme_dev
