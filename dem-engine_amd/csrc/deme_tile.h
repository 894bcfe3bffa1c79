// deme_tile.h -- the OWNER-TILE form of the contact-force pass (fast arithmetic mode, built-in models, no mesh contacts).
//
// Same physics as deme_force_fast.h (kernel/DEMCalcForceKernels.cu:44-267 with FullHertzianForceModel.cu /
// FrictionlessHertzianForceModel.cu; accumulation semantics of DEMCustomizablePolicies/ForceInKernelReductionStrat.cu:2-33),
// laid out around LDS-staged particle tiles:
//   * a workgroup owns a TILE of DEME_TILE_NB consecutive owners -- with the clumps numbered along a Z-order curve that is a
//     compact cluster of the bed -- and every contact whose A owner lies in the tile (one contiguous range of the list);
//   * the tile's owner records and those of the foreign owners its contacts touch (its HALO: on average 0.7 records per tile
//     owner at 1e6 packed clumps) are fetched ONCE per step, converted ONCE (position relative to the tile's origin as the exact
//     integer difference of the encoded positions times l, world-frame angular velocity, mass) and staged in LDS; a contact then
//     reads its two owners from LDS through two 10-bit slot numbers -- the per-contact gather record shrinks from 16 to 8 bytes
//     and the force pass issues no scattered global loads at all;
//   * both sides' contributions are summed inside the wavefront -- the 64 contacts of a chunk are sorted by A's slot as the
//     list comes, and by B's slot through a 6-bit rank the builder leaves in the gather record (one scatter through LDS), so
//     either side is a sequence of runs: a segmented DPP scan (row_shr 1, 2, 4, 8, row_bcast 15 / 31) leaves each run's sum in
//     its last lane, which adds it to the wavefront's own copy of the tile's sums with a plain LDS read-modify-write (no two
//     lanes of one instruction share a slot; LDS float atomics were measured at ~2 cycles PER LANE on gfx950: 12 of them per
//     contact made the first version of this kernel 2.6 times slower than k_forces_fast) -- fixed order, reproducible run to run.
//     The four copies leave the workgroup as ONE 32-byte sum per owner; only a contact whose B owner lives in another tile
//     writes a 24-byte record, which the integrator gathers through the per-owner list of such contacts (deme_kernels.h:
//     k_integrate<true> with GatherArgs::tile).
// What the per-detection builder below leaves behind: tInfo (8 B per contact), the halo list of every tile, the per-owner lists
// of tile-crossing contacts.  A tile whose halo does not fit the LDS area switches the whole context back to k_forces_fast for
// that list (RangeCounters::tileOverflow).
#pragma once
#include "deme_force_fast.h"

#ifndef DEME_TILE_NB
#define DEME_TILE_NB 128  // owners per tile
#endif
#ifndef DEME_TILE_HMAX
#define DEME_TILE_HMAX 192  // foreign owners a tile can stage (measured at 1e6 packed clumps: mean 93, largest 147)
#endif
#define DEME_TILE_T 256
#define DEME_TILE_HP2 256  // DEME_TILE_HMAX rounded up to a power of two (bitonic sort of the halo list)
#define DEME_TILE_HASH 1024u
#define DEME_TILE_REC 5  // uint4 per staged owner (80 bytes)
static_assert(DEME_TILE_NB + DEME_TILE_HMAX <= 1024, "slot numbers are 10 bits");
static_assert(DEME_TILE_HMAX <= DEME_TILE_HP2, "halo list sort size");

#pragma clang fp contract(fast)

namespace deme_dev {

// tInfo.x: slot of A (10) | slot of B (10) | class (2) | B lives in another tile: write a record (1) | 1 spare | material of A (4) |
//          material of B (4);  tInfo.y: component of A (13) | component of B or analytical-object index (13) | rank of the
//          contact among the 64 of its chunk when they are ordered by B's slot, B's in other tiles last (6)
__host__ __device__ inline uint32_t tile_info_x(uint32_t slotA, uint32_t slotB, uint32_t cls, uint32_t rec, uint32_t matA, uint32_t matB) {
    return slotA | (slotB << 10) | (cls << 20) | (rec << 22) | (matA << 24) | (matB << 28);
}

struct TileArgs {
    const OwnerRec* owners;
    const uint2* tInfo;
    const uint32_t* aStart;
    const uint32_t* hList;     // DEME_TILE_HMAX entries per tile
    const uint32_t* hCount;
    const uint32_t* tileMode;  // halo overlap: bit p = the tile is evaluated in pass p (1: reads no ghost owner, 2: does); null: no split
    float* wc;
    float4* tSum;              // two float4 per owner: the sum of the contributions of contacts evaluated by the owner's tile
    float4* conB4;             // per-contact records of the B sides that live in another tile
    float2* conB2;
    uint32_t nOwners, nTiles, pass, xcdGroup;
};

// one staged owner, as the contact loop reads it back from LDS
struct TileOwner {
    double px, py, pz;  // position relative to the tile's origin [m]
    float mass;
    uint32_t family;
    float qw, qx, qy, qz;
    float vx, vy, vz;
    float wx, wy, wz;  // angular velocity in the WORLD frame
};

__device__ inline void tile_stage_owner(const DevParams& p, const OwnerRec& r, int64_t u0x, int64_t u0y, int64_t u0z, uint4* dst) {
    int64_t ux, uy, uz;
    pos_units(r, p, ux, uy, uz);
    const double px = (double)(ux - u0x) * p.l, py = (double)(uy - u0y) * p.l, pz = (double)(uz - u0z) * p.l;
    const RotM R = rot_coeffs(r.qw, r.qx, r.qy, r.qz);
    const f3 w = frot_apply(R, mk3(r.wx, r.wy, r.wz));
    const float mass = p.massProps[r.inertiaOff].x;
    uint2 bx, by, bz;
    __builtin_memcpy(&bx, &px, 8), __builtin_memcpy(&by, &py, 8), __builtin_memcpy(&bz, &pz, 8);
    dst[0] = make_uint4(bx.x, bx.y, by.x, by.y);
    dst[1] = make_uint4(bz.x, bz.y, __float_as_uint(mass), r.family);
    dst[2] = make_uint4(__float_as_uint(r.qw), __float_as_uint(r.qx), __float_as_uint(r.qy), __float_as_uint(r.qz));
    dst[3] = make_uint4(__float_as_uint(r.vx), __float_as_uint(r.vy), __float_as_uint(r.vz), __float_as_uint(w.x));
    dst[4] = make_uint4(__float_as_uint(w.y), __float_as_uint(w.z), 0u, 0u);
}
__device__ inline TileOwner tile_read_owner(const uint4* sOwn, uint32_t slot) {
    const uint4* q = sOwn + slot * DEME_TILE_REC;
    const uint4 a = q[0], b = q[1], c = q[2], d = q[3], e = q[4];
    TileOwner o;
    uint2 t;
    t = make_uint2(a.x, a.y), __builtin_memcpy(&o.px, &t, 8);
    t = make_uint2(a.z, a.w), __builtin_memcpy(&o.py, &t, 8);
    t = make_uint2(b.x, b.y), __builtin_memcpy(&o.pz, &t, 8);
    o.mass = __uint_as_float(b.z), o.family = b.w;
    o.qw = __uint_as_float(c.x), o.qx = __uint_as_float(c.y), o.qy = __uint_as_float(c.z), o.qz = __uint_as_float(c.w);
    o.vx = __uint_as_float(d.x), o.vy = __uint_as_float(d.y), o.vz = __uint_as_float(d.z), o.wx = __uint_as_float(d.w);
    o.wy = __uint_as_float(e.x), o.wz = __uint_as_float(e.y);
    return o;
}

// One contact of the hot classes between two staged owners.  Arithmetic: forces_fast_body (deme_force_fast.h), with the
// owner-level quantities taken from the staged records.  Returns the world-frame force on A, the torques about A's and B's
// centres, and the updated history.
template <int MODEL>
__device__ inline void tile_contact(const DevParams& p, const uint2 inf, const TileOwner& A, const TileOwner& B, float4& hist, f3& force,
                                    f3& tA, f3& tB) {
    const uint32_t cls = (inf.x >> 20) & 3u;
    const float4 cA = p.comp[inf.y & 0x1FFFu];
    const uint32_t matA = (inf.x >> 24) & 15u;
    // sphere offsets with the reference's own rounding (no contraction: deme_device.h), see forces_fast_body
    const RotM RA = rot_coeffs(A.qw, A.qx, A.qy, A.qz);
    const RotM RB = rot_coeffs(B.qw, B.qx, B.qy, B.qz);
    const f3 relA = rot_apply(RA, mk3(cA.x, cA.y, cA.z));
    const float rA = cA.w;
    float extraMargin = 0.f;
    if (!p.familyTrivial) {
        const float eA = p.familyExtra[A.family & 0xFFu], eB = p.familyExtra[B.family & 0xFFu];
        extraMargin = fmaxf(eA, eB);
    }
    const double dOx = A.px - B.px, dOy = A.py - B.py, dOz = A.pz - B.pz;
    const f3 dO = mk3((float)dOx, (float)dOy, (float)dOz);
    f3 n, rAv, rBv;
    float depth, rB, massB;
    uint32_t matB;
    bool touching;
    if (cls == DEME_KEY_CLASS_SS) {
        const float4 cB = p.comp[(inf.y >> 13) & 0x1FFFu];
        matB = inf.x >> 28;
        rB = cB.w;
        massB = B.mass;
        const f3 relB = rot_apply(RB, mk3(cB.x, cB.y, cB.z));
        const double dx = (dOx + (double)relA.x) - (double)relB.x;
        const double dy = (dOy + (double)relA.y) - (double)relB.y;
        const double dz = (dOz + (double)relA.z) - (double)relB.z;
        const double d2 = dx * dx + dy * dy + dz * dz;
        const float sumR = rA + rB;
        const double num = (double)sumR * (double)sumR - d2;
        const float d2f = (float)d2;
        const float inv = frsq(d2f);
        const float dist = d2f * inv;
        n = mk3((float)dx * inv, (float)dy * inv, (float)dz * inv);
        depth = (float)num * frcp(sumR + dist);
        touching = !(depth < -extraMargin);
        const float s = rB - 0.5f * depth;
        rBv = mk3(relB.x + s * n.x, relB.y + s * n.y, relB.z + s * n.z);
        rAv = fsub(rBv, dO);
    } else {  // sphere-analytical: the reference's arithmetic in the tile's frame (only differences of positions enter)
        const AnalObj ob = p.anal[(inf.y >> 13) & 0x1FFFu];
        matB = ob.mat;
        rB = 1e15f;  // DEME_HUGE_FLOAT
        massB = ob.mass;
        const f3 relB = rot_apply(RB, mk3(ob.relx, ob.rely, ob.relz));
        const f3 dir = rot_apply(RB, mk3(ob.rotx, ob.roty, ob.rotz));
        const d3 PA{A.px, A.py, A.pz}, PB{B.px, B.py, B.pz};
        const d3 bodyA{PA.x + (double)relA.x, PA.y + (double)relA.y, PA.z + (double)relA.z};
        const d3 bodyB{PB.x + (double)relB.x, PB.y + (double)relB.y, PB.z + (double)relB.z};
        d3 cp;
        double dd;
        sphere_entity(bodyA, rA, ob.type, bodyB, dir, ob.size1, ob.normal, 0.0f, cp, n, dd);
        depth = (float)dd;
        touching = !(dd < -(double)extraMargin);
        rAv = mk3((float)(cp.x - PA.x), (float)(cp.y - PA.y), (float)(cp.z - PA.z));
        rBv = mk3((float)(cp.x - PB.x), (float)(cp.y - PB.y), (float)(cp.z - PB.z));
    }
    force = mk3(0, 0, 0);
    f3 torque_only = mk3(0, 0, 0);
    if (touching) {
        if (depth > 0.f) {
            const float massA = A.mass;
            const MatPair mp = p.matPair[matA * p.nMat + matB];
            const f3 rotVelA = fcross(mk3(A.wx, A.wy, A.wz), rAv), rotVelB = fcross(mk3(B.wx, B.wy, B.wz), rBv);
            const f3 velB2A = fsub(fadd(mk3(A.vx, A.vy, A.vz), rotVelA), fadd(mk3(B.vx, B.vy, B.vz), rotVelB));
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
                        const float ft_max = fabsf(Fn) * mp.mu;
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
            hist = make_float4(0, 0, 0, 0);
        }
        const f3 tot = fadd(force, torque_only);
        tA = fcross(rAv, tot);
        tB = fcross(tot, rBv);  // = r_B x (-F)
    } else {
        tA = mk3(0, 0, 0), tB = mk3(0, 0, 0);
        hist = make_float4(0, 0, 0, 0);  // _forceModelContactWildcardDestroy_
    }
}

// which tile a workgroup takes: the XCD-aware order of force_block_id (deme_force.h) over tiles
__device__ inline uint32_t tile_block_id(uint32_t G) {
    const uint32_t b = blockIdx.x;
    if (G == 0)
        return b;
    const uint32_t xcd = b & 7u, j = b >> 3;
    return ((j / G) * 8u + xcd) * G + (j % G);
}

// Segmented sum of six values over the wavefront + accumulation into the wavefront's own LDS sums.  `slot` ascends along the
// lanes (runs of equal slots; DEME_TILE_NONE = nothing to add, sorted last).  After the scan the last lane of every run holds
// the run's total and adds it to acc[slot]: distinct slots within the instruction, and the wavefront's LDS operations complete
// in program order, so the read-modify-write needs no atomic.
#define DEME_TILE_NONE 0x3FFu
#define DEME_TILE_ACC 8  // floats per slot of a wavefront's sums (two float4)
__device__ inline float tile_dpp(float v, const int ctrl, const int rowMask) {
    // (ctrl / rowMask are compile-time constants at every call site; the builtin needs literals, hence the switch)
    const int i = __builtin_bit_cast(int, v);
    int r;
    switch (ctrl) {
        case 0x111: r = __builtin_amdgcn_update_dpp(0, i, 0x111, 0xF, 0xF, true); break;  // row_shr:1
        case 0x112: r = __builtin_amdgcn_update_dpp(0, i, 0x112, 0xF, 0xF, true); break;
        case 0x114: r = __builtin_amdgcn_update_dpp(0, i, 0x114, 0xF, 0xF, true); break;
        case 0x118: r = __builtin_amdgcn_update_dpp(0, i, 0x118, 0xF, 0xF, true); break;
        case 0x142: r = __builtin_amdgcn_update_dpp(0, i, 0x142, 0xA, 0xF, true); break;  // row_bcast:15 into rows 1 and 3
        default: r = __builtin_amdgcn_update_dpp(0, i, 0x143, 0xC, 0xF, true); break;      // row_bcast:31 into rows 2 and 3
    }
    (void)rowMask;
    return __builtin_bit_cast(float, r);
}
__device__ inline void tile_seg_accumulate(float* acc, const uint32_t lane, const uint32_t slot, float v[6]) {
    // run structure: head = first lane of a run; dist = lanes back to the head of my run
    const uint32_t prev = (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)slot, 0x138, 0xF, 0xF, false);  // wave_shr:1
    const bool head = lane == 0u || prev != slot;
    const uint64_t heads = __ballot(head);
    const uint64_t upto = heads & (~0ull >> (63u - lane));
    const uint32_t headLane = 63u - (uint32_t)__clzll((long long)upto);
    const uint32_t dist = lane - headLane;
    const bool tail = lane == 63u || ((heads >> (lane + 1u)) & 1ull);
    const uint32_t row = lane & 48u;
    const float f1 = dist >= 1u ? 1.f : 0.f, f2 = dist >= 2u ? 1.f : 0.f, f4 = dist >= 4u ? 1.f : 0.f, f8 = dist >= 8u ? 1.f : 0.f;
    const float f15 = headLane < row ? 1.f : 0.f;   // my run began in an earlier row (rows 1, 3 take lane 15 / 47 of the row before)
    const float f31 = headLane < 32u ? 1.f : 0.f;   // rows 2, 3: my run began in the lower half
#pragma unroll
    for (int k = 0; k < 6; k++)
        v[k] = __builtin_fmaf(tile_dpp(v[k], 0x111, 0xF), f1, v[k]);
#pragma unroll
    for (int k = 0; k < 6; k++)
        v[k] = __builtin_fmaf(tile_dpp(v[k], 0x112, 0xF), f2, v[k]);
#pragma unroll
    for (int k = 0; k < 6; k++)
        v[k] = __builtin_fmaf(tile_dpp(v[k], 0x114, 0xF), f4, v[k]);
#pragma unroll
    for (int k = 0; k < 6; k++)
        v[k] = __builtin_fmaf(tile_dpp(v[k], 0x118, 0xF), f8, v[k]);
#pragma unroll
    for (int k = 0; k < 6; k++)
        v[k] = __builtin_fmaf(tile_dpp(v[k], 0x142, 0xA), f15, v[k]);
#pragma unroll
    for (int k = 0; k < 6; k++)
        v[k] = __builtin_fmaf(tile_dpp(v[k], 0x143, 0xC), f31, v[k]);
    if (tail && slot < DEME_TILE_NB) {
        float4* a = reinterpret_cast<float4*>(acc + slot * DEME_TILE_ACC);
        float4 x = a[0], y = a[1];
        x.x += v[0], x.y += v[1], x.z += v[2];
        y.x += v[3], y.y += v[4], y.z += v[5];
        a[0] = x, a[1] = y;
    }
}

#ifndef DEME_TILE_OCC
#define DEME_TILE_OCC 1
#endif
template <int MODEL>
__global__ __launch_bounds__(DEME_TILE_T, DEME_TILE_OCC) void k_tile_forces(const DevParams p, const TileArgs a) {
    __shared__ uint4 sOwn[(DEME_TILE_NB + DEME_TILE_HMAX) * DEME_TILE_REC];
    __shared__ float4 sAcc[DEME_TILE_T / 64][DEME_TILE_NB * DEME_TILE_ACC / 4];
    __shared__ float4 sScat[DEME_TILE_T / 64][64 * 2];  // per wavefront: the chunk's B-side contributions in B-slot order
    const uint32_t t = tile_block_id(a.xcdGroup);
    if (t >= a.nTiles)
        return;
    if (a.tileMode && !(a.tileMode[t] & (1u << a.pass)))
        return;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t o0 = t * DEME_TILE_NB;
    const uint32_t nLoc = min((uint32_t)DEME_TILE_NB, a.nOwners - o0);
    const uint32_t nH = a.hCount[t];
    const uint32_t c0 = a.aStart[o0], c1 = a.aStart[o0 + nLoc];
    // the first chunk's streams do not depend on the staging: issue them first
    const float4* wc4 = reinterpret_cast<const float4*>(a.wc);
    uint32_t c = c0 + wave * 64u + lane;
    uint2 inf = make_uint2(0, 0);
    float4 hist = make_float4(0, 0, 0, 0);
    if (c < c1) {
        inf = stream_load(a.tInfo + c);
        if (MODEL == 0)
            hist = stream_load(wc4 + c);
    }
    {   // stage the tile's owners and its halo
        int64_t u0x, u0y, u0z;
        {
            const OwnerRec f = load_owner(a.owners, o0);  // (one address for the whole workgroup)
            pos_units(f, p, u0x, u0y, u0z);
        }
        for (uint32_t s = tid; s < nLoc + nH; s += DEME_TILE_T) {
            const bool loc = s < nLoc;
            const uint32_t id = loc ? o0 + s : a.hList[(size_t)t * DEME_TILE_HMAX + (s - nLoc)];
            const OwnerRec r = load_owner(a.owners, id);
            tile_stage_owner(p, r, u0x, u0y, u0z, sOwn + (loc ? s : DEME_TILE_NB + (s - nLoc)) * DEME_TILE_REC);
        }
        float4* z = &sAcc[0][0];
        for (uint32_t i = tid; i < (DEME_TILE_T / 64) * DEME_TILE_NB * DEME_TILE_ACC / 4; i += DEME_TILE_T)
            z[i] = make_float4(0, 0, 0, 0);
    }
    __syncthreads();
    float* acc = reinterpret_cast<float*>(sAcc[wave]);
    float4* scat = sScat[wave];
    for (uint32_t base = c0 + wave * 64u; base < c1; base += DEME_TILE_T) {
        const bool valid = c < c1;
        // the next chunk's streams, before this chunk's arithmetic
        const uint32_t cn = c + DEME_TILE_T;
        uint2 infN = make_uint2(0, 0);
        float4 histN = make_float4(0, 0, 0, 0);
        if (cn < c1) {
            infN = stream_load(a.tInfo + cn);
            if (MODEL == 0)
                histN = stream_load(wc4 + cn);
        }
        uint32_t slotA = DEME_TILE_NONE, slotB = DEME_TILE_NONE;
        float va[6] = {0, 0, 0, 0, 0, 0};
        float4 bF = make_float4(0, 0, 0, __uint_as_float(DEME_TILE_NONE)), bT = make_float4(0, 0, 0, 0);
        if (valid) {
            slotA = inf.x & 1023u, slotB = (inf.x >> 10) & 1023u;
            const TileOwner A = tile_read_owner(sOwn, slotA), B = tile_read_owner(sOwn, slotB);
            f3 force, tA, tB;
            tile_contact<MODEL>(p, inf, A, B, hist, force, tA, tB);
            if (MODEL == 0)
                stream_store(reinterpret_cast<float4*>(a.wc) + c, hist);
            va[0] = force.x, va[1] = force.y, va[2] = force.z, va[3] = tA.x, va[4] = tA.y, va[5] = tA.z;
            if (slotB < DEME_TILE_NB) {
                bF = make_float4(-force.x, -force.y, -force.z, __uint_as_float(slotB));
                bT = make_float4(tB.x, tB.y, tB.z, 0.f);
            } else if (inf.x & (1u << 22)) {
                stream_store(a.conB4 + c, make_float4(-force.x, -force.y, -force.z, tB.x));
                stream_store(a.conB2 + c, make_float2(tB.y, tB.z));
            }
        }
        // A side: the lanes come sorted by A's slot (lanes past the end of the tile's range carry DEME_TILE_NONE)
        tile_seg_accumulate(acc, lane, slotA, va);
        // B side: through LDS into B-slot order (the builder's rank; lanes past the end keep their own position: the ranks of the
        // valid lanes of a chunk are 0 .. nValid - 1)
        {
            const uint32_t rank = valid ? (inf.y >> 26) : lane;
            wave_lds_fence();
            scat[2 * rank] = bF, scat[2 * rank + 1] = bT;
            wave_lds_fence();
            const float4 gF = scat[2 * lane], gT = scat[2 * lane + 1];
            float vb[6] = {gF.x, gF.y, gF.z, gT.x, gT.y, gT.z};
            tile_seg_accumulate(acc, lane, __float_as_uint(gF.w), vb);
        }
        c = cn, inf = infN, hist = histN;
    }
    __syncthreads();
    if (tid < 2u * nLoc) {  // the four wavefronts' copies in a fixed order; thread = (owner, half of its record)
        const float4 w0 = sAcc[0][tid], w1 = sAcc[1][tid], w2 = sAcc[2][tid], w3 = sAcc[3][tid];
        float4 r;
        r.x = ((w0.x + w1.x) + w2.x) + w3.x, r.y = ((w0.y + w1.y) + w2.y) + w3.y;
        r.z = ((w0.z + w1.z) + w2.z) + w3.z, r.w = ((w0.w + w1.w) + w2.w) + w3.w;
        a.tSum[2 * (size_t)o0 + tid] = r;  // (F.x F.y F.z 0 | t.x t.y t.z 0): the layout of aSum
    }
}

// ---- per-detection builder ------------------------------------------------------------------------------------------------------
// flag[j] = 1 iff the j-th entry of the B-sorted contact list crosses a tile boundary (its B owner's sum needs a record);
// flag[nC] = 0 closes the scan
__global__ __launch_bounds__(256) void k_tile_rflag(uint32_t nC, const uint32_t* __restrict__ bIdx, const uint4* __restrict__ info,
                                                    uint32_t* __restrict__ flag) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j > nC)
        return;
    uint32_t f = 0;
    if (j < nC) {
        const uint4 ci = info[bIdx[j]];
        f = ((ci.x & 0x3FFFFFFFu) / DEME_TILE_NB != ci.y / DEME_TILE_NB) ? 1u : 0u;
    }
    flag[j] = f;
}
// rIdx: the crossing contacts in B-owner order; rStart[o] = first of owner o's (o = 0 .. nOwners)
__global__ __launch_bounds__(256) void k_tile_rfill(uint32_t nC, uint32_t nOwners, const uint32_t* __restrict__ bIdx,
                                                    const uint32_t* __restrict__ flag, const uint32_t* __restrict__ rPos,
                                                    const uint32_t* __restrict__ bStart, uint32_t* __restrict__ rIdx,
                                                    uint32_t* __restrict__ rStart) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < nC && flag[j])
        rIdx[rPos[j]] = bIdx[j];
    if (j <= nOwners)
        rStart[j] = rPos[bStart[j]];
}

// one workgroup per tile: the sorted list of the foreign owners its contacts touch, the 8-byte gather records, the pass of the
// halo overlap the tile belongs to
__global__ __launch_bounds__(256) void k_tile_build(const DevParams p, uint32_t nOwners, const uint4* __restrict__ info,
                                                    const uint32_t* __restrict__ aStart, const OwnerRec* __restrict__ owners,
                                                    uint2* __restrict__ tInfo, uint32_t* __restrict__ hList,
                                                    uint32_t* __restrict__ hCount, uint32_t* __restrict__ tileMode,
                                                    RangeCounters* rc) {
    __shared__ uint32_t table[DEME_TILE_HASH];
    __shared__ uint32_t list[DEME_TILE_HP2];
    __shared__ uint32_t nU, nL, anyGhost;
    const uint32_t t = blockIdx.x, tid = threadIdx.x;
    const uint32_t o0 = t * DEME_TILE_NB, o1 = min(o0 + (uint32_t)DEME_TILE_NB, nOwners);
    const uint32_t c0 = aStart[o0], c1 = aStart[o1];
    for (uint32_t i = tid; i < DEME_TILE_HASH; i += 256)
        table[i] = 0xFFFFFFFFu;
    for (uint32_t i = tid; i < DEME_TILE_HP2; i += 256)
        list[i] = 0xFFFFFFFFu;
    if (tid == 0)
        nU = 0, nL = 0, anyGhost = 0;
    __syncthreads();
    for (uint32_t c = c0 + tid; c < c1; c += 256) {
        const uint32_t ob = info[c].y;
        if (ob >= o0 && ob < o1)
            continue;
        uint32_t h = (ob * 2654435761u) >> 22;  // 10 bits
        while (*(volatile uint32_t*)&nU <= DEME_TILE_HMAX) {
            const uint32_t old = atomicCAS(&table[h], 0xFFFFFFFFu, ob);
            if (old == 0xFFFFFFFFu) {
                atomicAdd(&nU, 1u);
                break;
            }
            if (old == ob)
                break;
            h = (h + 1u) & (DEME_TILE_HASH - 1u);
        }
    }
    __syncthreads();
    const uint32_t n = nU;
    if (n > DEME_TILE_HMAX) {  // the halo does not fit the LDS area of k_tile_forces: this list is evaluated by k_forces_fast
        if (tid == 0) {
            atomicOr(&rc->tileOverflow, 1u);
            hCount[t] = 0;
        }
        return;
    }
    for (uint32_t i = tid; i < DEME_TILE_HASH; i += 256) {
        const uint32_t v = table[i];
        if (v != 0xFFFFFFFFu)
            list[atomicAdd(&nL, 1u)] = v;
    }
    // bitonic sort (ascending; the padding sorts to the end)
    for (uint32_t k = 2; k <= DEME_TILE_HP2; k <<= 1)
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            __syncthreads();
            for (uint32_t i = tid; i < DEME_TILE_HP2; i += 256) {
                const uint32_t ixj = i ^ j;
                if (ixj > i) {
                    const uint32_t x = list[i], y = list[ixj];
                    const bool up = (i & k) == 0;
                    if ((x > y) == up)
                        list[i] = y, list[ixj] = x;
                }
            }
        }
    __syncthreads();
    for (uint32_t i = tid; i < n; i += 256)
        hList[(size_t)t * DEME_TILE_HMAX + i] = list[i];
    if (tid == 0) {
        hCount[t] = n;
        atomicMax(&rc->tileMaxHalo, n);
    }
    if (tileMode) {  // a tile that stages a ghost's record waits for the ghosts of this step (pass 1)
        bool g = false;
        for (uint32_t i = tid; i < (o1 - o0) + n; i += 256)
            g = g || ghost_of(owners[i < o1 - o0 ? o0 + i : list[i - (o1 - o0)]].family);
        if (g)
            anyGhost = 1;
        __syncthreads();
        if (tid == 0)
            tileMode[t] = anyGhost ? 2u : 1u;
    }
    const uint32_t lane = tid & 63u;
    for (uint32_t base = c0 + (tid & ~63u); base < c1; base += 256) {  // a wavefront = one 64-contact chunk of k_tile_forces
        const uint32_t c = base + lane;
        const bool valid = c < c1;
        uint4 ci = make_uint4(0, 0, 0, 0);
        if (valid)
            ci = info[c];
        const uint32_t oa = ci.x & 0x3FFFFFFFu, cls = ci.x >> 30, ob = ci.y;
        uint32_t slotB = DEME_TILE_NONE, rec = 0;
        if (valid) {
            if (ob >= o0 && ob < o1) {
                slotB = ob - o0;
            } else {
                uint32_t lo = 0, hi = n;
                while (lo < hi) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (list[mid] < ob)
                        lo = mid + 1;
                    else
                        hi = mid;
                }
                slotB = DEME_TILE_NB + lo;
                rec = 1;
            }
        }
        // rank among the chunk's valid lanes by (B's slot if it lives in this tile, else last; lane)
        const uint32_t key = ((slotB < DEME_TILE_NB ? slotB : DEME_TILE_NONE) << 6) | lane;
        uint32_t rank = 0;
        for (int j = 0; j < 64; j++) {
            const uint32_t kj = (uint32_t)__shfl((int)key, j);
            const int vj = __shfl((int)valid, j);
            rank += (vj && kj < key) ? 1u : 0u;
        }
        if (valid) {
            const uint32_t matA = ci.z >> 16, compA = ci.z & 0xFFFFu;
            const uint32_t matB = (cls == DEME_KEY_CLASS_SS) ? (ci.w >> 16) : 0u;  // (an analytical object's material is in its record)
            const uint32_t wB = (cls == DEME_KEY_CLASS_SS) ? (ci.w & 0xFFFFu) : ci.w;
            tInfo[c] = make_uint2(tile_info_x(oa - o0, slotB, cls, rec, matA, matB), compA | (wB << 13) | (rank << 26));
        }
    }
}

}  // namespace deme_dev

#pragma clang fp contract(off)
