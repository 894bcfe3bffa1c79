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
//   * the sums are PULLED by the owners: the tile's contacts are evaluated in rounds of 256 (one per thread); a round leaves
//     every contact's two contributions in LDS, and after a barrier thread o (of 128) adds the round's part of owner o's A run
//     -- a contiguous range of the list -- to its registers while thread 128 + o adds the round's part of owner o's list of
//     contacts that hold it as B (tile-local positions the builder left behind), both in ascending order: no atomics, no
//     ordering between wavefronts, a fixed summation order, reproducible run to run.  (LDS float atomics were measured at ~2 cycles
//     PER LANE on gfx950 -- 12 of them per contact made the first version of this kernel 2.6 times slower than k_forces_fast -- and
//     in-wavefront segmented DPP scans of the six sums cost 230 instructions per 64 contacts, a third of the kernel.)
//     The sums leave the workgroup as ONE 32-byte record per owner; only a contact whose B owner lives in another tile writes a
//     24-byte record, which the integrator gathers through the per-owner list of such contacts (deme_kernels.h:
//     k_integrate<true> with GatherArgs::tile).
// What the per-detection builder below leaves behind: tInfo (8 B per contact), the halo list of every tile, the per-owner lists
// of tile-crossing contacts.  A tile whose halo, contact range or local lists do not fit the LDS area is evaluated by
// k_tile_forces_big (below: one workgroup, nothing staged, every contact writes a record); the rest of the list stays tiled.
#pragma once
#include "deme_force_fast.h"

#ifndef DEME_TILE_NB
#define DEME_TILE_NB 128  // owners per tile
#endif
#ifndef DEME_TILE_HMAX
#define DEME_TILE_HMAX 192  // foreign owners a tile can stage (measured at 1e6 packed clumps: mean 93, largest 147)
#endif
#ifndef DEME_TILE_T
#define DEME_TILE_T 256  // threads per tile (a round = DEME_TILE_T contacts); the first 2 * DEME_TILE_NB of them pull the sums
#endif
#define DEME_TILE_HP2 256  // DEME_TILE_HMAX rounded up to a power of two (bitonic sort of the halo list)
#define DEME_TILE_LREG 4      // entries of a tile's local-B lists a thread carries from its load to the staging
#define DEME_TILE_LMAX (DEME_TILE_LREG * DEME_TILE_T)   // entries of a tile's local-B lists staged in LDS
static_assert(DEME_TILE_T >= 2 * DEME_TILE_NB, "one pulling thread per owner and side");
static_assert(DEME_TILE_NB + DEME_TILE_HMAX <= 2 * DEME_TILE_T, "a thread stages at most two owner records");
static_assert(DEME_TILE_LMAX == DEME_TILE_LREG * DEME_TILE_T, "list entries per thread");
#define DEME_TILE_HASH 1024u
#ifndef DEME_TILE_REC
#define DEME_TILE_REC 6    // uint4 per staged owner (96 bytes) for the built-in models
#endif
// A run-time compiled model (MODEL 2, deme_jit.h) is written against the reference's body-frame vocabulary (AOriQ, ARotVel, locCPA):
// its staged record carries the quaternion and the body-frame angular velocity instead of the nine rotation coefficients and the
// world-frame one -- 80 bytes; the coefficients are formed per contact.
#define DEME_TILE_REC_USER 5
#ifndef DEME_JIT_NW
#define DEME_JIT_NW 1  // contact wildcards of the run-time compiled model (the generated source defines it; at least 1)
#endif
#ifndef DEME_JIT_HAS_WC
#define DEME_JIT_HAS_WC 1  // 0: the model declares no contact wildcards (nothing is streamed)
#endif
__host__ __device__ constexpr uint32_t tile_rec16(int model) { return model == 2 ? DEME_TILE_REC_USER : DEME_TILE_REC; }
#define DEME_TILE_CMAX 8192   // contacts of one tile (their tile-local positions are 16-bit)
static_assert(DEME_TILE_NB + DEME_TILE_HMAX <= 1024, "slot numbers are 10 bits");
static_assert(DEME_TILE_HMAX <= DEME_TILE_HP2 && DEME_TILE_HMAX <= 256, "halo list: one entry per builder thread");

#pragma clang fp contract(fast)

namespace deme_dev {

// Measurement builds (make variant EXTRA=-DDEME_TILE_KI=bits): "knock-in" duplicates of one section of k_tile_forces behind opaque
// values -- the physics is untouched, the extra time of a variant is what that section costs (bit 0: the two sphere-offset
// rotations + their fp64 conversions; 1: the pulls; 2: the staging conversion; 3: the whole per-contact evaluation) -- and
// "knock-outs" of stores (bit 4: history; 5: crossing records; 6: tile sums: wrong results, timing only); bit 7: one more level of
// dependent loads in front of the ids of the foreign owners; bit 8: one more barrier per round.
#ifndef DEME_TILE_KI
#define DEME_TILE_KI 0
#endif
// -DDEME_TILE_STAMPS=1: thread 0 of every tile leaves the 100 MHz wall clock at its phase boundaries (words 0 start, 1 tables in LDS,
// 2 staged, 3.. the end of each round, 11 the end) and where it ran (word 12: HW_ID | XCC_ID << 32; 13: contacts; 14: foreign owners;
// 15: shader cycles (s_memtime) over the tile's life -- with words 0 and 11 the clock the kernel runs at)
#ifndef DEME_TILE_STAMPS
#define DEME_TILE_STAMPS 0
#endif
#if DEME_TILE_STAMPS
#define TILE_STAMP(k)                                                                  \
    do {                                                                               \
        if (a.stamps && threadIdx.x == 0)                                              \
            a.stamps[(size_t)t * 16u + (k)] = (unsigned long long)wall_clock64();      \
    } while (0)
#else
#define TILE_STAMP(k) \
    do {              \
    } while (0)
#endif
typedef float v2f __attribute__((ext_vector_type(2)));  // a register pair: += compiles to v_pk_add_f32
__device__ inline void ki_opaque(float& v) { asm volatile("" : "+v"(v)); }
__device__ inline void ki_opaque(uint32_t& v) { asm volatile("" : "+v"(v)); }
__device__ inline void ki_sink(float v) { asm volatile("" ::"v"(v)); }
__device__ inline void ki_sink(uint32_t v) { asm volatile("" ::"v"(v)); }

// tInfo.x: slot of A (10) | slot of B (10) | class (2) | B lives in another tile: write a record (1) | 1 spare | material of A (4) |
//          material of B (4);  tInfo.y: component of A (16) | component of B or analytical-object index (16)
__host__ __device__ inline uint32_t tile_info_x(uint32_t slotA, uint32_t slotB, uint32_t cls, uint32_t rec, uint32_t matA, uint32_t matB) {
    return slotA | (slotB << 10) | (cls << 20) | (rec << 22) | (matA << 24) | (matB << 28);
}

struct TileArgs {
    const OwnerRec* owners;
    const uint2* tInfo;
    const uint32_t* aStart;
    const uint32_t* hList;     // DEME_TILE_HMAX entries per tile
    const uint32_t* hCount;
    // per tile: the origin of its frame in sub-voxel units (x, y, z) -- its first owner's position when the list was built; any
    // fixed point near the tile serves, and one that does not move between detections is not another load behind the records
    const int64_t* org;
    const uint32_t* tileMode;  // halo overlap: bit p = the tile is evaluated in pass p (1: reads no ghost owner, 2: does); null: no split
    const uint16_t* lOff;      // per tile NB + 1 entries: where each owner's list of contacts that hold it as B with A in the same tile begins
    const uint32_t* lCount;    // per tile: entries of all its lists together
    const uint16_t* lPos;      // those contacts as positions in their tile's range of the list, ascending per owner; a tile's lists start
                               // at the index of the tile's first contact (there are never more of them than the tile has contacts)
    float* wc;
    float4* tSum;              // two float4 per owner: the sum of the contributions of contacts evaluated by the owner's tile
    float4* rec32;             // the B-side records of the contacts whose B owner lives in another tile: 32 bytes each, dense, in
                               // list order (the k-th such contact writes record k: full cache lines instead of scattered pieces)
    const uint32_t* rankC;     // per contact: the number of such contacts before it
    // scenes with a mesh (k_tile_forces<MODEL, true>): the sphere-triangle contacts were evaluated by the mesh variant of the general
    // kernel just before (launch_forces); the tile takes their per-contact records instead of evaluating them
    const float4 *conA4, *conB4;
    const float2 *conA2, *conB2;
    const uint32_t* tileBig;   // per tile: 1 = its halo / contact range / local lists do not fit: k_tile_forces leaves it to k_tile_forces_big
    const uint32_t* bigList;   // those tiles (k_tile_forces_big takes one per workgroup)
    const uint4* info;         // their contacts are evaluated from the 16-byte gather records (owner ids instead of staging slots)
    // a run-time compiled model (MODEL 2): sphere ids of the pairs, the user's owner / geometry wildcard arrays, the time
    const uint64_t* keys;
    float* ownerWc[8];
    float* geoWcSph[8];
    float* geoWcAnal[8];
    float timeElapsed;
    float* rec[4];             // contact recording (REC variants): force, torque-only force, contact point in A's / B's body frame
    uint32_t nOwners, nTiles, pass, xcdGroup;
    // Stride of the staged records in LDS, in 16-byte units: tile_rec16(model), or ONE MORE for the built-in models when the workgroups
    // per CU stay the same -- 96-byte records start on 8 different banks (24 words: gcd(24, 64) = 8), so two lanes' ds_read_b128 of two
    // random slots collide twice as often as with 112-byte records (28 words: 16 starting banks x 4 words = all 64); measured
    // -3 us on the pass (profiles/r04/r04p_record_stride.txt)
    uint32_t rs16;
    // ... and where a padded record would cost a workgroup (a list whose largest tile stages more than 128 foreign owners): the six
    // cells of every record whose slot has bit 3 set are rotated by one (swz = 1) -- 6 s + f + bit3(s) runs through all 16 starting
    // bank groups over 16 consecutive slots, at no cost in LDS
    uint32_t swz;
    uint32_t hCap, lCap;       // LDS capacities of this launch: foreign owners / local-B list entries of the largest tile (rounded up)
    uint32_t nComp, nAnal, nMass;  // table sizes (tile_table_bytes)
    uint32_t* tileCtr;             // k_tile_forces_p (deme_tile_p.h): tiles handed out per partition, workgroups through; zero between launches
    uint32_t ctrParts;             // ... partitions of the tiles (and of the workgroups), one counter each
    unsigned long long* stamps;    // measurement builds (-DDEME_TILE_STAMPS=1): 16 words per tile, see k_tile_forces
};

// one staged owner, as the contact loop reads it back from LDS
struct TileOwner {
    double px, py, pz;  // position relative to the tile's origin [m]
    float mass;
    uint32_t family;
    RotM R;            // rotation coefficients of the orientation (reference rounding: rot_coeffs of deme_device.h)
    float vx, vy, vz;
    float wx, wy, wz;  // angular velocity in the WORLD frame (MODEL 2: in the body frame, as the user vocabulary wants it)
    float qw, qx, qy, qz;  // MODEL 2 only
};

// `rot` (0 / 1): the record's six 16-byte cells rotated by one (TileArgs::swz) -- cell f lies at dst[(f + rot) % 6]
__device__ inline void tile_stage_owner_m(const DevParams& p, const float mass, const OwnerRec& r, int64_t u0x, int64_t u0y, int64_t u0z,
                                          uint4* dst0, uint32_t rot = 0u) {
    uint4* const dst = dst0 + rot;
    uint4* const last = dst0 + (rot ? 0 : 5);
    int64_t ux, uy, uz;
    pos_units(r, p, ux, uy, uz);
    const double px = (double)(ux - u0x) * p.l, py = (double)(uy - u0y) * p.l, pz = (double)(uz - u0z) * p.l;
    const RotM R = rot_coeffs(r.qw, r.qx, r.qy, r.qz);
    const f3 w = frot_apply(R, mk3(r.wx, r.wy, r.wz));
    uint2 bx, by, bz;
    __builtin_memcpy(&bx, &px, 8), __builtin_memcpy(&by, &py, 8), __builtin_memcpy(&bz, &pz, 8);
    dst[0] = make_uint4(bx.x, bx.y, by.x, by.y);
    dst[1] = make_uint4(bz.x, bz.y, __float_as_uint(mass), r.family);
    dst[2] = make_uint4(__float_as_uint(R.xx), __float_as_uint(R.xy), __float_as_uint(R.xz), __float_as_uint(R.yx));
    dst[3] = make_uint4(__float_as_uint(R.yy), __float_as_uint(R.yz), __float_as_uint(R.zx), __float_as_uint(R.zy));
    dst[4] = make_uint4(__float_as_uint(R.zz), __float_as_uint(r.vx), __float_as_uint(r.vy), __float_as_uint(r.vz));
    // (the spare word takes the record's last one, the detection margin: nobody reads it back, but a word of a load that is dead on
    // arrival is a register the allocator hands out again at once -- and whoever writes it has to wait for the load to land first)
    *last = make_uint4(__float_as_uint(w.x), __float_as_uint(w.y), __float_as_uint(w.z), __float_as_uint(r.margin));
}

// MODEL 2: position, mass, family | inertia offset << 16, quaternion, velocity, body-frame angular velocity
__device__ inline void tile_stage_owner_user(const DevParams& p, const float mass, const OwnerRec& r, int64_t u0x, int64_t u0y, int64_t u0z,
                                             uint4* dst) {
    int64_t ux, uy, uz;
    pos_units(r, p, ux, uy, uz);
    const double px = (double)(ux - u0x) * p.l, py = (double)(uy - u0y) * p.l, pz = (double)(uz - u0z) * p.l;
    uint2 bx, by, bz;
    __builtin_memcpy(&bx, &px, 8), __builtin_memcpy(&by, &py, 8), __builtin_memcpy(&bz, &pz, 8);
    dst[0] = make_uint4(bx.x, bx.y, by.x, by.y);
    dst[1] = make_uint4(bz.x, bz.y, __float_as_uint(mass), (r.family & 0xFFFFu) | ((uint32_t)r.inertiaOff << 16));
    dst[2] = make_uint4(__float_as_uint(r.qw), __float_as_uint(r.qx), __float_as_uint(r.qy), __float_as_uint(r.qz));
    dst[3] = make_uint4(__float_as_uint(r.vx), __float_as_uint(r.vy), __float_as_uint(r.vz), __float_as_uint(r.wx));
    dst[4] = make_uint4(__float_as_uint(r.wy), __float_as_uint(r.wz), 0u, 0u);
}
__device__ inline TileOwner tile_read_owner_user(const uint4* sOwn, uint32_t at) {
    const uint4* q = sOwn + at;
    const uint4 a = q[0], b = q[1], c = q[2], d = q[3], e = q[4];
    TileOwner o;
    uint2 t;
    t = make_uint2(a.x, a.y), __builtin_memcpy(&o.px, &t, 8);
    t = make_uint2(a.z, a.w), __builtin_memcpy(&o.py, &t, 8);
    t = make_uint2(b.x, b.y), __builtin_memcpy(&o.pz, &t, 8);
    o.mass = __uint_as_float(b.z), o.family = b.w;
    o.qw = __uint_as_float(c.x), o.qx = __uint_as_float(c.y), o.qy = __uint_as_float(c.z), o.qz = __uint_as_float(c.w);
    o.R = rot_coeffs(o.qw, o.qx, o.qy, o.qz);
    o.vx = __uint_as_float(d.x), o.vy = __uint_as_float(d.y), o.vz = __uint_as_float(d.z);
    o.wx = __uint_as_float(d.w), o.wy = __uint_as_float(e.x), o.wz = __uint_as_float(e.y);
    return o;
}
template <int MODEL>
__device__ inline void tile_stage(const DevParams& p, const float mass, const OwnerRec& r, int64_t u0x, int64_t u0y, int64_t u0z, uint4* dst,
                                  uint32_t rot = 0u) {
    if (MODEL == 2)
        tile_stage_owner_user(p, mass, r, u0x, u0y, u0z, dst);
    else
        tile_stage_owner_m(p, mass, r, u0x, u0y, u0z, dst, rot);
}
template <int MODEL>
__device__ inline TileOwner tile_read(const uint4* sOwn, uint32_t at, uint32_t rot);  // `at`: slot x the launch's record stride (TileArgs::rs16)
__device__ inline TileOwner tile_read_owner(const uint4* sOwn, uint32_t at, uint32_t rot) {
    const uint4* q = sOwn + at + rot;
    const uint4 a = q[0], b = q[1], c = q[2], d = q[3], e = q[4], f = sOwn[at + (rot ? 0u : 5u)];
    TileOwner o;
    uint2 t;
    t = make_uint2(a.x, a.y), __builtin_memcpy(&o.px, &t, 8);
    t = make_uint2(a.z, a.w), __builtin_memcpy(&o.py, &t, 8);
    t = make_uint2(b.x, b.y), __builtin_memcpy(&o.pz, &t, 8);
    o.mass = __uint_as_float(b.z), o.family = b.w;
    o.R.xx = __uint_as_float(c.x), o.R.xy = __uint_as_float(c.y), o.R.xz = __uint_as_float(c.z), o.R.yx = __uint_as_float(c.w);
    o.R.yy = __uint_as_float(d.x), o.R.yz = __uint_as_float(d.y), o.R.zx = __uint_as_float(d.z), o.R.zy = __uint_as_float(d.w);
    o.R.zz = __uint_as_float(e.x), o.vx = __uint_as_float(e.y), o.vy = __uint_as_float(e.z), o.vz = __uint_as_float(e.w);
    o.wx = __uint_as_float(f.x), o.wy = __uint_as_float(f.y), o.wz = __uint_as_float(f.z);
    return o;
}
template <>
__device__ inline TileOwner tile_read<0>(const uint4* sOwn, uint32_t at, uint32_t rot) { return tile_read_owner(sOwn, at, rot); }
template <>
__device__ inline TileOwner tile_read<1>(const uint4* sOwn, uint32_t at, uint32_t rot) { return tile_read_owner(sOwn, at, rot); }
template <>
__device__ inline TileOwner tile_read<2>(const uint4* sOwn, uint32_t at, uint32_t) { return tile_read_owner_user(sOwn, at); }

// what only a run-time compiled model needs beside the staged records (MODEL 2; everything here that the user's statements do
// not name is dead code after inlining)
struct TileUser {
    double ox, oy, oz;        // the tile's origin in world coordinates: the vocabulary's positions are world positions
    uint32_t ownerA, ownerB;  // engine slots of the two owners
    uint32_t c;               // the contact's index in the list
    const uint64_t* keys;     // (AGeo, BGeo)
    float* const* ownerWc;
    float* const* geoWcSph;
    float* const* geoWcAnal;
    float time;
};

// what contact recording keeps per contact beside the force (deme_set_record_contacts; ContactInfoWriteBack.cu): the torque-only
// force and the contact point in the two owners' body frames
struct TileRecOut {
    f3 torqueOnly, locA, locB;
};

// One contact of the hot classes between two staged owners.  Arithmetic: forces_fast_body (deme_force_fast.h), with the
// owner-level quantities taken from the staged records.  Returns the world-frame force on A, the torques about A's and B's
// centres, and the updated history.
// The small tables a contact indexes per lane (components, material pairs, analytical objects, family margins) are copied to LDS
// by every workgroup: a global load in the contact loop makes its wait drain everything that was issued before it -- the vector
// memory counter retires in order -- and that is where the streams of the next rounds are in flight.
struct TileTables {
    const float4* comp;
    const MatPair* mat;
    const AnalObj* anal;
    const float* fam;   // family extra margins (read only when the scene has any)
    const float* mass;  // per mass-property entry
};
template <int MODEL>
__device__ inline void tile_contact(const DevParams& p, const TileTables& T, const uint2 inf, const TileOwner& A, const TileOwner& B,
                                    float4& hist, f3& force, f3& tA, f3& tB, float* uw = nullptr, const TileUser* U = nullptr,
                                    TileRecOut* ro = nullptr) {
    const uint32_t cls = (inf.x >> 20) & 3u;
    const float4 cA = T.comp[inf.y & 0xFFFFu];
    const uint32_t matA = (inf.x >> 24) & 15u;
    // sphere offsets with the reference's own rounding (no contraction: deme_device.h), see forces_fast_body
    const RotM &RA = A.R, &RB = B.R;
    const f3 relA = rot_apply(RA, mk3(cA.x, cA.y, cA.z));
    const float rA = cA.w;
#if DEME_TILE_KI & 1
    {
        float qx = cA.x, qy = cA.y;
        ki_opaque(qx), ki_opaque(qy);
        const f3 r2 = rot_apply(RA, mk3(qx, qy, cA.z)), r3 = rot_apply(RB, mk3(qy, qx, cA.z));
        const double e = (((A.px - B.px) + (double)r2.x) - (double)r3.x) + (((A.py - B.py) + (double)r2.y) - (double)r3.y) +
                         (((A.pz - B.pz) + (double)r2.z) - (double)r3.z);
        ki_sink((float)e);
    }
#endif
    float extraMargin = 0.f;
    if (!p.familyTrivial) {
        const float eA = T.fam[A.family & 0xFFu], eB = T.fam[B.family & 0xFFu];
        extraMargin = fmaxf(eA, eB);
    }
    const double dOx = A.px - B.px, dOy = A.py - B.py, dOz = A.pz - B.pz;
    const f3 dO = mk3((float)dOx, (float)dOy, (float)dOz);
    f3 n, rAv, rBv;
    f3 relBw = mk3(0, 0, 0);      // B's geometry relative to its owner, world frame (a run-time compiled model's bodyBPos)
    uint32_t userType = 1u;       // ... and its ContactType
    float depth, rB, massB;
    uint32_t matB;
    bool touching;
    if (cls == DEME_KEY_CLASS_SS) {
        const float4 cB = T.comp[inf.y >> 16];
        matB = inf.x >> 28;
        rB = cB.w;
        massB = B.mass;
        const f3 relB = rot_apply(RB, mk3(cB.x, cB.y, cB.z));
        relBw = relB;
        const double dx = (dOx + (double)relA.x) - (double)relB.x;
        const double dy = (dOy + (double)relA.y) - (double)relB.y;
        const double dz = (dOz + (double)relA.z) - (double)relB.z;
        const double d2 = dx * dx + dy * dy + dz * dz;
        // (the sum of the radii in fp64: rounded to fp32 it is off by up to 3e-8 of itself -- nothing for equal radii, whose sum is
        // exact, but 5e-10 m between unequal ones, 1e-5 of a typical overlap: found on the polydisperse bed of configs[4])
        const double sumRd = (double)rA + (double)rB;
        const float sumR = (float)sumRd;
        const double num = sumRd * sumRd - d2;
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
        const AnalObj ob = T.anal[inf.y >> 16];
        matB = ob.mat;
        rB = 1e15f;  // DEME_HUGE_FLOAT
        massB = ob.mass;
        const f3 relB = rot_apply(RB, mk3(ob.relx, ob.rely, ob.relz));
        const f3 dir = rot_apply(RB, mk3(ob.rotx, ob.roty, ob.rotz));
        relBw = relB;
        userType = ob.type == 0u ? 11u : 13u;
        if (ob.type == 0u) {  // a plane -- the walls of every box -- in a few lines (checkSphereEntityOverlap's plane branch,
            // DEMHelperKernels.cuh:459-478): half of a packed bed's tiles touch the floor, and the general branch below is 150
            // instructions that every wavefront with one wall contact among its 64 would execute
            const double px = (dOx + (double)relA.x) - (double)relB.x, py = (dOy + (double)relA.y) - (double)relB.y,
                         pz = (dOz + (double)relA.z) - (double)relB.z;
            const float dist = (float)(px * (double)dir.x + py * (double)dir.y + pz * (double)dir.z);
            const double dd = (double)rA - (double)dist;
            depth = (float)dd;
            touching = !(dd < -(double)extraMargin);
            const float sOff = (float)((double)dist + dd / 2.0);
            n = dir;
            rAv = mk3(relA.x - dir.x * sOff, relA.y - dir.y * sOff, relA.z - dir.z * sOff);  // contact point = sphere centre - s n
            rBv = fadd(rAv, dO);
        } else {
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
    }
    force = mk3(0, 0, 0);
    f3 torque_only = mk3(0, 0, 0);
#ifdef DEME_JIT
    if (MODEL == 2) {  // the user's statements, written against the reference's vocabulary (DEMCalcForceKernels.cu:233-251, Models.h:219-378)
        if (touching) {
            UserModelIO io;
            io.overlapDepth = (double)depth;
            io.B2A = make_float3(n.x, n.y, n.z);
            io.AOwnerPos = make_double3(U->ox + A.px, U->oy + A.py, U->oz + A.pz);
            io.BOwnerPos = make_double3(U->ox + B.px, U->oy + B.py, U->oz + B.pz);
            io.contactPnt = make_double3(io.AOwnerPos.x + (double)rAv.x, io.AOwnerPos.y + (double)rAv.y, io.AOwnerPos.z + (double)rAv.z);
            io.bodyAPos = make_double3(io.AOwnerPos.x + (double)relA.x, io.AOwnerPos.y + (double)relA.y, io.AOwnerPos.z + (double)relA.z);
            io.bodyBPos = make_double3(io.BOwnerPos.x + (double)relBw.x, io.BOwnerPos.y + (double)relBw.y, io.BOwnerPos.z + (double)relBw.z);
            io.AOwnerMass = A.mass, io.BOwnerMass = massB, io.ARadius = rA, io.BRadius = rB;
            io.AOriQ = make_float4(A.qx, A.qy, A.qz, A.qw), io.BOriQ = make_float4(B.qx, B.qy, B.qz, B.qw);  // float4 is (x, y, z, w)
            io.bodyAMatType = (uint16_t)matA, io.bodyBMatType = (uint16_t)matB;
            io.ContactType = (uint8_t)userType, io.AOwnerFamily = (uint8_t)(A.family & 0xFFu), io.BOwnerFamily = (uint8_t)(B.family & 0xFFu);
            const f3 lA = rot_apply(rot_transpose(RA), rAv), lB = rot_apply(rot_transpose(RB), rBv);
            io.locCPA = make_float3(lA.x, lA.y, lA.z), io.locCPB = make_float3(lB.x, lB.y, lB.z);
            io.force = make_float3(0, 0, 0), io.torque_only_force = make_float3(0, 0, 0);
            io.ts = p.h, io.time = U->time;
            io.ALinVel = make_float3(A.vx, A.vy, A.vz), io.BLinVel = make_float3(B.vx, B.vy, B.vz);
            io.ARotVel = make_float3(A.wx, A.wy, A.wz), io.BRotVel = make_float3(B.wx, B.wy, B.wz);
            {
                const float4 mA = p.massProps[A.family >> 16], mB = p.massProps[B.family >> 16];
                io.AOwnerMOI = make_float3(mA.y, mA.z, mA.w), io.BOwnerMOI = make_float3(mB.y, mB.z, mB.w);
            }
            io.AOwner = p.o2e ? p.o2e[U->ownerA] : U->ownerA, io.BOwner = p.o2e ? p.o2e[U->ownerB] : U->ownerB;
            io.myContactID = U->c;
            {
                const uint64_t key = U->keys[U->c];
                const uint32_t ga = key_a(key), gb = key_b(key);
                io.AGeo = p.s2e ? p.s2e[ga] : ga, io.BGeo = (p.s2e && cls == DEME_KEY_CLASS_SS) ? p.s2e[gb] : gb;
            }
            io.wc = uw;
            io.ownerWc = U->ownerWc;
            io.geoWcA = U->geoWcSph;
            io.geoWcB = (cls == DEME_KEY_CLASS_SS) ? U->geoWcSph : U->geoWcAnal;
            deme_user_model(io);
            force = mk3(io.force.x, io.force.y, io.force.z);
            const f3 tot = fadd(force, mk3(io.torque_only_force.x, io.torque_only_force.y, io.torque_only_force.z));
            tA = fcross(rAv, tot);
            tB = fcross(tot, rBv);
        } else {
            tA = mk3(0, 0, 0), tB = mk3(0, 0, 0);
#pragma unroll
            for (int k = 0; k < DEME_JIT_NW; k++)
                uw[k] = 0.f;  // _forceModelContactWildcardDestroy_
        }
        return;
    }
#endif
    if (touching) {
        if (depth > 0.f) {
            const float massA = A.mass;
            const MatPair mp = T.mat[matA * p.nMat + matB];
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
        if (ro) {
            ro->torqueOnly = torque_only;
            ro->locA = rot_apply(rot_transpose(RA), rAv), ro->locB = rot_apply(rot_transpose(RB), rBv);
        }
    } else {
        tA = mk3(0, 0, 0), tB = mk3(0, 0, 0);
        hist = make_float4(0, 0, 0, 0);  // _forceModelContactWildcardDestroy_
        if (ro)
            ro->torqueOnly = mk3(0, 0, 0), ro->locA = mk3(0, 0, 0), ro->locB = mk3(0, 0, 0);
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

#ifndef DEME_TILE_OCC
#define DEME_TILE_OCC 1
#endif
#ifndef DEME_TILE_UNROLL
#define DEME_TILE_UNROLL 0  // 1: the rounds unrolled DEME_TILE_DEPTH at a time, no rotation of the stream registers (measured: no gain, 3x the code)
#endif
#ifndef DEME_TILE_PKPULL
#define DEME_TILE_PKPULL 1  // the pulls add register pairs (v_pk_add_f32); a missing entry reads the zero slot
#endif
#ifndef DEME_TILE_SPLITLOOP
#define DEME_TILE_SPLITLOOP 0  // 1: a tile of at most DEME_TILE_DEPTH rounds (its streams were all asked for in the prologue) runs a copy of the round
                               // loop without refills: no load is in flight there, so the rotation of the stream registers waits for nothing --
                               // in the refilling loop the compiler's s_waitcnt vmcnt(0) in front of the rotation also drains the round's STORES
#endif
#ifndef DEME_TILE_PRIO_ROUNDS
#define DEME_TILE_PRIO_ROUNDS 0  // wavefront priority of the rounds (the prologue runs at DEME_TILE_PRIO)
#endif
#ifndef DEME_TILE_PRIO_PULL
#define DEME_TILE_PRIO_PULL 0  // > 0: wavefront priority between a round's two barriers (the pulls: the whole workgroup waits for them)
#endif
#ifndef DEME_TILE_PULLW
#define DEME_TILE_PULLW 4  // entries an owner thread pulls per trip (a missing entry reads the zero slot)
#endif
#ifndef DEME_TILE_NOZERO
#define DEME_TILE_NOZERO 1  // 1: no zero defaults for values that are only read where they were set (refilled stream stages, crossing records)
#endif
#ifndef DEME_TILE_PRIO
#define DEME_TILE_PRIO 0  // > 0: wavefront priority of the prologue (ids -> records -> staging); the rounds run at 0
#endif
#ifndef DEME_TILE_DEPTH
#define DEME_TILE_DEPTH 3  // rounds whose streams are in flight (tInfo + history: 24 bytes per thread and round)
#endif
// LDS of one tile, laid out at launch from the list's own extremes (TileArgs::hCap, lCap): occupancy is bound by LDS here, and
// the compile-time capacities (DEME_TILE_HMAX foreign owners, DEME_TILE_LMAX list entries) are twice what a packed bed needs
//   own   [(NB + hCap) x 96 B]   staged owners
//   recA4 [(T + 2) x 16 B]  recT [(T + 2) x 16 B]  recA2 [(T + 2) x 8 B]   a round's contributions: (F.x F.y F.z tA.x) (tB.y tB.z tB.x -)
//                                                        (tA.y tA.z); B's side is -F and tB; slot T of each array stays zero
//   aLo, lLo [(NB + 1) x 4 B each]   owner o's A run = positions [aLo[o], aLo[o + 1]) of the tile's range, its local-B list =
//                                    lPos[lLo[o] .. lLo[o + 1])
//   lPos  [lCap x 2 B]
//   tables: components [nComp x 16 B], material pairs [nMat^2 x 32 B], analytical objects [nAnal x 64 B], masses [nMass x 4 B],
//           family margins [256 x 4 B, only when a family has one]
#define DEME_TILE_BOUNDS_BYTES ((2u * (DEME_TILE_NB + 1u) * 2u + 15u) & ~15u)  // the owners' run bounds: two 16-bit arrays of NB + 1
#define DEME_TILE_RSLOTS (DEME_TILE_T + 2)  // a round's contribution slots + the ZERO slot (index DEME_TILE_T) the pulls read for "no entry"
                                            // (+ 1: the arrays behind stay 16-byte aligned)
#define DEME_TILE_TABLE_MAX 4096u  // bytes of tables a scene may have and still take the tile path
__host__ __device__ inline uint32_t tile_table_bytes(uint32_t nComp, uint32_t nMat, uint32_t nAnal, uint32_t nMass, uint32_t famTrivial) {
    return nComp * 16u + nMat * nMat * 32u + nAnal * 64u + ((nMass * 4u + 15u) & ~15u) + (famTrivial ? 0u : 1024u);
}
__host__ __device__ inline uint32_t tile_lds_bytes(uint32_t hCap, uint32_t lCap, uint32_t tableBytes, uint32_t rec16 = DEME_TILE_REC) {
    return (DEME_TILE_NB + hCap) * rec16 * 16u + DEME_TILE_RSLOTS * 40u + DEME_TILE_BOUNDS_BYTES + ((lCap * 2u + 15u) & ~15u) +
           tableBytes + 16u;
}

template <bool B>
struct TileFlag {
    static constexpr bool value = B;
};

template <int MODEL, bool MESH, bool REC = false>
__global__ __launch_bounds__(DEME_TILE_T, DEME_TILE_OCC) void k_tile_forces(const DevParams p, const TileArgs a) {
    extern __shared__ uint4 tileLds[];
#if DEME_TILE_PRIO
    __builtin_amdgcn_s_setprio(DEME_TILE_PRIO);
#endif
    uint4* const sOwn = tileLds;
    const uint32_t RSZ = a.rs16;  // uint4 per staged record: what the record holds, or one more (see TileArgs::rs16)
    float4* const recA4 = reinterpret_cast<float4*>(sOwn + (DEME_TILE_NB + a.hCap) * RSZ);
    float4* const recT = recA4 + DEME_TILE_RSLOTS;
    float2* const recA2 = reinterpret_cast<float2*>(recT + DEME_TILE_RSLOTS);
    uint16_t* const sALo = reinterpret_cast<uint16_t*>(recA2 + DEME_TILE_RSLOTS);  // (a tile's contact range is at most DEME_TILE_CMAX long)
    uint16_t* const sLLo = sALo + (DEME_TILE_NB + 1);
    uint16_t* const sLPos = sALo + DEME_TILE_BOUNDS_BYTES / 2u;  // (16-byte aligned)
    const uint32_t t = tile_block_id(a.xcdGroup);
    if (t >= a.nTiles)
        return;
    // Whether this tile is evaluated here at all -- the pass of a split step, a tile left to k_tile_forces_big -- is decided in FRONT of
    // the tile's loads.  (DEME_TILE_LATE_SKIP=1 decides it behind the first loads: two scalar round trips off every tile's start --
    // measured: 88.9 against 89.0 us, nothing, and the ghost-dependent pass of a slab step, which leaves 95 % of its tiles at once,
    // would fetch every tile's records first.)
#ifndef DEME_TILE_LATE_SKIP
#define DEME_TILE_LATE_SKIP 0
#endif
#if DEME_TILE_LATE_SKIP
    const uint32_t skipBig = a.tileBig[t];
    const uint32_t skipMode = a.tileMode ? a.tileMode[t] : 0xFFFFFFFFu;
#else
    if (a.tileMode && !(a.tileMode[t] & (1u << a.pass)))
        return;
    if (a.tileBig[t])
        return;
#endif
    const uint32_t tid = threadIdx.x;
    const uint32_t o0 = t * DEME_TILE_NB;
    const uint32_t nLoc = min((uint32_t)DEME_TILE_NB, a.nOwners - o0);
    TILE_STAMP(0);
#if DEME_TILE_STAMPS
    const unsigned long long stampClk0 = (unsigned long long)clock64();  // (s_memtime: shader cycles; word 15 = cycles over the tile's life)
#endif
    // ---- every load that needs nothing but the tile number goes out first: the ids of the foreign owners behind my staging slots
    // (the list is padded to DEME_TILE_HMAX per tile: reading past the tile's own count is harmless), my local owner's record,
    // my entries of the small tables.  A tile's lifetime is a chain of memory latencies; this makes it two deep
    // (ids -> foreign records), with the scalars -> streams chain beside it.
    const uint32_t* hl = a.hList + (size_t)t * DEME_TILE_HMAX;
#if DEME_TILE_KI & 128  // knock-in: one more level in the chain of dependent loads (a vector load in front of the ids)
    {
        uint32_t zz = a.rankC[t];
        asm volatile("v_and_b32 %0, 0, %0" : "+v"(zz));
        hl += zz;
    }
#endif
    const uint32_t h0 = tid - nLoc, h1 = tid + DEME_TILE_T - nLoc;  // my foreign slots (meaningful when < nH)
    uint32_t id0 = (tid >= nLoc && h0 < DEME_TILE_HMAX) ? hl[h0] : 0u;
    uint32_t id1 = (h1 < DEME_TILE_HMAX) ? hl[h1] : 0u;
    OwnerRec rec0;
    if (tid < nLoc)
        rec0 = load_owner(a.owners, o0 + tid);
    TileTables T;
    uint4* const tabBase = reinterpret_cast<uint4*>(reinterpret_cast<char*>(sLPos) + ((a.lCap * 2u + 15u) & ~15u));
    const uint32_t nTab16 = a.nComp + p.nMat * p.nMat * 2u + a.nAnal * 4u;  // 16-byte pieces of the first three tables
    {
        float4* sComp = reinterpret_cast<float4*>(tabBase);
        MatPair* sMat = reinterpret_cast<MatPair*>(sComp + a.nComp);
        AnalObj* sAnal = reinterpret_cast<AnalObj*>(sMat + p.nMat * p.nMat);
        float* sMass = reinterpret_cast<float*>(sAnal + a.nAnal);
        float* sFam = sMass + ((a.nMass + 3u) & ~3u);
        T.comp = sComp, T.mat = sMat, T.anal = sAnal, T.mass = sMass, T.fam = sFam;
    }
    uint4 tab16 = make_uint4(0, 0, 0, 0);
    float tabMass = 0.f, tabFam = 0.f;
    if (tid < nTab16) {
        const uint32_t k = tid;
        tab16 = k < a.nComp ? reinterpret_cast<const uint4*>(p.comp)[k]
                : (k < a.nComp + p.nMat * p.nMat * 2u ? reinterpret_cast<const uint4*>(p.matPair)[k - a.nComp]
                                                       : reinterpret_cast<const uint4*>(p.anal)[k - a.nComp - p.nMat * p.nMat * 2u]);
    }
    if (tid < a.nMass)
        tabMass = p.massProps[tid].x;
    if (!p.familyTrivial)
        tabFam = p.familyExtra[tid & 255u];
    // ---- the scalars of the tile, then what hangs on them: the streams of the first rounds (a tile is ~2 rounds; memory latency
    // under load is longer than a round takes), the run bounds, the local-B lists
    const uint32_t nH = a.hCount[t];
    const uint32_t c0 = a.aStart[o0], c1 = a.aStart[o0 + nLoc];
    const uint16_t* const lOffT = a.lOff + (size_t)t * (DEME_TILE_NB + 1);
    const uint32_t nL = a.lCount[t];  // (a 32-bit word: a scalar load -- a 16-bit one would be a vector load, whose wait drains the loads issued before it)
    const int64_t u0x = a.org[3 * (size_t)t], u0y = a.org[3 * (size_t)t + 1], u0z = a.org[3 * (size_t)t + 2];
#if DEME_TILE_LATE_SKIP
    if (skipBig || !(skipMode & (1u << a.pass)))
        return;
#endif
    const float4* wc4 = reinterpret_cast<const float4*>(a.wc);
    uint2 inf[DEME_TILE_DEPTH];
    float4 hist[DEME_TILE_DEPTH];
    constexpr int NWU = MODEL == 2 ? DEME_JIT_NW : 1;
    float uwv[DEME_TILE_DEPTH][NWU];  // MODEL 2: the contact wildcards of the user's model ride where the history does
    uint32_t rbase[DEME_TILE_DEPTH];  // record number of the first crossing contact of this wavefront's 64 contacts
#pragma unroll
    for (int d = 0; d < DEME_TILE_DEPTH; d++) {
        const uint32_t cd = c0 + tid + d * DEME_TILE_T;
        inf[d] = make_uint2(0, 0), hist[d] = make_float4(0, 0, 0, 0), rbase[d] = 0u;
#pragma unroll
        for (int k = 0; k < NWU; k++)
            uwv[d][k] = 0.f;
        if (cd < c1) {
            inf[d] = stream_load(a.tInfo + cd);
            if (MODEL == 0)
                hist[d] = stream_load(wc4 + cd);
            if (MODEL == 2 && DEME_JIT_HAS_WC) {
#pragma unroll
                for (int k = 0; k < NWU; k++)
                    uwv[d][k] = stream_load(a.wc + (size_t)cd * NWU + k);
            }
            rbase[d] = a.rankC[cd - (tid & 63u)];
        }
    }
    uint32_t bA = 0, bL = 0;
    if (tid <= DEME_TILE_NB) {
        const uint32_t o = min(tid, nLoc);
        bA = a.aStart[o0 + o], bL = lOffT[o];
    }
    uint32_t lp[DEME_TILE_LREG];
#pragma unroll
    for (int k = 0; k < DEME_TILE_LREG; k++) {
        const uint32_t i = tid + k * DEME_TILE_T;
        lp[k] = (i < nL) ? (uint32_t)a.lPos[c0 + i] : 0u;
    }
    // ---- the foreign owners' records (their ids have arrived by now), the tables into LDS
    OwnerRec rec1;
    if (tid >= nLoc && h0 < nH)
        rec0 = load_owner(a.owners, id0);
    if (h1 < nH)
        rec1 = load_owner(a.owners, id1);
    // (the touches go out with the foreign records -- the second level of this tile's own chain: they come back with them, and
    // nothing in front of the rounds waits for a load of its own)
    if (tid < nTab16)
        tabBase[tid] = tab16;
    if (tid < a.nMass)
        const_cast<float*>(T.mass)[tid] = tabMass;
    if (!p.familyTrivial)
        const_cast<float*>(T.fam)[tid & 255u] = tabFam;
    __syncthreads();  // (the staging below reads the masses)
    TILE_STAMP(1);
    {   // stage the tile's owners, its halo, the owners' run bounds and local-B lists
#if DEME_TILE_KI & 4
        if (tid < nLoc + nH) {
            OwnerRec r2 = rec0;
            ki_opaque(r2.qw), ki_opaque(r2.wx), ki_opaque(r2.family);
            tile_stage<MODEL>(p, T.mass[r2.inertiaOff], r2, u0x, u0y, u0z, sOwn + (tid < nLoc ? tid : DEME_TILE_NB + h0) * RSZ, ((tid < nLoc ? tid : DEME_TILE_NB + h0) >> 3) & a.swz);
        }
        if (h1 < nH) {
            OwnerRec r2 = rec1;
            ki_opaque(r2.qw), ki_opaque(r2.wx), ki_opaque(r2.family);
            tile_stage<MODEL>(p, T.mass[r2.inertiaOff], r2, u0x, u0y, u0z, sOwn + (DEME_TILE_NB + h1) * RSZ, ((DEME_TILE_NB + h1) >> 3) & a.swz);
        }
#endif
        if (tid < nLoc + nH)
            tile_stage<MODEL>(p, T.mass[rec0.inertiaOff], rec0, u0x, u0y, u0z, sOwn + (tid < nLoc ? tid : DEME_TILE_NB + h0) * RSZ, ((tid < nLoc ? tid : DEME_TILE_NB + h0) >> 3) & a.swz);
        if (h1 < nH)
            tile_stage<MODEL>(p, T.mass[rec1.inertiaOff], rec1, u0x, u0y, u0z, sOwn + (DEME_TILE_NB + h1) * RSZ, ((DEME_TILE_NB + h1) >> 3) & a.swz);
        if (tid <= DEME_TILE_NB)
            sALo[tid] = (uint16_t)(bA - c0), sLLo[tid] = (uint16_t)bL;
        if (tid == DEME_TILE_T - 1u)  // the zero slot of the contribution arrays
            recA4[DEME_TILE_T] = make_float4(0, 0, 0, 0), recT[DEME_TILE_T] = make_float4(0, 0, 0, 0), recA2[DEME_TILE_T] = make_float2(0, 0);
#pragma unroll
        for (int k = 0; k < DEME_TILE_LREG; k++)
            if (tid + k * DEME_TILE_T < nL)
                sLPos[tid + k * DEME_TILE_T] = (uint16_t)lp[k];
    }
    __syncthreads();
    TILE_STAMP(2);
#if DEME_TILE_PRIO || DEME_TILE_PRIO_ROUNDS
    __builtin_amdgcn_s_setprio(DEME_TILE_PRIO_ROUNDS);
#endif

    // the pulling side of this thread: threads 0 .. NB - 1 take the A runs, NB .. 2 NB - 1 the local-B lists
    const uint32_t po = tid % DEME_TILE_NB;
    const bool sideA = tid < DEME_TILE_NB, sideB = !sideA && tid < 2 * DEME_TILE_NB;
    uint32_t plo = sideB ? sLLo[po] : sALo[po];
    const uint32_t phi = (sideA || sideB) ? (sideB ? sLLo[po + 1] : sALo[po + 1]) : plo;
    // the six sums of this thread's owner and side as three register pairs (the pulls add with v_pk_add_f32):
    //   A side: (F.x F.y) (F.z tA.x) (tA.y tA.z);  B side: (-F.x -F.y) (-F.z tB.x) (tB.y tB.z)
    v2f s01 = {0.f, 0.f}, s23 = {0.f, 0.f}, s45 = {0.f, 0.f};
    const uint32_t nCt = c1 - c0;
    auto refill = [&](const int d, const uint32_t c) __attribute__((always_inline)) {  // stage d takes the round DEME_TILE_DEPTH rounds ahead of contact c
        const uint32_t cd = c + DEME_TILE_DEPTH * DEME_TILE_T;
#if !DEME_TILE_NOZERO
        inf[d] = make_uint2(0, 0), hist[d] = make_float4(0, 0, 0, 0), rbase[d] = 0u;
#pragma unroll
        for (int k = 0; k < NWU; k++)
            uwv[d][k] = 0.f;
#endif
        if (cd < c1) {
            inf[d] = stream_load(a.tInfo + cd);
            if (MODEL == 0)
                hist[d] = stream_load(wc4 + cd);
            if (MODEL == 2 && DEME_JIT_HAS_WC) {
#pragma unroll
                for (int k = 0; k < NWU; k++)
                    uwv[d][k] = stream_load(a.wc + (size_t)cd * NWU + k);
            }
            rbase[d] = a.rankC[cd - (tid & 63u)];
        }
    };
    auto round = [&](auto RF, const int d, const uint32_t rlo) __attribute__((always_inline)) {
            const uint32_t c = c0 + rlo + tid;
            bool crossing = false;
#if DEME_TILE_NOZERO
            float4 x4, x2;  // (read only where `crossing` was set)
#else
            float4 x4 = make_float4(0, 0, 0, 0), x2 = x4;
#endif
            if (c < c1) {
                const uint2 ci = inf[d];
                float4 h = hist[d];
                const uint32_t slotA = ci.x & 1023u, slotB = (ci.x >> 10) & 1023u;
                const TileOwner A = tile_read<MODEL>(sOwn, slotA * RSZ, (slotA >> 3) & a.swz), B = tile_read<MODEL>(sOwn, slotB * RSZ, (slotB >> 3) & a.swz);
                f3 force, tA, tB;
                if (MESH && ((ci.x >> 20) & 3u) == DEME_KEY_CLASS_SM) {  // (rare, and only in tiles along the mesh: the loads sit behind a branch)
                    const float4 a4 = a.conA4[c], b4 = a.conB4[c];
                    const float2 a2 = a.conA2[c], b2 = a.conB2[c];
                    force = mk3(a4.x, a4.y, a4.z), tA = mk3(a4.w, a2.x, a2.y), tB = mk3(b4.w, b2.x, b2.y);
                } else {
#if DEME_TILE_KI & 8
                    {
                        uint2 ci2 = ci;
                        ki_opaque(ci2.x), ki_opaque(ci2.y);
                        float4 h2 = h;
                        f3 f2, u2, w2;
                        tile_contact<MODEL>(p, T, ci2, A, B, h2, f2, u2, w2);
                        ki_sink(f2.x + f2.y + f2.z + u2.x + u2.y + u2.z + w2.x + w2.y + w2.z + h2.x + h2.y + h2.z + h2.w);
                    }
#endif
                    if (MODEL == 2) {
                        float uw[NWU];
#pragma unroll
                        for (int k = 0; k < NWU; k++)
                            uw[k] = uwv[d][k];
                        TileUser U;
                        U.ox = (double)u0x * p.l + (double)p.LBFX, U.oy = (double)u0y * p.l + (double)p.LBFY, U.oz = (double)u0z * p.l + (double)p.LBFZ;
                        U.ownerA = o0 + slotA;
                        U.ownerB = slotB < DEME_TILE_NB ? o0 + slotB : hl[slotB - DEME_TILE_NB];
                        U.c = c, U.keys = a.keys, U.ownerWc = a.ownerWc, U.geoWcSph = a.geoWcSph, U.geoWcAnal = a.geoWcAnal, U.time = a.timeElapsed;
                        tile_contact<MODEL>(p, T, ci, A, B, h, force, tA, tB, uw, &U);
                        if (DEME_JIT_HAS_WC) {
#pragma unroll
                            for (int k = 0; k < NWU; k++)
                                stream_store(a.wc + (size_t)c * NWU + k, uw[k]);
                        }
                    } else if (REC) {  // the script wants per-contact forces and contact points (the reference's default contact output)
                        TileRecOut ro;
                        tile_contact<MODEL>(p, T, ci, A, B, h, force, tA, tB, nullptr, nullptr, &ro);
                        float* r = a.rec[0] + 3ull * c;
                        r[0] = force.x, r[1] = force.y, r[2] = force.z;
                        r = a.rec[1] + 3ull * c;
                        r[0] = ro.torqueOnly.x, r[1] = ro.torqueOnly.y, r[2] = ro.torqueOnly.z;
                        r = a.rec[2] + 3ull * c;
                        r[0] = ro.locA.x, r[1] = ro.locA.y, r[2] = ro.locA.z;
                        r = a.rec[3] + 3ull * c;
                        r[0] = ro.locB.x, r[1] = ro.locB.y, r[2] = ro.locB.z;
                    } else {
                        tile_contact<MODEL>(p, T, ci, A, B, h, force, tA, tB);
                    }
#if !(DEME_TILE_KI & 16)
                    if (MODEL == 0)
                        stream_store(reinterpret_cast<float4*>(a.wc) + c, h);
#endif
                }
                recA4[tid] = make_float4(force.x, force.y, force.z, tA.x);
                recA2[tid] = make_float2(tA.y, tA.z);
                if (slotB < DEME_TILE_NB) {
                    recT[tid] = make_float4(tB.y, tB.z, tB.x, 0.f);
                } else if (ci.x & (1u << 22)) {
                    crossing = true;
                    x4 = make_float4(-force.x, -force.y, -force.z, tB.x), x2 = make_float4(tB.y, tB.z, 0.f, 0.f);
                }
            }
            {   // the wavefront's crossing contacts write consecutive records
                const uint64_t m = __ballot(crossing);
                if (crossing) {
                    const uint32_t k = rbase[d] + (uint32_t)__popcll(m & ((1ull << (tid & 63u)) - 1ull));
#if !(DEME_TILE_KI & 32)
#if DEME_REC24
                    float2* const r24 = reinterpret_cast<float2*>(a.rec32) + 3 * (size_t)k;  // (-F.x -F.y) (-F.z tB.x) (tB.y tB.z)
                    stream_store(r24, make_float2(x4.x, x4.y));
                    stream_store(r24 + 1, make_float2(x4.z, x4.w));
                    stream_store(r24 + 2, make_float2(x2.x, x2.y));
#else
                    stream_store(a.rec32 + 2 * (size_t)k, x4);
                    stream_store(a.rec32 + 2 * (size_t)k + 1, x2);
#endif
#else
                    ki_sink(x4.x + x2.x + (float)k);
#endif
                }
            }
            
#if DEME_TILE_UNROLL
            refill(d, c);
#else
#pragma unroll
            for (int q = 0; q + 1 < DEME_TILE_DEPTH; q++) {
                inf[q] = inf[q + 1], hist[q] = hist[q + 1], rbase[q] = rbase[q + 1];
#pragma unroll
                for (int k = 0; k < NWU; k++)
                    uwv[q][k] = uwv[q + 1][k];
            }
            if (decltype(RF)::value)
                refill(DEME_TILE_DEPTH - 1, c);
#endif

            __syncthreads();
#if DEME_TILE_KI & 256  // knock-in: one more barrier per round
            asm volatile("s_nop 0" ::: "memory");
            __syncthreads();
#endif
#if DEME_TILE_PRIO_PULL
            __builtin_amdgcn_s_setprio(DEME_TILE_PRIO_PULL);
#endif
            const uint32_t rhi = rlo + DEME_TILE_T;
            auto pulls = [&]() {
                if (sideA) {  // my A run's part of this round: positions [plo, min(phi, rhi)); a missing entry reads the zero slot
                    const uint32_t e = min(phi, rhi);
                    while (plo < e) {
                        float4 v4[DEME_TILE_PULLW];
                        float2 v2[DEME_TILE_PULLW];
#if DEME_TILE_PKPULL
#pragma unroll
                        for (int k = 0; k < DEME_TILE_PULLW; k++) {
                            const uint32_t i = (plo + k < e) ? plo + k - rlo : (uint32_t)DEME_TILE_T;
                            v4[k] = recA4[i], v2[k] = recA2[i];
                        }
#pragma unroll
                        for (int k = 0; k < DEME_TILE_PULLW; k++) {
                            s01 += v2f{v4[k].x, v4[k].y};
                            s23 += v2f{v4[k].z, v4[k].w};
                            s45 += v2f{v2[k].x, v2[k].y};
                        }
#else
#pragma unroll
                        for (int k = 0; k < DEME_TILE_PULLW; k++) {
                            const uint32_t i = min(plo + k, e - 1u) - rlo;
                            v4[k] = recA4[i], v2[k] = recA2[i];
                        }
#pragma unroll
                        for (int k = 0; k < DEME_TILE_PULLW; k++)
                            if (plo + k < e)
                                s01.x += v4[k].x, s01.y += v4[k].y, s23.x += v4[k].z, s23.y += v4[k].w, s45.x += v2[k].x, s45.y += v2[k].y;
#endif
                        plo = min(plo + (uint32_t)DEME_TILE_PULLW, e);
                    }
                } else if (sideB) {  // my local-B list's entries that fall into this round: -F and tB of those contacts
                    while (plo < phi) {
                        uint32_t pos[DEME_TILE_PULLW];
#pragma unroll
                        for (int k = 0; k < DEME_TILE_PULLW; k++)
                            pos[k] = (plo + k < phi) ? (uint32_t)sLPos[plo + k] : 0xFFFFFFFFu;
                        if (pos[0] >= rhi)
                            break;
                        float4 v4[DEME_TILE_PULLW], vt[DEME_TILE_PULLW];
                        uint32_t used = 0;
#if DEME_TILE_PKPULL
#pragma unroll
                        for (int k = 0; k < DEME_TILE_PULLW; k++) {
                            const bool in = pos[k] < rhi;  // (ascending: the entries of this round come first)
                            const uint32_t i = in ? pos[k] - rlo : (uint32_t)DEME_TILE_T;
                            v4[k] = recA4[i], vt[k] = recT[i];
                            used += in ? 1u : 0u;
                        }
#pragma unroll
                        for (int k = 0; k < DEME_TILE_PULLW; k++) {
                            s01 -= v2f{v4[k].x, v4[k].y};
                            s23.x -= v4[k].z;
                            s23.y += vt[k].z;
                            s45 += v2f{vt[k].x, vt[k].y};
                        }
#else
#pragma unroll
                        for (int k = 0; k < DEME_TILE_PULLW; k++) {
                            const uint32_t i = (pos[k] < rhi ? pos[k] : pos[0]) - rlo;
                            v4[k] = recA4[i], vt[k] = recT[i];
                        }
#pragma unroll
                        for (int k = 0; k < DEME_TILE_PULLW; k++)
                            if (pos[k] < rhi) {
                                s01.x -= v4[k].x, s01.y -= v4[k].y, s23.x -= v4[k].z, s23.y += vt[k].z, s45.x += vt[k].x, s45.y += vt[k].y;
                                used++;
                            }
#endif
                        plo += used;
                        if (used < (uint32_t)DEME_TILE_PULLW)
                            break;
                    }
                }
            };
#if DEME_TILE_KI & 2
            {
                const v2f k01 = s01, k23 = s23, k45 = s45;
                uint32_t kp = plo;
                pulls();
                ki_sink(s01.x + s01.y + s23.x + s23.y + s45.x + s45.y);
                ki_opaque(kp);
                s01 = k01, s23 = k23, s45 = k45, plo = kp;
            }
#endif
            pulls();
#if DEME_TILE_PRIO_PULL
            __builtin_amdgcn_s_setprio(DEME_TILE_PRIO_ROUNDS);
#endif
            __syncthreads();
            TILE_STAMP(min(3u + rlo / DEME_TILE_T, 10u));
    };
#if DEME_TILE_UNROLL
    // The rounds, DEME_TILE_DEPTH at a time with the stage of the stream registers a round uses fixed at compile time: nothing is
    // rotated between rounds
    for (uint32_t rb = 0; rb < nCt; rb += DEME_TILE_DEPTH * DEME_TILE_T) {
#pragma unroll
        for (int d = 0; d < DEME_TILE_DEPTH; d++) {
            const uint32_t rlo = rb + d * DEME_TILE_T;
            if (rlo >= nCt)  // (uniform over the workgroup)
                break;
            round(TileFlag<true>{}, d, rlo);
        }
    }
#else
#if DEME_TILE_SPLITLOOP
    if (nCt <= DEME_TILE_DEPTH * DEME_TILE_T) {
        for (uint32_t rlo = 0; rlo < nCt; rlo += DEME_TILE_T)
            round(TileFlag<false>{}, 0, rlo);
    } else
#endif
    for (uint32_t rlo = 0; rlo < nCt; rlo += DEME_TILE_T)  // stage 0 is the current round; the stages are rotated after it
        round(TileFlag<true>{}, 0, rlo);
#endif
    // A-side sum + B-side sum, through LDS (the contribution arrays are free now)
    if (sideB) {
        recA4[po] = make_float4(s01.x, s01.y, s23.x, s23.y);
        recA2[po] = make_float2(s45.x, s45.y);
    }
    __syncthreads();
    if (sideA && po < nLoc) {
        const float4 b4 = recA4[po];
        const float2 b2 = recA2[po];
#if !(DEME_TILE_KI & 64)
#if DEME_REC24
        float2* const t24 = reinterpret_cast<float2*>(a.tSum) + 3 * (size_t)(o0 + po);  // (F.x F.y) (F.z t.x) (t.y t.z)
        t24[0] = make_float2(s01.x + b4.x, s01.y + b4.y);
        t24[1] = make_float2(s23.x + b4.z, s23.y + b4.w);
        t24[2] = make_float2(s45.x + b2.x, s45.y + b2.y);
#else
        a.tSum[2 * (size_t)(o0 + po)] = make_float4(s01.x + b4.x, s01.y + b4.y, s23.x + b4.z, 0.f);
        a.tSum[2 * (size_t)(o0 + po) + 1] = make_float4(s23.y + b4.w, s45.x + b2.x, s45.y + b2.y, 0.f);
#endif
#else
        ki_sink(s01.x + b4.x + s01.y + b4.y + s23.x + b4.z + s23.y + b4.w + s45.x + b2.x + s45.y + b2.y);
#endif
    }
#if DEME_TILE_STAMPS
    TILE_STAMP(11);
    if (a.stamps && tid == 0) {
        uint32_t hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        a.stamps[(size_t)t * 16u + 12u] = (unsigned long long)hw | ((unsigned long long)xcc << 32);
        a.stamps[(size_t)t * 16u + 13u] = nCt;
        a.stamps[(size_t)t * 16u + 14u] = nH;
        a.stamps[(size_t)t * 16u + 15u] = (unsigned long long)clock64() - stampClk0;
    }
#endif
}

// ---- the tiles that do not fit ---------------------------------------------------------------------------------------------------
// A tile whose foreign owners exceed DEME_TILE_HMAX (a big sphere with hundreds of neighbours, a polydisperse pocket, a numbering
// that is not spatial), whose contact range exceeds DEME_TILE_CMAX or whose local-B lists exceed DEME_TILE_LMAX is evaluated by this
// kernel instead -- one workgroup per such tile, the rest of the list stays with k_tile_forces.  Same arithmetic (tile_contact), same
// outputs (the per-owner sums of the A runs in tSum, 32-byte records for the integrator's gather), but nothing is staged: a contact
// loads its two owner records from memory and converts them itself, the small tables are read where they lie, and EVERY contact
// writes a B-side record (the builder numbered them: rankC holds a contact's own record number here), also those whose B owner
// belongs to the tile.  Slower per contact -- it is the exception path -- and exact about the order of the sums like the tile pass.
template <int MODEL, bool MESH, bool REC = false>
__global__ __launch_bounds__(DEME_TILE_T) void k_tile_forces_big(const DevParams p, const TileArgs a) {
    __shared__ float4 recA4[DEME_TILE_RSLOTS];
    __shared__ float2 recA2[DEME_TILE_RSLOTS];
    const uint32_t t = a.bigList[blockIdx.x];
    if (a.tileMode && !(a.tileMode[t] & (1u << a.pass)))
        return;
    const uint32_t tid = threadIdx.x;
    const uint32_t o0 = t * DEME_TILE_NB;
    const uint32_t nLoc = min((uint32_t)DEME_TILE_NB, a.nOwners - o0);
    const uint32_t c0 = a.aStart[o0], c1 = a.aStart[o0 + nLoc];
    const int64_t u0x = a.org[3 * (size_t)t], u0y = a.org[3 * (size_t)t + 1], u0z = a.org[3 * (size_t)t + 2];
    TileTables T;
    T.comp = p.comp, T.mat = p.matPair, T.anal = p.anal, T.fam = p.familyExtra, T.mass = nullptr;
    if (tid == DEME_TILE_T - 1u)
        recA4[DEME_TILE_T] = make_float4(0, 0, 0, 0), recA2[DEME_TILE_T] = make_float2(0, 0);
    const bool sideA = tid < nLoc;
    uint32_t plo = 0, phi = 0;
    if (sideA)
        plo = a.aStart[o0 + tid] - c0, phi = a.aStart[o0 + tid + 1] - c0;
    v2f s01 = {0.f, 0.f}, s23 = {0.f, 0.f}, s45 = {0.f, 0.f};
    float4* const wc4 = reinterpret_cast<float4*>(a.wc);
    __syncthreads();
    const uint32_t nCt = c1 - c0;
    for (uint32_t rlo = 0; rlo < nCt; rlo += DEME_TILE_T) {
        const uint32_t c = c0 + rlo + tid;
        if (c < c1) {
            const uint4 gi = a.info[c];
            const uint32_t oa = gi.x & 0x3FFFFFFFu, cls = gi.x >> 30, ob = gi.y;
            const uint32_t matA = gi.z >> 16, compA = gi.z & 0xFFFFu;
            const uint32_t matB = (cls == DEME_KEY_CLASS_SS) ? (gi.w >> 16) : 0u, wB = (cls == DEME_KEY_CLASS_SS) ? (gi.w & 0xFFFFu) : gi.w;
            const uint2 ci = make_uint2(tile_info_x(0u, 0u, cls, 1u, matA, matB), compA | (wB << 16));
            float4 h = make_float4(0, 0, 0, 0);
            if (MODEL == 0)
                h = stream_load(wc4 + c);
            f3 force, tA, tB;
            if (MESH && cls == DEME_KEY_CLASS_SM) {
                const float4 a4 = a.conA4[c], b4 = a.conB4[c];
                const float2 a2 = a.conA2[c], b2 = a.conB2[c];
                force = mk3(a4.x, a4.y, a4.z), tA = mk3(a4.w, a2.x, a2.y), tB = mk3(b4.w, b2.x, b2.y);
            } else {
                __attribute__((aligned(16))) uint4 sa[DEME_TILE_REC], sb[DEME_TILE_REC];
                const OwnerRec ra = load_owner(a.owners, oa), rb = load_owner(a.owners, ob);
                tile_stage<MODEL>(p, p.massProps[ra.inertiaOff].x, ra, u0x, u0y, u0z, sa);
                tile_stage<MODEL>(p, p.massProps[rb.inertiaOff].x, rb, u0x, u0y, u0z, sb);
                const TileOwner A = tile_read<MODEL>(sa, 0u, 0u), B = tile_read<MODEL>(sb, 0u, 0u);
                if (MODEL == 2) {
                    constexpr int NWU = DEME_JIT_NW;
                    float uw[NWU];
#pragma unroll
                    for (int k = 0; k < NWU; k++)
                        uw[k] = DEME_JIT_HAS_WC ? a.wc[(size_t)c * NWU + k] : 0.f;
                    TileUser U;
                    U.ox = (double)u0x * p.l + (double)p.LBFX, U.oy = (double)u0y * p.l + (double)p.LBFY, U.oz = (double)u0z * p.l + (double)p.LBFZ;
                    U.ownerA = oa, U.ownerB = ob;
                    U.c = c, U.keys = a.keys, U.ownerWc = a.ownerWc, U.geoWcSph = a.geoWcSph, U.geoWcAnal = a.geoWcAnal, U.time = a.timeElapsed;
                    tile_contact<MODEL>(p, T, ci, A, B, h, force, tA, tB, uw, &U);
                    if (DEME_JIT_HAS_WC) {
#pragma unroll
                        for (int k = 0; k < NWU; k++)
                            a.wc[(size_t)c * NWU + k] = uw[k];
                    }
                } else if (REC) {
                    TileRecOut ro;
                    tile_contact<MODEL>(p, T, ci, A, B, h, force, tA, tB, nullptr, nullptr, &ro);
                    float* r = a.rec[0] + 3ull * c;
                    r[0] = force.x, r[1] = force.y, r[2] = force.z;
                    r = a.rec[1] + 3ull * c;
                    r[0] = ro.torqueOnly.x, r[1] = ro.torqueOnly.y, r[2] = ro.torqueOnly.z;
                    r = a.rec[2] + 3ull * c;
                    r[0] = ro.locA.x, r[1] = ro.locA.y, r[2] = ro.locA.z;
                    r = a.rec[3] + 3ull * c;
                    r[0] = ro.locB.x, r[1] = ro.locB.y, r[2] = ro.locB.z;
                } else {
                    tile_contact<MODEL>(p, T, ci, A, B, h, force, tA, tB);
                }
                if (MODEL == 0)
                    stream_store(wc4 + c, h);
            }
            recA4[tid] = make_float4(force.x, force.y, force.z, tA.x);
            recA2[tid] = make_float2(tA.y, tA.z);
            const uint32_t k = a.rankC[c];  // (this contact's own record)
#if DEME_REC24
            float2* const r24 = reinterpret_cast<float2*>(a.rec32) + 3 * (size_t)k;
            stream_store(r24, make_float2(-force.x, -force.y));
            stream_store(r24 + 1, make_float2(-force.z, tB.x));
            stream_store(r24 + 2, make_float2(tB.y, tB.z));
#else
            stream_store(a.rec32 + 2 * (size_t)k, make_float4(-force.x, -force.y, -force.z, tB.x));
            stream_store(a.rec32 + 2 * (size_t)k + 1, make_float4(tB.y, tB.z, 0.f, 0.f));
#endif
        }
        __syncthreads();
        if (sideA) {
            const uint32_t e = min(phi, rlo + DEME_TILE_T);
            while (plo < e) {
                float4 v4[4];
                float2 v2[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const uint32_t i = (plo + k < e) ? plo + k - rlo : (uint32_t)DEME_TILE_T;
                    v4[k] = recA4[i], v2[k] = recA2[i];
                }
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    s01 += v2f{v4[k].x, v4[k].y};
                    s23 += v2f{v4[k].z, v4[k].w};
                    s45 += v2f{v2[k].x, v2[k].y};
                }
                plo = min(plo + 4u, e);
            }
        }
        __syncthreads();
    }
    if (sideA) {
#if DEME_REC24
        float2* const t24 = reinterpret_cast<float2*>(a.tSum) + 3 * (size_t)(o0 + tid);
        t24[0] = make_float2(s01.x, s01.y), t24[1] = make_float2(s23.x, s23.y), t24[2] = make_float2(s45.x, s45.y);
#else
        a.tSum[2 * (size_t)(o0 + tid)] = make_float4(s01.x, s01.y, s23.x, 0.f);
        a.tSum[2 * (size_t)(o0 + tid) + 1] = make_float4(s23.y, s45.x, s45.y, 0.f);
#endif
    }
}

#if !defined(DEME_JIT) && !defined(DEME_TILE_FORCE_ONLY)  // (the run-time compiled copy of this header and deme_tile_p.hip hold the force pass only)
// ---- per-detection builders ----------------------------------------------------------------------------------------------------
// (1) k_contact_owners counts, per tile, the contacts whose B owner lives in another tile (tileRem; deme_kernels.h); an exclusive
//     scan gives every tile the number of its first record (tileBase).
// (2) k_tile_build, one workgroup per tile: the sorted list of the foreign owners the tile's contacts touch, the 8-byte gather
//     records, the tile's origin, the pass of the halo overlap it belongs to; the per-owner lists of contacts that hold an owner as
//     B from inside the tile (LDS counting sort, each list put in ascending order afterwards: the summation order is fixed); for
//     every contact the number of crossing contacts before it (rankC) and for the crossing ones the pair (B owner, record number).
// (3) a radix sort of those pairs by owner -- 41 % of the contacts of a packed bed; the round-2 pipeline sorted all of them -- gives
//     the integrator its per-owner lists of records (rIdx, rStart).
__global__ __launch_bounds__(256) void k_tile_build(const DevParams p, uint32_t nOwners, const uint4* __restrict__ info,
                                                    const uint32_t* __restrict__ ownerBList,
                                                    const uint32_t* __restrict__ aStart, const OwnerRec* __restrict__ owners,
                                                    const uint32_t* __restrict__ tileBase, uint2* __restrict__ tInfo,
                                                    uint32_t* __restrict__ hList, uint32_t* __restrict__ hCount,
                                                    uint32_t* __restrict__ tileMode, uint16_t* __restrict__ lOff,
                                                    uint16_t* __restrict__ lPos, uint32_t* __restrict__ lCount,
                                                    uint32_t* __restrict__ rankC, uint32_t* __restrict__ remKey,
                                                    uint32_t* __restrict__ remVal, int64_t* __restrict__ org, RangeCounters* rc,
                                                    uint32_t nTiles, uint32_t* __restrict__ tileBig, uint32_t* __restrict__ bigList,
                                                    uint32_t* __restrict__ recContact) {
    __shared__ uint32_t table[DEME_TILE_HASH];
    __shared__ uint16_t slotTab[DEME_TILE_HASH];
    __shared__ uint32_t list[DEME_TILE_HP2];
    __shared__ uint32_t cnt[DEME_TILE_NB], off[DEME_TILE_NB + 1], wsum[4];
    __shared__ uint16_t lp[DEME_TILE_LMAX];
    __shared__ uint32_t nU, nL, anyGhost;
    const uint32_t t = blockIdx.x, tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t o0 = t * DEME_TILE_NB, o1 = min(o0 + (uint32_t)DEME_TILE_NB, nOwners);
    const uint32_t c0 = aStart[o0], c1 = aStart[o1];
    for (uint32_t i = tid; i < DEME_TILE_HASH; i += 256)
        table[i] = 0xFFFFFFFFu;
    for (uint32_t i = tid; i < DEME_TILE_HP2; i += 256)
        list[i] = 0xFFFFFFFFu;
    if (tid < DEME_TILE_NB)
        cnt[tid] = 0;
    if (tid == 0)
        nU = 0, nL = 0, anyGhost = 0;
    __syncthreads();
    for (uint32_t c = c0 + tid; c < c1; c += 256) {
        const uint32_t ob = ownerBList[c];  // (the B owners alone: 4 bytes per contact; the 16-byte gather records are read once, below)
        if (ob >= o0 && ob < o1) {
            atomicAdd(&cnt[ob - o0], 1u);
            continue;
        }
        uint32_t h = (ob * 2654435761u) >> 22;  // 10 bits
        while (*(volatile uint32_t*)&nU <= DEME_TILE_HMAX) {
            const uint32_t old = atomicCAS(&table[h], 0xFFFFFFFFu, ob);
            if (old == 0xFFFFFFFFu) {  // a new foreign owner: it takes the next staging slot.  (Which one depends on who comes first: the
                                       // slot numbers are addresses of LDS records, not an order anything is summed in)
                const uint32_t slot = atomicAdd(&nU, 1u);
                slotTab[h] = (uint16_t)slot;
                if (slot < DEME_TILE_HP2)
                    list[slot] = ob;
                break;
            }
            if (old == ob)
                break;
            h = (h + 1u) & (DEME_TILE_HASH - 1u);
        }
    }
    __syncthreads();
    // exclusive scan of the owners' local-B counts (one value per thread, wavefront scans + the wavefronts' totals)
    {
        static_assert(DEME_TILE_NB <= 256, "the builder has one thread per owner of the tile");
        uint32_t v = (tid < DEME_TILE_NB) ? cnt[tid] : 0u, inc = v;
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t u = (uint32_t)__shfl_up((int)inc, d);
            if ((int)lane >= d)
                inc += u;
        }
        if (lane == 63u)
            wsum[wave] = inc;
        __syncthreads();
        if (tid < DEME_TILE_NB) {
            uint32_t base = 0;
            for (uint32_t w = 0; w < wave; w++)
                base += wsum[w];
            off[tid] = base + inc - v;
            if (tid == DEME_TILE_NB - 1u)
                off[DEME_TILE_NB] = base + inc;
        }
        __syncthreads();
    }
    const uint32_t n = nU, nLoc = off[DEME_TILE_NB];
    if (n > DEME_TILE_HMAX || c1 - c0 > DEME_TILE_CMAX || nLoc > DEME_TILE_LMAX) {
        // The tile does not fit the LDS area of k_tile_forces: k_tile_forces_big evaluates it (nothing staged, EVERY contact writes a
        // record).  Its crossing contacts take the record numbers the scan reserved for them; the contacts that hold an owner of the
        // tile as B get numbers behind all of those -- one reservation per such tile -- and join the pairs the integrator's per-owner
        // record lists are sorted from.
        __shared__ uint32_t extraBase, bigGhost;
        if (tid == 0) {
            tileBig[t] = 1u;
            bigList[atomicAdd(&rc->nBig, 1u)] = t;
            extraBase = tileBase[nTiles] + atomicAdd(&rc->nExtra, nLoc);
            bigGhost = 0;
            hCount[t] = 0;
            lCount[t] = 0;
            int64_t ux, uy, uz;
            pos_units(load_owner(owners, o0), p, ux, uy, uz);
            org[3 * (size_t)t] = ux, org[3 * (size_t)t + 1] = uy, org[3 * (size_t)t + 2] = uz;
        }
        for (uint32_t i = tid; i <= DEME_TILE_NB; i += 256)
            lOff[(size_t)t * (DEME_TILE_NB + 1) + i] = 0;
        __syncthreads();
        uint32_t runR = tileBase[t], runL = extraBase;
        bool g = false;
        for (uint32_t cb = c0; cb < c1; cb += 256) {
            const uint32_t c = cb + tid;
            bool remote = false, local = false;
            uint32_t ob = 0;
            if (c < c1) {
                ob = ownerBList[c];
                local = ob >= o0 && ob < o1;
                remote = !local;
                if (tileMode)
                    g = g || ghost_of(owners[ob].family) || ghost_of(owners[info[c].x & 0x3FFFFFFFu].family);
            }
            const unsigned long long mr = __ballot(remote), ml = __ballot(local);
            if (lane == 0)
                wsum[wave] = (uint32_t)__popcll(mr) | ((uint32_t)__popcll(ml) << 16);
            __syncthreads();
            uint32_t beforeR = 0, beforeL = 0, totR = 0, totL = 0;
            for (uint32_t w = 0; w < 4; w++) {
                const uint32_t v = wsum[w];
                beforeR += (w < wave) ? (v & 0xFFFFu) : 0u, beforeL += (w < wave) ? (v >> 16) : 0u;
                totR += v & 0xFFFFu, totL += v >> 16;
            }
            if (c < c1) {
                const unsigned long long below = (1ull << lane) - 1ull;
                const uint32_t r = remote ? runR + beforeR + (uint32_t)__popcll(mr & below) : runL + beforeL + (uint32_t)__popcll(ml & below);
                rankC[c] = r;  // (the contact's OWN record, whichever kind it is)
                remKey[r] = ob, remVal[r] = r, recContact[r] = c;
            }
            runR += totR, runL += totL;
            __syncthreads();
        }
        if (tileMode) {
            if (g)
                bigGhost = 1;
            __syncthreads();
            if (tid == 0)
                tileMode[t] = bigGhost ? 2u : 1u;
        }
        return;
    }
    if (tid == 0)
        tileBig[t] = 0u;
    if (tid < DEME_TILE_NB)
        cnt[tid] = 0;  // (from here on: entries already placed in an owner's list)
    __syncthreads();
    for (uint32_t i = tid; i < n; i += 256)
        hList[(size_t)t * DEME_TILE_HMAX + i] = list[i];
    if (tid == 0) {
        hCount[t] = n;
        lCount[t] = nLoc;
        int64_t ux, uy, uz;
        pos_units(load_owner(owners, o0), p, ux, uy, uz);
        org[3 * (size_t)t] = ux, org[3 * (size_t)t + 1] = uy, org[3 * (size_t)t + 2] = uz;
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
    // the contacts in list order, 256 at a time: gather records, local-B lists, record numbers of the crossing ones
    uint32_t run = tileBase[t];
    for (uint32_t cb = c0; cb < c1; cb += 256) {
        const uint32_t c = cb + tid;
        bool remote = false;
        uint32_t ob = 0;
        if (c < c1) {
            const uint4 ci = info[c];
            const uint32_t oa = ci.x & 0x3FFFFFFFu, cls = ci.x >> 30;
            ob = ci.y;
            uint32_t slotB;
            if (ob >= o0 && ob < o1) {
                slotB = ob - o0;
                lp[off[slotB] + atomicAdd(&cnt[slotB], 1u)] = (uint16_t)(c - c0);
            } else {
                uint32_t h = (ob * 2654435761u) >> 22;
                while (table[h] != ob)  // (inserted above: found after one or two probes)
                    h = (h + 1u) & (DEME_TILE_HASH - 1u);
                slotB = DEME_TILE_NB + slotTab[h];
                remote = true;
            }
            const uint32_t matA = ci.z >> 16, compA = ci.z & 0xFFFFu;
            const uint32_t matB = (cls == DEME_KEY_CLASS_SS) ? (ci.w >> 16) : 0u;  // (an analytical object's material is in its record)
            const uint32_t wB = (cls == DEME_KEY_CLASS_SS) ? (ci.w & 0xFFFFu) : ci.w;
            tInfo[c] = make_uint2(tile_info_x(oa - o0, slotB, cls, remote ? 1u : 0u, matA, matB), compA | (wB << 16));
        }
        const unsigned long long m = __ballot(remote);
        if (lane == 0)
            wsum[wave] = (uint32_t)__popcll(m);
        __syncthreads();
        uint32_t before = 0, total = 0;
        for (uint32_t w = 0; w < 4; w++) {
            const uint32_t v = wsum[w];
            before += (w < wave) ? v : 0u;
            total += v;
        }
        const uint32_t r = run + before + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        if (c < c1) {
            rankC[c] = r;
            if (remote)
                remKey[r] = ob, remVal[r] = r, recContact[r] = c;  // (recContact: which contact a record belongs to -- deme_tile_step.h)
        }
        run += total;
        __syncthreads();
    }
    // every owner's list in ascending order (a few entries each)
    if (tid < DEME_TILE_NB) {
        const uint32_t b = off[tid], e = off[tid + 1];
        for (uint32_t i = b + 1; i < e; i++) {
            const uint16_t v = lp[i];
            uint32_t j = i;
            while (j > b && lp[j - 1] > v) {
                lp[j] = lp[j - 1];
                j--;
            }
            lp[j] = v;
        }
    }
    __syncthreads();
    for (uint32_t i = tid; i < nLoc; i += 256)
        lPos[c0 + i] = lp[i];
    for (uint32_t i = tid; i <= DEME_TILE_NB; i += 256)
        lOff[(size_t)t * (DEME_TILE_NB + 1) + i] = (uint16_t)off[i];
}

// the largest tile's foreign-owner count and local-B list length (they size the LDS of k_tile_forces): a reduction over the
// per-tile values with one pair of atomics per workgroup -- 8 000 atomicMax on one address from k_tile_build cost 190 us
// (same-address atomics serialise at ~12 ns)
__global__ __launch_bounds__(256) void k_tile_stats(uint32_t nTiles, const uint32_t* __restrict__ hCount,
                                                    const uint32_t* __restrict__ lCount, RangeCounters* rc) {
    __shared__ uint32_t mh[4], ml[4], sh[4];
    uint32_t a = 0, b = 0, sum = 0;
    for (uint32_t k = 0; k < 4; k++) {
        const uint32_t t = (blockIdx.x * 4u + k) * 256u + threadIdx.x;
        if (t < nTiles)
            a = max(a, hCount[t]), b = max(b, lCount[t]), sum += hCount[t];
    }
    for (int o = 32; o > 0; o >>= 1) {
        a = max(a, (uint32_t)__shfl_xor((int)a, o));
        b = max(b, (uint32_t)__shfl_xor((int)b, o));
        sum += (uint32_t)__shfl_xor((int)sum, o);
    }
    if ((threadIdx.x & 63u) == 0)
        mh[threadIdx.x >> 6] = a, ml[threadIdx.x >> 6] = b, sh[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(&rc->tileHaloSum, sh[0] + sh[1] + sh[2] + sh[3]);
        atomicMax(&rc->tileMaxHalo, max(max(mh[0], mh[1]), max(mh[2], mh[3])));
        atomicMax(&rc->tileMaxList, max(max(ml[0], ml[1]), max(ml[2], ml[3])));
    }
}

// heavy / fixed flags of the owners when the list has tile structures: an owner's B-side entries are its tile-local list plus its
// records (k_owner_ranges, deme_kernels.h, reads the B-sorted list of the round-2 pipeline instead)
__global__ __launch_bounds__(256) void k_owner_ranges_tile(const DevParams p, const OwnerRec* __restrict__ owners,
                                                           const uint32_t* __restrict__ aStart, const uint16_t* __restrict__ lOff,
                                                           const uint32_t* __restrict__ rStart, uint8_t* __restrict__ heavy,
                                                           uint8_t* __restrict__ fixedFlag, uint32_t* __restrict__ heavyList,
                                                           uint32_t heavyCap, RangeCounters* rc) {
    const uint32_t o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= p.nOwners)
        return;
    const uint32_t t = o / DEME_TILE_NB, k = o - t * DEME_TILE_NB;
    const uint16_t* lo = lOff + (size_t)t * (DEME_TILE_NB + 1);
    const uint32_t nA = aStart[o + 1] - aStart[o], nB = (uint32_t)(lo[k + 1] - lo[k]) + (rStart[o + 1] - rStart[o]);
    const uint32_t fw = owners[o].family;
    const bool isFixed = (p.familyFlags[fam_of(fw)] & 1u) != 0 || ghost_of(fw);
    fixedFlag[o] = isFixed ? 1 : 0;
    const bool hv = nA + nB > DEME_HEAVY_THRESHOLD || shared_of(fw);
    heavy[o] = hv ? 1 : 0;
    if (hv) {
        const unsigned int slot = atomicAdd(&rc->nHeavy, 1u);
        if (slot < heavyCap)
            heavyList[slot] = o;
        if (!isFixed)
            atomicAdd(&rc->nHeavyFree, 1u);
    }
}

#endif  // DEME_JIT

}  // namespace deme_dev

#pragma clang fp contract(off)
