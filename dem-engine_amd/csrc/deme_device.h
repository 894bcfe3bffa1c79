// deme_device.h -- device-side data layout and arithmetic helpers (gfx950 / CDNA4).
//
// Self-contained on purpose (only the HIP runtime header): the same text is fed to
// hipRTC when a user force-model fragment is compiled at run time (deme_jit.h).
//
// Arithmetic follows the reference function named at each helper (paths relative
// to the reference's src/); the operand types and narrowing points are the
// contract that makes contact / bin decisions bit-identical to the CPU oracle.
// This translation unit is compiled with -ffp-contract=off.
#pragma once
#ifdef __HIPCC_RTC__
// hipRTC: the HIP runtime declarations are built in; no system headers are available
typedef unsigned char uint8_t;
typedef unsigned short uint16_t;
typedef unsigned int uint32_t;
typedef unsigned long long uint64_t;
typedef long long int64_t;
#else
#include <hip/hip_runtime.h>
#include <stdint.h>
#endif

namespace deme_dev {

// ---------------------------------------------------------------------------
// HBM layout.  The reference keeps ~20 separate per-owner arrays (DEM/Defines.h:269-373);
// the force kernel gathers two owners per contact, so here an owner is ONE 64-byte,
// 64-byte-aligned record: a gather touches one cache-line sector instead of 14.
// ---------------------------------------------------------------------------
struct __attribute__((aligned(16))) OwnerRec {
    uint64_t voxelID;            // DEM/Defines.h:272 voxelID
    uint16_t locX, locY, locZ;   // sub-voxel offsets
    uint16_t inertiaOff;         // index into the mass-property table
    float qw, qx, qy, qz;        // orientation
    float vx, vy, vz;            // linear velocity
    uint32_t family;             // bits 0-7 family id; bit 8: ghost copy of a clump another rank owns; bit 9: replicated owner (OWNER_*_BIT)
    float wx, wy, wz;            // body-frame angular velocity (omgBar)
    float margin;                // contact-detection margin (kT's marginSize)
};
static_assert(sizeof(OwnerRec) == 64, "OwnerRec must be 64 bytes");

#define OWNER_GHOST_BIT 0x100u
// a free body every slab keeps a replica of (a mesh or an analytical object that moves under contact forces): each slab sums the
// contributions of the spheres it owns, the sums are added up across slabs every step, and every replica is integrated alike
#define OWNER_SHARED_BIT 0x200u
// a ghost whose contacts with this slab's own clumps are evaluated by the rank that owns it, which sends the reaction back
// (one evaluation and one history per cross-cut contact: deme_halo_group_set_cross_contacts)
#define OWNER_PASSIVE_BIT 0x400u
#define OWNER_FLAG_BITS (OWNER_GHOST_BIT | OWNER_SHARED_BIT | OWNER_PASSIVE_BIT)
__host__ __device__ inline uint32_t fam_of(uint32_t familyWord) { return familyWord & 0xFFu; }
__host__ __device__ inline bool ghost_of(uint32_t familyWord) { return (familyWord & OWNER_GHOST_BIT) != 0; }
__host__ __device__ inline bool shared_of(uint32_t familyWord) { return (familyWord & OWNER_SHARED_BIT) != 0; }
__host__ __device__ inline bool passive_of(uint32_t familyWord) { return (familyWord & OWNER_PASSIVE_BIT) != 0; }

struct SphereRec {  // 8 bytes: ownerClumpBody + clumpComponentOffset + sphereMaterialOffset
    uint32_t owner;
    uint16_t comp;
    uint16_t mat;
};

struct __attribute__((aligned(16))) AccRec {  // a and alpha of one owner, one 32-byte atomic target
    float ax, ay, az, pad0;
    float lx, ly, lz, pad1;
};

struct __attribute__((aligned(16))) GeoRec {  // per-sphere world geometry staged for the bin sweep
    double x, y, z;  // LBF-shifted frame (the kinematic side works without LBF, DEMBinSphereKernels.cu:39-48)
    float r;         // radius + margin, fp32 sum (DEMContactKernels_SphereSphere.cu:31-41)
    uint32_t owner;
};
static_assert(sizeof(GeoRec) == 32, "GeoRec must be 32 bytes");

struct __attribute__((aligned(16))) AnalObj {  // AnalyticalCompDefJitify.cu:2-15, one record per component
    float relx, rely, relz, size1;
    float rotx, roty, rotz, size2;
    float normal, mass, size3;
    uint32_t owner;
    uint32_t type;
    uint32_t mat;
    uint32_t pad0, pad1;
};

// One record per owner whose family carries a motion prescription (DEME_FAMILY_PRESCRIBED): what the user's
// prescription code (compiled at run time, deme_jit.h) decided for this step.  Consumed by k_integrate with the
// per-component semantics of integrateVelPos (kernel/DEMIntegrationKernels.cu:100-236).
struct __attribute__((aligned(8))) PrescRec {
    double X, Y, Z;           // position after applyPrescribedPos (world frame, LBF included)
    float vx, vy, vz;         // velocities after applyPrescribedVel
    float wx, wy, wz;
    float qw, qx, qy, qz;     // orientation after applyPrescribedPos
    float ax, ay, az;         // applyAddedAcceleration
    float lx, ly, lz;
    uint32_t flags;           // bit 0-2 LinVelX/Y/Z, 3-5 RotVelX/Y/Z, 6-8 LinX/Y/Z, 9 Rot prescribed
    uint32_t pad;
};

struct MatPair {  // per (matA, matB): everything the Hertzian models derive from materials alone
    float E_cnt, G_cnt, CoR, mu, Crr, beta, E_A_unused, pad;
};

struct DevParams {
    uint32_t nvXp2, nvYp2;
    uint32_t nbX, nbY, nbZ;
    uint32_t mNbX, mNbXY;  // floor(2^32 / nbX), floor(2^32 / (nbX * nbY)) for fast_div (refresh_dev_params)
    double l, voxelSize, binSize;
    float LBFX, LBFY, LBFZ;
    float Gx, Gy, Gz;
    float h;
    float approxMaxVel, expSafetyMulti, expSafetyAdder, errOutVel;
    uint32_t integrator, forceModel, nW;
    uint32_t nOwners, nSpheres, nAnal, nMat;
    uint32_t errOutBinSphNum;
    uint32_t familyTrivial;  // 1: all masks allow and all extra margins are 0 -> skip family logic
    uint32_t hasGhosts;      // bit 0: some owners are ghost copies (the sweep reads the family words to leave ghost-ghost pairs out);
                             // bit 1: every cross-cut contact is evaluated by ONE rank (pairs of an own clump with a passive ghost,
                             // and a ghost sphere's contacts with walls and meshes, are the owning rank's)
    // tables
    const float4* comp;       // per component: relx, rely, relz, radius
    const float4* massProps;  // per mass property: mass, moiX, moiY, moiZ
    const AnalObj* anal;
    const MatPair* matPair;   // nMat*nMat
    const float* E;           // raw per-material arrays for custom models
    const float* nu;
    const float* CoR;
    const float* mu;
    const float* Crr;
    const uint8_t* familyMasks;
    const float* familyExtra;
    const uint8_t* familyFlags;
    const void* tris;  // TriRec[nTri] (deme_mesh.h)
    uint32_t nTri;
    // engine-side spatial order (deme_order.inc): the caller's id of every sphere / owner when the engine numbers them along a
    // Z-order curve of its own; null = the caller's order is the engine's
    const uint32_t* s2e;
    const uint32_t* o2e;
};

// n / d for a run-time divisor without the ~35-instruction integer division: m = floor(2^32 / d) (0xFFFFFFFF for d = 1) gives
// umulhi(n, m) in {q - 1, q}; one correction step makes it exact for every 32-bit n
__host__ __device__ inline uint32_t fast_div_magic(uint32_t d) { return d <= 1u ? 0xFFFFFFFFu : (uint32_t)(0x100000000ull / d); }
__device__ inline uint32_t fast_div(uint32_t n, uint32_t d, uint32_t m) {
    const uint32_t q = __umulhi(n, m);
    return (n - q * d >= d) ? q + 1u : q;
}

// contact key: sphere A (31 bits) | type class (2 bits) | B (31 bits).  Sorting keys ascending yields the
// canonical list order: by A, then sphere-sphere / sphere-mesh / sphere-analytical, then B.  Spheres are
// clump-major, so the list is also sorted by A's owner: every owner's A-side contacts are one contiguous run.
#define DEME_KEY_CLASS_SS 0ull
#define DEME_KEY_CLASS_SM 1ull
#define DEME_KEY_CLASS_SA 2ull
__host__ __device__ inline uint64_t make_key(uint64_t cls, uint32_t a, uint32_t b) {
    return ((uint64_t)a << 33) | (cls << 31) | (uint64_t)b;
}
__host__ __device__ inline uint32_t key_class(uint64_t k) { return (uint32_t)((k >> 31) & 3u); }
__host__ __device__ inline uint32_t key_a(uint64_t k) { return (uint32_t)(k >> 33); }
__host__ __device__ inline uint32_t key_b(uint64_t k) { return (uint32_t)(k & 0x7FFFFFFFu); }

// ---------------------------------------------------------------------------
// small vector math (own header: the reference's CUDAMathHelpers.cuh clashes with HIP's
// vector operators)
// ---------------------------------------------------------------------------
struct f3 {
    float x, y, z;
};
struct d3 {
    double x, y, z;
};
__device__ inline f3 mk3(float x, float y, float z) { return {x, y, z}; }
__device__ inline f3 operator+(f3 a, f3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ inline f3 operator-(f3 a, f3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ inline f3 operator*(float s, f3 a) { return {s * a.x, s * a.y, s * a.z}; }
__device__ inline f3 operator*(f3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ inline f3 operator/(f3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
__device__ inline float dot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ inline f3 cross3(f3 a, f3 b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__device__ inline float len3(f3 a) { return sqrtf(dot3(a, a)); }

// kernel/DEMHelperKernels.cuh:92-134 (IDChopper, voxelIDToPosition)
__device__ inline d3 decode_pos(uint64_t id, uint32_t sx, uint32_t sy, uint32_t sz, const DevParams& p) {
    const uint64_t vx = id & (((uint64_t)1 << p.nvXp2) - 1);
    const uint64_t vy = (id >> p.nvXp2) & (((uint64_t)1 << p.nvYp2) - 1);
    const uint64_t vz = id >> (p.nvXp2 + p.nvYp2);
    d3 r;
    r.x = (double)vx * p.voxelSize + (double)sx * p.l;
    r.y = (double)vy * p.voxelSize + (double)sy * p.l;
    r.z = (double)vz * p.voxelSize + (double)sz * p.l;
    return r;
}
// kernel/DEMHelperKernels.cuh:136-159 (positionToVoxelID), truncating division
__device__ inline void encode_pos(d3 X, const DevParams& p, uint64_t& id, uint16_t& sx, uint16_t& sy, uint16_t& sz) {
    const uint64_t nx = (uint64_t)(X.x / p.voxelSize);
    const uint64_t ny = (uint64_t)(X.y / p.voxelSize);
    const uint64_t nz = (uint64_t)(X.z / p.voxelSize);
    sx = (uint16_t)((X.x - (double)nx * p.voxelSize) / p.l);
    sy = (uint16_t)((X.y - (double)ny * p.voxelSize) / p.l);
    sz = (uint16_t)((X.z - (double)nz * p.voxelSize) / p.l);
    id = nx + (ny << p.nvXp2) + (nz << (p.nvXp2 + p.nvYp2));
}

// kernel/DEMHelperKernels.cuh:161-173 (applyOriQToVector3): nine fp32 coefficients
struct RotM {
    float xx, xy, xz, yx, yy, yz, zx, zy, zz;
};
__device__ inline RotM rot_coeffs(float w, float x, float y, float z) {
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
__device__ inline RotM rot_transpose(const RotM& m) {
    RotM t;
    t.xx = m.xx, t.xy = m.yx, t.xz = m.zx;
    t.yx = m.xy, t.yy = m.yy, t.yz = m.zy;
    t.zx = m.xz, t.zy = m.yz, t.zz = m.zz;
    return t;
}
__device__ inline f3 rot_apply(const RotM& m, f3 v) {
    f3 r;
    r.x = m.xx * v.x + m.xy * v.y + m.xz * v.z;
    r.y = m.yx * v.x + m.yy * v.y + m.yz * v.z;
    r.z = m.zx * v.x + m.zy * v.y + m.zz * v.z;
    return r;
}
__device__ inline d3 rot_apply_d(const RotM& m, d3 v) {
    d3 r;
    r.x = m.xx * v.x + m.xy * v.y + m.xz * v.z;
    r.y = m.yx * v.x + m.yy * v.y + m.yz * v.z;
    r.z = m.zx * v.x + m.zy * v.y + m.zz * v.z;
    return r;
}

// kernel/DEMHelperKernels.cuh:57-62 (locateMaskPair)
__device__ inline uint32_t mask_pair(uint32_t i, uint32_t j) {
    if (i > j) {
        const uint32_t t = i;
        i = j;
        j = t;
    }
    return (1 + j) * j / 2 + i;
}

// kernel/DEMHelperKernels.cuh:329-336 (getPointBinID)
__device__ inline uint32_t point_bin(double X, double Y, double Z, const DevParams& p) {
    const uint32_t bx = (uint32_t)(X / p.binSize);
    const uint32_t by = (uint32_t)(Y / p.binSize);
    const uint32_t bz = (uint32_t)(Z / p.binSize);
    return bx + by * p.nbX + bz * p.nbX * p.nbY;
}

// kernel/DEMHelperKernels.cuh:292-326 (checkSpheresOverlap<double,float>)
__device__ inline bool spheres_overlap(double XA, double YA, double ZA, double rA, double XB, double YB, double ZB,
                                       double rB, d3& CP, f3& n, double& depth) {
    const double d2 = (XA - XB) * (XA - XB) + (YA - YB) * (YA - YB) + (ZA - ZB) * (ZA - ZB);
    const bool hit = !(d2 > (rA + rB) * (rA + rB));
    n.x = (float)(XA - XB);
    n.y = (float)(YA - YB);
    n.z = (float)(ZA - ZB);
    const float mag = sqrtf(n.x * n.x + n.y * n.y + n.z * n.z);
    n.x /= mag;
    n.y /= mag;
    n.z /= mag;
    depth = rA + rB - sqrt(d2);
    CP.x = XB + (rB - depth / 2.0) * n.x;
    CP.y = YB + (rB - depth / 2.0) * n.y;
    CP.z = ZB + (rB - depth / 2.0) * n.z;
    return hit;
}

// kernel/DEMHelperKernels.cuh:459-521 (checkSphereEntityOverlap<double3,float,double>)
__device__ inline uint32_t sphere_entity(d3 A, float radA, uint32_t typeB, d3 B, f3 dirB, float size1,
                                         float normal_sign, float beta4Entity, d3& CP, f3& nrm, double& depth) {
    if (typeB == 0) {  // plane
        const d3 p2s{A.x - B.x, A.y - B.y, A.z - B.z};
        const double dist = (float)(p2s.x * dirB.x + p2s.y * dirB.y + p2s.z * dirB.z);
        depth = (radA + beta4Entity - dist);
        const uint32_t t = (depth < 0.0) ? 0u : 11u;
        const float s = (float)(dist + depth / 2.0);
        const f3 off = dirB * s;
        CP = {A.x - (double)off.x, A.y - (double)off.y, A.z - (double)off.z};
        nrm = dirB;
        return t;
    } else if (typeB == 2) {  // infinite cylinder
        d3 s2c{B.x - A.x, B.y - A.y, B.z - A.z};
        const double proj = (float)(s2c.x * dirB.x + s2c.y * dirB.y + s2c.z * dirB.z);
        const f3 pd = (float)proj * dirB;
        s2c.x -= pd.x;
        s2c.y -= pd.y;
        s2c.z -= pd.z;
        const double dr = sqrt(s2c.x * s2c.x + s2c.y * s2c.y + s2c.z * s2c.z);
        const float cyl_rad = size1 - normal_sign * beta4Entity;
        depth = radA - normal_sign * (cyl_rad - dr);
        const uint32_t t = (depth < 0.0) ? 0u : 13u;
        if (dr >= 1e-12) {
            const double f = normal_sign / dr;
            nrm = {(float)(f * s2c.x), (float)(f * s2c.y), (float)(f * s2c.z)};
            const float s = (float)(radA - depth / 2.0);
            const f3 off = nrm * s;
            CP = {A.x - (double)off.x, A.y - (double)off.y, A.z - (double)off.z};
        } else {
            nrm = dirB;
            CP = A;
        }
        return t;
    }
    depth = 0;
    CP = A;
    nrm = dirB;
    return 0u;
}

// kernel/DEMBinSphereKernels.cu:51-69 (bin span of an inflated sphere along one axis)
__device__ inline void bin_range(double pos, double radius, double binSize, uint32_t nb, uint32_t& lo, uint32_t& hi) {
    const double b = pos / binSize;
    const double span = radius / binSize;
    hi = (b + span < (double)nb) ? (uint32_t)(b + span) : nb - 1;
    lo = (uint32_t)((b - span > 0.0) ? b - span : 0.0);
}

// 64-byte owner record as four 16-byte loads (one cache-line sector)
__device__ inline OwnerRec load_owner(const OwnerRec* owners, uint32_t o) {
    OwnerRec r;
    const uint4* p = reinterpret_cast<const uint4*>(owners + o);
    uint4* q = reinterpret_cast<uint4*>(&r);
    q[0] = p[0];
    q[1] = p[1];
    q[2] = p[2];
    q[3] = p[3];
    return r;
}
__device__ inline void store_owner(OwnerRec* owners, uint32_t o, const OwnerRec& r) {
    uint4* p = reinterpret_cast<uint4*>(owners + o);
    const uint4* q = reinterpret_cast<const uint4*>(&r);
    p[0] = q[0];
    p[1] = q[1];
    p[2] = q[2];
    p[3] = q[3];
}
__device__ inline SphereRec load_sphere(const SphereRec* s, uint32_t i) {
    const uint2 v = *reinterpret_cast<const uint2*>(s + i);
    SphereRec r;
    r.owner = v.x;
    r.comp = (uint16_t)(v.y & 0xFFFFu);
    r.mat = (uint16_t)(v.y >> 16);
    return r;
}

}  // namespace deme_dev
