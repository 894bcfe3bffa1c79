/*
 * deme_oracle.cpp -- CPU restatement of the DEM-Engine hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (dem-engine_amd/, the
 * C-ABI library, bench.py's timed GPU leg) may link, import or call this file.
 * It exists so that tests/, __graft_entry__.smoke() and bench.py's
 * `cpu_baseline` leg have something to compare the HIP path against.
 *
 * Every function states the reference file:line (relative to the reference's
 * src/ directory) whose arithmetic it follows.  The arithmetic (operand types,
 * evaluation order, narrowing points) is reproduced exactly; the code
 * structure is this repo's own.  Build with -ffp-contract=off so no FMA is
 * formed: decisions (contact / bin) then agree bit-for-bit with the HIP kernels,
 * which are compiled the same way.
 *
 * Pinning: the element-wise functions below (orc_el_*) are checked against the
 * reference's own __host__ __device__ helpers and force-model fragments built
 * into oracle/_ref/libdeme_ref.so (see oracle/Makefile, oracle/ref_glue.cpp) and
 * against the golden fixtures under tests/golden/ generated from that build.
 * The reference has no test suite of its own, so those fixtures
 * are the pin.  The sphere-triangle functions are restated but NOT pinned:
 * DEMCollisionKernels.cu needs device-only round-up intrinsics and does not
 * build on the host.
 */
#include <algorithm>
#include <cfenv>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <iterator>
#include <vector>

#include "../include/deme_hip.h"

#ifdef _OPENMP
#include <omp.h>
// ORC_PERF (oracle/Makefile: libdeme_oracle_perf.so, -O3 -march=native): the build bench.py TIMES as its cpu_baseline.  Same
// algorithmic steps, with the list building, the sorts, the history map and the accumulation spread over the OpenMP team
// ("OpenMP over owners / contacts / bins").  The parity build (no ORC_PERF, -O2 -ffp-contract=off) is what the tests
// compare the HIP path with; tests/test_oracle_perf_build.py holds the two builds to each other.
#ifdef ORC_PERF
#include <parallel/algorithm>
#define ORC_SORT __gnu_parallel::sort
#define ORC_STABLE_SORT __gnu_parallel::stable_sort
#else
#define ORC_SORT std::sort
#define ORC_STABLE_SORT std::stable_sort
#endif
#endif

namespace {

constexpr double kTiny = 1e-12;                   // DEME_TINY_FLOAT, DEM/Defines.h:25
constexpr float kHugeF = 1e15;                    // DEME_HUGE_FLOAT, DEM/Defines.h:26
constexpr double kTwoThirds = 2. / 3.;            // DEM/Defines.h:37-43
constexpr double kFourThirds = 4. / 3.;
constexpr double kFiveThirds = 5. / 3.;
constexpr double kTwoSqrt56 = 1.825741858350554;
constexpr double kPi = 3.1415926535897932385;
constexpr double kPiSq = 9.869604401089358;

struct V3f {
    float x, y, z;
};
struct V3d {
    double x, y, z;
};
struct Q4 {
    float w, x, y, z;
};

inline V3f operator+(V3f a, V3f b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3f operator-(V3f a, V3f b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3f operator*(float s, V3f a) { return {s * a.x, s * a.y, s * a.z}; }
inline V3f operator*(V3f a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline V3f operator/(V3f a, float s) { return {a.x / s, a.y / s, a.z / s}; }
inline float dotf(V3f a, V3f b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3f crossf(V3f a, V3f b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline float lenf(V3f a) { return sqrtf(dotf(a, a)); }

// ---------------------------------------------------------------------------
// Position codec.  kernel/DEMHelperKernels.cuh:92-101 (IDChopper), :118-134
// (voxelIDToPosition), :138-159 (positionToVoxelID: truncating division).
// ---------------------------------------------------------------------------
inline void decode_pos(uint64_t id, uint16_t sx, uint16_t sy, uint16_t sz, unsigned nvXp2, unsigned nvYp2,
                       double voxelSize, double l, double& X, double& Y, double& Z) {
    const uint64_t vx = id & (((uint64_t)1 << nvXp2) - 1);
    const uint64_t vy = (id >> nvXp2) & (((uint64_t)1 << nvYp2) - 1);
    const uint64_t vz = id >> (nvXp2 + nvYp2);
    X = (double)vx * voxelSize + (double)sx * l;
    Y = (double)vy * voxelSize + (double)sy * l;
    Z = (double)vz * voxelSize + (double)sz * l;
}

inline void encode_pos(double X, double Y, double Z, unsigned nvXp2, unsigned nvYp2, double voxelSize, double l,
                       uint64_t& id, uint16_t& sx, uint16_t& sy, uint16_t& sz) {
    const uint64_t nx = (uint64_t)(X / voxelSize);
    const uint64_t ny = (uint64_t)(Y / voxelSize);
    const uint64_t nz = (uint64_t)(Z / voxelSize);
    sx = (uint16_t)((X - (double)nx * voxelSize) / l);
    sy = (uint16_t)((Y - (double)ny * voxelSize) / l);
    sz = (uint16_t)((Z - (double)nz * voxelSize) / l);
    id = nx;
    id += ny << nvXp2;
    id += nz << (nvXp2 + nvYp2);
}

// ---------------------------------------------------------------------------
// Quaternion rotation.  kernel/DEMHelperKernels.cuh:161-173 (applyOriQToVector3):
// nine coefficients formed in the quaternion's type (float), then a 3-term sum
// evaluated left to right in the vector's type.
// ---------------------------------------------------------------------------
struct RotM {
    float xx, xy, xz, yx, yy, yz, zx, zy, zz;
};
inline RotM rot_coeffs(float w, float x, float y, float z) {
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
inline V3f rotate_f(const RotM& m, V3f v) {
    V3f r;
    r.x = m.xx * v.x + m.xy * v.y + m.xz * v.z;
    r.y = m.yx * v.x + m.yy * v.y + m.yz * v.z;
    r.z = m.zx * v.x + m.zy * v.y + m.zz * v.z;
    return r;
}
inline V3d rotate_d(const RotM& m, V3d v) {  // <double, float> instantiation (triangle nodes)
    V3d r;
    r.x = m.xx * v.x + m.xy * v.y + m.xz * v.z;
    r.y = m.yx * v.x + m.yy * v.y + m.yz * v.z;
    r.z = m.zx * v.x + m.zy * v.y + m.zz * v.z;
    return r;
}
inline V3f rotate_q(Q4 q, V3f v) { return rotate_f(rot_coeffs(q.w, q.x, q.y, q.z), v); }
inline V3f rotate_q_inv(Q4 q, V3f v) { return rotate_f(rot_coeffs(q.w, -q.x, -q.y, -q.z), v); }

// kernel/DEMHelperKernels.cuh:228-245 (HamiltonProduct)
inline void hamilton(float& A, float& B, float& C, float& D, float a1, float b1, float c1, float d1, float a2,
                     float b2, float c2, float d2) {
    A = a1 * a2 - b1 * b2 - c1 * c2 - d1 * d2;
    B = a1 * b2 + b1 * a2 + c1 * d2 - d1 * c2;
    C = a1 * c2 - b1 * d2 + c1 * a2 + d1 * b2;
    D = a1 * d2 + b1 * c2 - c1 * b2 + d1 * a2;
}

// kernel/DEMHelperKernels.cuh:57-62 (locateMaskPair): upper-triangular, column-major
inline unsigned mask_pair(unsigned i, unsigned j) {
    if (i > j)
        std::swap(i, j);
    return (1 + j) * j / 2 + i;
}

// kernel/DEMHelperKernels.cuh:329-336 (getPointBinID): truncating conversion per axis
inline uint32_t point_bin(double X, double Y, double Z, double binSize, uint32_t nbX, uint32_t nbY) {
    const uint32_t bx = (uint32_t)(X / binSize);
    const uint32_t by = (uint32_t)(Y / binSize);
    const uint32_t bz = (uint32_t)(Z / binSize);
    return bx + by * nbX + bz * nbX * nbY;
}

// ---------------------------------------------------------------------------
// Sphere-sphere narrow phase.  kernel/DEMHelperKernels.cuh:292-326
// (checkSpheresOverlap<double,float>): distance test in fp64, normal built in
// fp32 from the fp64 differences, contact point back in fp64.
// ---------------------------------------------------------------------------
inline uint8_t spheres_overlap(double XA, double YA, double ZA, double rA, double XB, double YB, double ZB, double rB,
                               double& CPX, double& CPY, double& CPZ, float& nx, float& ny, float& nz,
                               double& depth) {
    const double d2 = (XA - XB) * (XA - XB) + (YA - YB) * (YA - YB) + (ZA - ZB) * (ZA - ZB);
    const uint8_t type = (d2 > (rA + rB) * (rA + rB)) ? DEME_NOT_A_CONTACT : DEME_SPHERE_SPHERE_CONTACT;
    nx = (float)(XA - XB);
    ny = (float)(YA - YB);
    nz = (float)(ZA - ZB);
    const float mag = sqrtf(nx * nx + ny * ny + nz * nz);
    nx /= mag;
    ny /= mag;
    nz /= mag;
    depth = rA + rB - sqrt(d2);
    CPX = XB + (rB - depth / 2.0) * nx;
    CPY = YB + (rB - depth / 2.0) * ny;
    CPZ = ZB + (rB - depth / 2.0) * nz;
    return type;
}

// ---------------------------------------------------------------------------
// Sphere vs analytical entity.  kernel/DEMHelperKernels.cuh:459-521
// (checkSphereEntityOverlap<double3,float,double>).  Note dot(double3,float3)
// returns float (CUDAMathHelpers.cuh:1221), and float3*double narrows the
// scalar to float (CUDAMathHelpers.cuh:674).
// ---------------------------------------------------------------------------
inline uint8_t sphere_entity(V3d A, float radA, uint8_t typeB, V3d B, V3f dirB, float size1, float /*size2*/,
                             float /*size3*/, float normal_sign, float beta4Entity, V3d& CP, V3f& nrm, double& depth) {
    switch (typeB) {
        case DEME_ANAL_OBJ_TYPE_PLANE: {
            const V3d p2s{A.x - B.x, A.y - B.y, A.z - B.z};
            const double dist = (float)(p2s.x * dirB.x + p2s.y * dirB.y + p2s.z * dirB.z);
            depth = (radA + beta4Entity - dist);
            const uint8_t t = (depth < 0.0) ? DEME_NOT_A_CONTACT : DEME_SPHERE_PLANE_CONTACT;
            const float s = (float)(dist + depth / 2.0);
            const V3f off = dirB * s;
            CP = {A.x - (double)off.x, A.y - (double)off.y, A.z - (double)off.z};
            nrm = dirB;
            return t;
        }
        case DEME_ANAL_OBJ_TYPE_CYL_INF: {
            V3d s2c{B.x - A.x, B.y - A.y, B.z - A.z};
            const double proj = (float)(s2c.x * dirB.x + s2c.y * dirB.y + s2c.z * dirB.z);
            // sph2cyl -= proj_dist * dirB : double * float3 narrows to float*float3, then double3 -= float3
            const V3f pd = (float)proj * dirB;
            s2c.x -= pd.x;
            s2c.y -= pd.y;
            s2c.z -= pd.z;
            const double dr = sqrt(s2c.x * s2c.x + s2c.y * s2c.y + s2c.z * s2c.z);
            const float cyl_rad = size1 - normal_sign * beta4Entity;
            depth = radA - normal_sign * (cyl_rad - dr);
            const uint8_t t = (depth < 0.0) ? DEME_NOT_A_CONTACT : DEME_SPHERE_CYL_CONTACT;
            if (dr >= kTiny) {
                // normal_sign / dist_delta_r * sph2cyl : (double) * double3, then narrowed to float3
                const double f = normal_sign / dr;
                nrm = {(float)(f * s2c.x), (float)(f * s2c.y), (float)(f * s2c.z)};
                const float s = (float)(radA - depth / 2.0);
                const V3f off = nrm * s;
                CP = {A.x - (double)off.x, A.y - (double)off.y, A.z - (double)off.z};
            } else {
                nrm = dirB;
                CP = A;
            }
            return t;
        }
        default:
            return DEME_NOT_A_CONTACT;
    }
}

// kernel/DEMHelperKernels.cuh:433-455 (matProxy2ContactParam<float>)
inline void mat_proxy(float& E_eff, float& G_eff, float Y1, float nu1, float Y2, float nu2) {
    const float invE = (1.0f - nu1 * nu1) / Y1 + (1.0f - nu2 * nu2) / Y2;
    E_eff = 1.0f / invE;
    const float invG = 2.0f * (2.0f - nu1) * (1.0f + nu1) / Y1 + 2.0f * (2.0f - nu2) * (1.0f + nu2) / Y2;
    G_eff = 1.0f / invG;
}

// ---------------------------------------------------------------------------
// Built-in force models.
// kernel/DEMCustomizablePolicies/FullHertzianForceModel.cu:5-135 and
// FrictionlessHertzianForceModel.cu:3-42.  Mixed precision is deliberate in the
// reference: overlapDepth is fp64, most physics fp32, with double literals
// promoting a few products before they are narrowed on assignment.
// Overload note: the fragments call unqualified `sqrt(x)` / `log(x)`.  With the C++ overload set
// in scope (CUDA device code; the pinned oracle/_ref host build, whose translation unit includes
// <math.h> through the reference's own DEMTriangleBoxIntersect.cu) a float argument selects the
// float overload: log(CoR_cnt), sqrt(Sn * mass_eff) and sqrt(mass_eff * kt) are fp32 operations,
// while sqrt(overlapDepth * ...) and sqrt(loge * loge + PI_SQUARED) have double arguments and stay fp64.
// ---------------------------------------------------------------------------
struct ForceIn {
    double overlapDepth;
    V3f B2A;
    float AOwnerMass, BOwnerMass, ARadius, BRadius;
    Q4 AOriQ, BOriQ;
    V3f locCPA, locCPB;
    V3f ALinVel, BLinVel, ARotVel, BRotVel;
    float ts;
    float E_A, nu_A, E_B, nu_B, CoR, mu, Crr;
};
struct ForceHist {
    float delta_tan_x, delta_tan_y, delta_tan_z, delta_time;
};

inline void hertz_common(const ForceIn& in, V3f& rotVelCPA, V3f& rotVelCPB, V3f& velB2A, float& projection) {
    rotVelCPA = rotate_q(in.AOriQ, crossf(in.ARotVel, in.locCPA));
    rotVelCPB = rotate_q(in.BOriQ, crossf(in.BRotVel, in.locCPB));
    velB2A = (in.ALinVel + rotVelCPA) - (in.BLinVel + rotVelCPB);
    projection = dotf(velB2A, in.B2A);
}

inline void force_hertz_full(const ForceIn& in, ForceHist& h, V3f& force, V3f& torque_only) {
    if (in.overlapDepth > 0) {
        float E_cnt, G_cnt;
        mat_proxy(E_cnt, G_cnt, in.E_A, in.nu_A, in.E_B, in.nu_B);
        V3f rotVelCPA, rotVelCPB, velB2A;
        float projection;
        hertz_common(in, rotVelCPA, rotVelCPB, velB2A, projection);
        const V3f vrel_tan = velB2A - projection * in.B2A;
        V3f delta_tan{h.delta_tan_x, h.delta_tan_y, h.delta_tan_z};
        delta_tan = delta_tan + in.ts * vrel_tan;
        const float disp_proj = dotf(delta_tan, in.B2A);
        delta_tan = delta_tan - disp_proj * in.B2A;
        h.delta_time += in.ts;

        const float mass_eff = (in.AOwnerMass * in.BOwnerMass) / (in.AOwnerMass + in.BOwnerMass);
        const float sqrt_Rd =
            (float)sqrt(in.overlapDepth * (double)(in.ARadius * in.BRadius) / (double)(in.ARadius + in.BRadius));
        const float Sn = (float)(2. * E_cnt * sqrt_Rd);
        const float loge = (float)((in.CoR < kTiny) ? log(kTiny) : logf(in.CoR));
        const float beta = (float)(loge / sqrt(loge * loge + kPiSq));
        const float k_n = (float)(kTwoThirds * Sn);
        const float gamma_n = (float)(kTwoSqrt56 * beta * sqrtf(Sn * mass_eff));
        force = force + (float)(k_n * in.overlapDepth + gamma_n * projection) * in.B2A;

        if (in.Crr > 0.0) {
            bool roll = true;
            const float R_eff = sqrtf((in.ARadius * in.BRadius) / (in.ARadius + in.BRadius));
            const float kn_simple = (float)(kFourThirds * E_cnt * sqrtf(R_eff));
            const float gn_simple =
                -2.f * sqrtf((float)(kFiveThirds * mass_eff * E_cnt)) * beta * powf(R_eff, 0.25f);
            const float d_coeff = gn_simple / (2.f * sqrtf(kn_simple * mass_eff));
            if (d_coeff < 1.0) {
                const float t_collision = (float)(kPi * sqrtf(mass_eff / (kn_simple * (1.f - d_coeff * d_coeff))));
                if (h.delta_time <= t_collision)
                    roll = false;
            }
            if (roll) {
                const V3f v_rot = rotVelCPB - rotVelCPA;
                const float v_rot_mag = lenf(v_rot);
                if (v_rot_mag > kTiny)
                    torque_only = (v_rot / v_rot_mag) * (in.Crr * lenf(force));
            }
        }

        if (in.mu > 0.0) {
            const float kt = (float)(8. * G_cnt * sqrt_Rd);
            const float gt = (float)(-kTwoSqrt56 * beta * sqrtf(mass_eff * kt));
            V3f tangent_force = (-kt) * delta_tan - gt * vrel_tan;
            const float ft = lenf(tangent_force);
            if (ft > kTiny) {
                const float ft_max = lenf(force) * in.mu;
                if (ft > ft_max) {
                    tangent_force = (ft_max / ft) * tangent_force;
                    delta_tan = (tangent_force + gt * vrel_tan) / (-kt);
                }
            } else {
                tangent_force = {0, 0, 0};
            }
            force = force + tangent_force;
        }
        h.delta_tan_x = delta_tan.x;
        h.delta_tan_y = delta_tan.y;
        h.delta_tan_z = delta_tan.z;
    } else {
        h.delta_time = 0;
        h.delta_tan_x = 0;
        h.delta_tan_y = 0;
        h.delta_tan_z = 0;
    }
}

inline void force_hertz_frictionless(const ForceIn& in, V3f& force) {
    if (in.overlapDepth > 0) {
        // matProxy2ContactParam<float> no-tangent overload, DEMHelperKernels.cuh:447-455
        const float invE = (1.0f - in.nu_A * in.nu_A) / in.E_A + (1.0f - in.nu_B * in.nu_B) / in.E_B;
        const float E_cnt = 1.0f / invE;
        V3f rotVelCPA, rotVelCPB, velB2A;
        float projection;
        hertz_common(in, rotVelCPA, rotVelCPB, velB2A, projection);
        const float mass_eff = (in.AOwnerMass * in.BOwnerMass) / (in.AOwnerMass + in.BOwnerMass);
        const float sqrt_Rd =
            (float)sqrt(in.overlapDepth * (double)(in.ARadius * in.BRadius) / (double)(in.ARadius + in.BRadius));
        const float Sn = (float)(2. * E_cnt * sqrt_Rd);
        const float loge = (float)((in.CoR < kTiny) ? log(kTiny) : logf(in.CoR));
        const float beta = (float)(loge / sqrt(loge * loge + kPiSq));
        const float k_n = (float)(kTwoThirds * Sn);
        const float gamma_n = (float)(kTwoSqrt56 * beta * sqrtf(Sn * mass_eff));
        force = force + (float)(k_n * in.overlapDepth + gamma_n * projection) * in.B2A;
    }
}

// ---------------------------------------------------------------------------
// Triangle-sphere path.  PARITY UNPINNED: kernel/DEMCollisionKernels.cu is __device__-only (round-up
// intrinsics) and kernel/DEMTriangleBoxIntersect.cu / DEMBinTriangleKernels.cu are __device__-only or
// carry placeholders, so none of them can be built on the host here; these restatements are checked
// only against analytic cases (tests/test_oracle_mesh.py).
// ---------------------------------------------------------------------------
// __drcp_ru / __dmul_ru (kernel/DEMCollisionKernels.cu:76-78): round-toward-+inf reciprocal and product,
// emulated with one fma-based correction step (exact for finite, non-zero, non-subnormal operands).
inline double rcp_ru(double x) {
    double r = 1.0 / x;
    const double e = fma(-x, r, 1.0);  // 1 - x*r, exact sign of the rounding error
    // true value 1/x = r + e/x: below means r < 1/x -> bump toward +inf
    if ((e > 0.0 && x > 0.0) || (e < 0.0 && x < 0.0))
        r = nextafter(r, INFINITY);
    return r;
}
inline double mul_ru(double a, double b) {
    double p = a * b;
    const double e = fma(a, b, -p);  // exact a*b - p
    if (e > 0.0)
        p = nextafter(p, INFINITY);
    return p;
}

template <typename T>
struct V3 {
    T x, y, z;
};
template <typename T>
inline V3<T> vsub(V3<T> a, V3<T> b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
template <typename T>
inline V3<T> vadd(V3<T> a, V3<T> b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
template <typename T>
inline V3<T> vscale(T s, V3<T> a) { return {s * a.x, s * a.y, s * a.z}; }
template <typename T>
inline T vdot(V3<T> a, V3<T> b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <typename T>
inline V3<T> vcross(V3<T> a, V3<T> b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
// normalize(): float3 -> v * rsqrtf(dot); double3 -> v * (double)rsqrtf(dot) (CUDAMathHelpers.cuh:1090, 1402);
// host form of rsqrtf = 1.0f / sqrtf(x) (CUDAMathHelpers.cuh:66-68)
template <typename T>
inline V3<T> vnormalize(V3<T> v) {
    const T il = (T)(1.0f / sqrtf((float)vdot(v, v)));
    return {v.x * il, v.y * il, v.z * il};
}
inline float vlen(V3<float> v) { return sqrtf(vdot(v, v)); }
inline double vlen(V3<double> v) { return sqrt(vdot(v, v)); }

// kernel/DEMCollisionKernels.cu:16-82 (snap_to_face; Ericson, Real-Time Collision Detection p.141).
// VT = component type of the points, ST = scalar type of the barycentric arithmetic (the reference's T2:
// float inside triangle_sphere_CD<float3,float>, double when called with defaulted template arguments
// from the bin sweep, DEMContactKernels_SphereTriangle.cu:247).
template <typename VT, typename ST>
inline bool snap_to_face(V3<VT> A, V3<VT> B, V3<VT> C, V3<VT> P, V3<VT>& res) {
    const V3<VT> AB = vsub(B, A), AC = vsub(C, A), AP = vsub(P, A);
    const ST d1 = vdot(AB, AP), d2 = vdot(AC, AP);
    if (d1 <= 0 && d2 <= 0) {
        res = A;
        return true;
    }
    const V3<VT> BP = vsub(P, B);
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
    const V3<VT> CP = vsub(P, C);
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

// kernel/DEMCollisionKernels.cu:99-159 (triangle_sphere_CD) and :177-236 (directional flavour)
template <typename T, bool DIRECTIONAL>
inline bool tri_sphere_cd(V3<T> A, V3<T> B, V3<T> C, V3<T> sp, T radius, V3<T>& normal, T& depth, V3<T>& pt1) {
    const V3<T> face_n = vnormalize(vcross(vsub(B, A), vsub(C, A)));
    const T h = vdot(vsub(sp, A), face_n);
    V3<T> faceLoc;
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
        normal = vscale((T)(1.0 / dist), normal);  // (1.0 / dist) is double, narrowed for float3 (CUDAMathHelpers.cuh:677)
        pt1 = faceLoc;
        if (DIRECTIONAL)
            in_contact = !(depth >= 0. || h >= radius);
        else
            in_contact = !(depth >= 0. || h >= radius || h <= -radius);
    }
    return in_contact;
}

// boundingBoxIntersectBin (DEMHelperKernels.cuh:528-565): bin range of one triangle's bounding box, enlarged by
// DEME_BIN_ENLARGE_RATIO_FOR_FACETS * binSize (float -= double, narrowed), divided in fp64, narrowed to fp32,
// clamped as a float against [0, nb-1] and truncated (clampBetween3Comp<float3,int3>, :73-80).
inline void tri_bbox_bins(const V3<float> v[3], double binSize, const int nbm[3], int L[3], int U[3]) {
    const float mn[3] = {std::min(v[0].x, std::min(v[1].x, v[2].x)), std::min(v[0].y, std::min(v[1].y, v[2].y)),
                         std::min(v[0].z, std::min(v[1].z, v[2].z))};
    const float mx[3] = {std::max(v[0].x, std::max(v[1].x, v[2].x)), std::max(v[0].y, std::max(v[1].y, v[2].y)),
                         std::max(v[0].z, std::max(v[1].z, v[2].z))};
    for (int d = 0; d < 3; d++) {
        const float lo = (float)(mn[d] - 0.001 * binSize), hi = (float)(mx[d] + 0.001 * binSize);
        const float ql = (float)(lo / binSize), qh = (float)(hi / binSize);
        const float cl = std::min(std::max(ql, 0.f), (float)nbm[d]), ch = std::min(std::max(qh, 0.f), (float)nbm[d]);
        L[d] = (int)cl;
        U[d] = (int)ch;
    }
}

// kernel/DEMTriangleBoxIntersect.cu:176-374 (Akenine-Moller triangle/AABB separating-axis test, fp32)
inline bool plane_box_overlap(const float n[3], const float vert[3], const float maxbox[3]) {
    float vmin[3], vmax[3];
    for (int q = 0; q < 3; q++) {
        const float v = vert[q];
        if (n[q] > 0.0f) {
            vmin[q] = -maxbox[q] - v;
            vmax[q] = maxbox[q] - v;
        } else {
            vmin[q] = maxbox[q] - v;
            vmax[q] = -maxbox[q] - v;
        }
    }
    if (n[0] * vmin[0] + n[1] * vmin[1] + n[2] * vmin[2] > 0.0f)
        return false;
    return n[0] * vmax[0] + n[1] * vmax[1] + n[2] * vmax[2] >= 0.0f;
}
inline bool axis_sep(float pa, float pb, float rad) {
    const float mn = (pa < pb) ? pa : pb, mx = (pa < pb) ? pb : pa;
    return mn > rad || mx < -rad;
}
using T3f = V3<float>;
using T3d = V3<double>;
inline bool tri_box_overlap(const float bc[3], const float bh[3], T3f vA, T3f vB, T3f vC) {
    const float v0[3] = {vA.x - bc[0], vA.y - bc[1], vA.z - bc[2]};
    const float v1[3] = {vB.x - bc[0], vB.y - bc[1], vB.z - bc[2]};
    const float v2[3] = {vC.x - bc[0], vC.y - bc[1], vC.z - bc[2]};
    const float e0[3] = {v1[0] - v0[0], v1[1] - v0[1], v1[2] - v0[2]};
    const float e1[3] = {v2[0] - v1[0], v2[1] - v1[1], v2[2] - v1[2]};
    const float e2[3] = {v0[0] - v2[0], v0[1] - v2[1], v0[2] - v2[2]};
    float fex, fey, fez;
    // edge 0: X01, Y02, Z12
    fex = fabsf(e0[0]), fey = fabsf(e0[1]), fez = fabsf(e0[2]);
    if (axis_sep(e0[2] * v0[1] - e0[1] * v0[2], e0[2] * v2[1] - e0[1] * v2[2], fez * bh[1] + fey * bh[2])) return false;
    if (axis_sep(-e0[2] * v0[0] + e0[0] * v0[2], -e0[2] * v2[0] + e0[0] * v2[2], fez * bh[0] + fex * bh[2])) return false;
    if (axis_sep(e0[1] * v1[0] - e0[0] * v1[1], e0[1] * v2[0] - e0[0] * v2[1], fey * bh[0] + fex * bh[1])) return false;
    // edge 1: X01, Y02, Z0
    fex = fabsf(e1[0]), fey = fabsf(e1[1]), fez = fabsf(e1[2]);
    if (axis_sep(e1[2] * v0[1] - e1[1] * v0[2], e1[2] * v2[1] - e1[1] * v2[2], fez * bh[1] + fey * bh[2])) return false;
    if (axis_sep(-e1[2] * v0[0] + e1[0] * v0[2], -e1[2] * v2[0] + e1[0] * v2[2], fez * bh[0] + fex * bh[2])) return false;
    if (axis_sep(e1[1] * v0[0] - e1[0] * v0[1], e1[1] * v1[0] - e1[0] * v1[1], fey * bh[0] + fex * bh[1])) return false;
    // edge 2: X2, Y1, Z12
    fex = fabsf(e2[0]), fey = fabsf(e2[1]), fez = fabsf(e2[2]);
    if (axis_sep(e2[2] * v0[1] - e2[1] * v0[2], e2[2] * v1[1] - e2[1] * v1[2], fez * bh[1] + fey * bh[2])) return false;
    if (axis_sep(-e2[2] * v0[0] + e2[0] * v0[2], -e2[2] * v1[0] + e2[0] * v1[2], fez * bh[0] + fex * bh[2])) return false;
    if (axis_sep(e2[1] * v1[0] - e2[0] * v1[1], e2[1] * v2[0] - e2[0] * v2[1], fey * bh[0] + fex * bh[1])) return false;
    // AABB of the triangle against the box
    for (int d = 0; d < 3; d++) {
        float mn = v0[d], mx = v0[d];
        if (v1[d] < mn) mn = v1[d];
        if (v1[d] > mx) mx = v1[d];
        if (v2[d] < mn) mn = v2[d];
        if (v2[d] > mx) mx = v2[d];
        if (mn > bh[d] || mx < -bh[d])
            return false;
    }
    const float nrm[3] = {e0[1] * e1[2] - e0[2] * e1[1], e0[2] * e1[0] - e0[0] * e1[2], e0[0] * e1[1] - e0[1] * e1[0]};
    return plane_box_overlap(nrm, v0, bh);
}

// kernel/DEMBinTriangleKernels.cu:7-20 (sandwichVertex) and DEMHelperKernels.cuh:215-225 (triangleIncenter)
inline T3f sandwich_vertex(T3f vertex, T3f incenter, T3f side, T3f normal, float beta) {
    const T3f ev = vnormalize(vsub(vertex, incenter));
    const T3f nev{-ev.x, -ev.y, -ev.z};
    const float cos_half = vdot(nev, side) / vlen(side);
    const float enlarge = (float)(beta / sqrt(1. - cos_half * cos_half));
    vertex = vadd(vertex, vscale(enlarge, ev));  // expandVec * enlarge_dist: float3 * float
    vertex = vadd(vertex, vscale(beta, normal));
    return vertex;
}
inline T3f tri_incenter(T3f p1, T3f p2, T3f p3) {
    const float a = vlen(vsub(p2, p3)), b = vlen(vsub(p1, p3)), c = vlen(vsub(p1, p2));
    return {(a * p1.x + b * p2.x + c * p3.x) / (a + b + c), (a * p1.y + b * p2.y + c * p3.y) / (a + b + c),
            (a * p1.z + b * p2.z + c * p3.z) / (a + b + c)};
}

// ===========================================================================
// Simulation object mirroring the C-ABI context, one stage per entry point.
// ===========================================================================
struct Key {
    uint32_t a, b;
    uint8_t t;
};

struct Sim {
    DemeParams p{};
    uint32_t nOwners = 0, nOwnerClumps = 0, nSpheres = 0, nAnal = 0, nMat = 0, nComp = 0, nMassProps = 0;
    // owner state
    std::vector<uint64_t> voxelID;
    std::vector<uint16_t> locX, locY, locZ;
    std::vector<float> oriQw, oriQx, oriQy, oriQz, vX, vY, vZ, omgX, omgY, omgZ, aX, aY, aZ, alX, alY, alZ;
    std::vector<uint8_t> familyID;
    std::vector<uint8_t> ghost;  // per owner: copy of a clump another rank owns (DemeScene.ownerGhost); empty = none
    std::vector<uint16_t> inertiaOff;
    std::vector<float> margin;
    // spheres
    std::vector<uint32_t> ownerOfSphere;
    std::vector<uint16_t> compOff, sphMat;
    // triangles (mesh-major, owner-local nodes)
    uint32_t nTri = 0;
    std::vector<uint32_t> ownerMesh;
    std::vector<float> tri1, tri2, tri3;  // xyz interleaved
    std::vector<uint16_t> triMat;
    std::vector<uint32_t> triIncBin, triIncTri;  // bin-sorted (stable) triangle incidence list
    // tables
    std::vector<float> Radii, relX, relY, relZ, mass, moiX, moiY, moiZ;
    std::vector<uint8_t> objType;
    std::vector<uint32_t> objOwner;
    std::vector<float> objNormal, objRelX, objRelY, objRelZ, objRotX, objRotY, objRotZ, objS1, objS2, objS3, objMass;
    std::vector<uint16_t> objMat;
    std::vector<float> E, nu, CoR, mu, Crr;
    std::vector<uint8_t> masks, famFlags;
    std::vector<float> famExtra;
    // detection products
    std::vector<double> sphX, sphY, sphZ;
    std::vector<float> sphR;
    std::vector<uint32_t> incBin, incSph;  // bin-sorted (stable) incidence list
    uint64_t nActiveBins = 0;
    uint32_t maxInBin = 0;
    std::vector<uint32_t> cA, cB, cMap;
    std::vector<uint8_t> cType;
    std::vector<uint32_t> pA, pB;  // previous list (history source)
    std::vector<uint8_t> pType;
    std::vector<float> wc[DEME_MAX_WILDCARD_NUM];
    // DEME_FORCE_CUSTOM in a parametric form (the oracle cannot take a C++ fragment): customKind 1 = the fragment of
    // kernel/DEMUserScripts/ForceModelWithCohesion.cu's normal part as bench.py's configs[4] flavour writes it -- frictionless
    // Hertz + a pairwise `Cohesion` material property pulling along -B2A + one contact wildcard counting the contact's age
    int customKind = 0;
    std::vector<float> cohesion;  // nMat * nMat
    // per-contact records
    std::vector<float> recF, recT, recCPA, recCPB;
    uint64_t nSteps = 0, nDetections = 0;
    uint32_t stepsSinceCD = 0;
    bool haveList = false;
    bool seeded = false;  // list loaded by orc_sim_seed_contacts: only feeds the next history map
    // family motion prescriptions in a parametric test form: quantity k of family f = c0 + c1*t + c2*sinf(c3*t) (fp32), the
    // same expression the GPU tests hand to the run-time compiler as a string.  k: 0-2 v, 3-5 omgBar, 6-8 position,
    // 9-11 added linear acc, 12-14 added angular acc.  `has` bit k: quantity k is assigned; `flags`: PrescRec::flags bits.
    struct Presc {
        bool used = false;
        uint32_t has = 0, flags = 0;
        float c[15][4] = {};
    };
    std::vector<Presc> presc = std::vector<Presc>(256);
    // on-the-fly family changes in a parametric test form: owners of family `from` whose quantity q
    // (0-2 X,Y,Z; 3-5 vX,vY,vZ; 6-8 accX,accY,accZ; 9 time) is > (op 0) or < (op 1) thr move to family `to`
    struct FamRule {
        uint32_t from, to, q, op;
        double thr;
    };
    std::vector<FamRule> famRules;
    // persistent contacts (DEM/APIPrivate.cpp:33-117 marks them in the previous list; algorithms/DEMCubContactDetection.cu:
    // 605-802 adds every marked contact the detection did not find back into the new list): kept here as the sorted set of
    // marked (A, class, B) keys, which is what the flag array travelling with the list amounts to
    std::vector<Key> persist;
    std::vector<float> volume;  // per mass-property entry ("clump_volume" inspector)
    // DEMTracker::AddAcc / AddAngAcc (dT.cpp:3160-3174): added to a / alpha of the coming step only, after the contact sums
    std::vector<float> nextAcc;  // 6 per owner, empty when nothing is pending
};

template <typename T>
void assign(std::vector<T>& v, const T* src, size_t n) {
    if (src)
        v.assign(src, src + n);
    else
        v.assign(n, T(0));
}

// Sphere world position in the LBF-shifted frame plus the two inflated radii the
// reference uses: fp64 sum for binning (kernel/DEMBinSphereKernels.cu:30-48), fp32 sum
// for the sweep (kernel/DEMContactKernels_SphereSphere.cu:31-54).
inline void sphere_geometry(const Sim& s, uint32_t sph, double& X, double& Y, double& Z, double& rBin, float& rSweep) {
    const uint32_t o = s.ownerOfSphere[sph];
    const uint16_t c = s.compOff[sph];
    double oX, oY, oZ;
    decode_pos(s.voxelID[o], s.locX[o], s.locY[o], s.locZ[o], s.p.nvXp2, s.p.nvYp2, s.p.voxelSize, s.p.l, oX, oY, oZ);
    const V3f rel = rotate_q({s.oriQw[o], s.oriQx[o], s.oriQy[o], s.oriQz[o]}, {s.relX[c], s.relY[c], s.relZ[c]});
    X = oX + (double)rel.x;
    Y = oY + (double)rel.y;
    Z = oZ + (double)rel.z;
    rBin = (double)s.Radii[c];
    rBin += s.margin[o];
    rSweep = s.Radii[c];
    rSweep += s.margin[o];
}

// Bin range of a sphere along one axis.  kernel/DEMBinSphereKernels.cu:51-69, 181-196.
inline void bin_range(double pos, double radius, double binSize, uint32_t nb, uint32_t& lo, uint32_t& hi) {
    const double b = pos / binSize;
    const double span = radius / binSize;
    hi = (b + span < (double)nb) ? (uint32_t)(b + span) : nb - 1;
    lo = (uint32_t)((b - span > 0.0) ? b - span : 0.0);
}

// kernel/DEMMiscKernels.cu:37-61 (computeMarginFromAbsv) with absv from
// DEM/AuxClasses.cpp:54-61 (fp64 norm of the fp32 velocity, stored as float).
void compute_margins(Sim& s, uint32_t drift) {
    for (uint32_t o = 0; o < s.nOwners; o++) {
        const double vx = s.vX[o], vy = s.vY[o], vz = s.vZ[o];
        float absv = (float)sqrt(vx * vx + vy * vy + vz * vz);
        if (absv > s.p.approxMaxVel)
            absv = s.p.approxMaxVel;
        s.margin[o] = (float)((double)(absv * s.p.expSafetyMulti + s.p.expSafetyAdder) * (double)s.p.h * (double)drift +
                              (double)s.famExtra[s.familyID[o]]);
    }
}

inline int type_class(uint8_t t) {  // within one sphere A: sphere-sphere, sphere-mesh, sphere-analytical
    return t == DEME_SPHERE_SPHERE_CONTACT ? 0 : (t == DEME_SPHERE_MESH_CONTACT ? 1 : 2);
}
// canonical list order of this build: by sphere A, then type class, then B.  (The reference's final
// order is type-major then A, DEMCubContactDetection.cu:1045-1051; only the SET is comparable with it.)
inline bool key_less(const Key& x, const Key& y) {
    if (x.a != y.a)
        return x.a < y.a;
    const int cx = type_class(x.t), cy = type_class(y.t);
    if (cx != cy)
        return cx < cy;
    return x.b < y.b;
}

int detect(Sim& s) {
    const uint32_t nS = s.nSpheres;
    s.sphX.resize(nS);
    s.sphY.resize(nS);
    s.sphZ.resize(nS);
    s.sphR.resize(nS);
    std::vector<double> rBin(nS);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)nS; i++)
        sphere_geometry(s, (uint32_t)i, s.sphX[i], s.sphY[i], s.sphZ[i], rBin[i], s.sphR[i]);

    // (bin, sphere) incidences in sphere order, z-y-x loop nest: DEMBinSphereKernels.cu:181-203
    std::vector<uint32_t> ub, us;
#ifdef ORC_PERF
    {   // count per sphere, prefix, fill: the same list in the same order, built by the whole team
        std::vector<size_t> first((size_t)nS + 1, 0);
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < (int64_t)nS; i++) {
            uint32_t lx, hx, ly, hy, lz, hz;
            bin_range(s.sphX[i], rBin[i], s.p.binSize, s.p.nbX, lx, hx);
            bin_range(s.sphY[i], rBin[i], s.p.binSize, s.p.nbY, ly, hy);
            bin_range(s.sphZ[i], rBin[i], s.p.binSize, s.p.nbZ, lz, hz);
            hx = std::min(hx, s.p.nbX - 1), hy = std::min(hy, s.p.nbY - 1), hz = std::min(hz, s.p.nbZ - 1);
            first[i + 1] = (lx <= hx && ly <= hy && lz <= hz) ? (size_t)(hx - lx + 1) * (hy - ly + 1) * (hz - lz + 1) : 0;
        }
        for (size_t i = 0; i < nS; i++)
            first[i + 1] += first[i];
        ub.resize(first[nS]);
        us.resize(first[nS]);
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < (int64_t)nS; i++) {
            uint32_t lx, hx, ly, hy, lz, hz;
            bin_range(s.sphX[i], rBin[i], s.p.binSize, s.p.nbX, lx, hx);
            bin_range(s.sphY[i], rBin[i], s.p.binSize, s.p.nbY, ly, hy);
            bin_range(s.sphZ[i], rBin[i], s.p.binSize, s.p.nbZ, lz, hz);
            size_t w = first[i];
            for (uint32_t k = lz; k <= hz && k < s.p.nbZ; k++)
                for (uint32_t j = ly; j <= hy && j < s.p.nbY; j++)
                    for (uint32_t ii = lx; ii <= hx && ii < s.p.nbX; ii++) {
                        ub[w] = ii + j * s.p.nbX + k * s.p.nbX * s.p.nbY;
                        us[w++] = (uint32_t)i;
                    }
        }
    }
#else
    ub.reserve((size_t)nS * 4);
    us.reserve((size_t)nS * 4);
    for (uint32_t i = 0; i < nS; i++) {
        uint32_t lx, hx, ly, hy, lz, hz;
        bin_range(s.sphX[i], rBin[i], s.p.binSize, s.p.nbX, lx, hx);
        bin_range(s.sphY[i], rBin[i], s.p.binSize, s.p.nbY, ly, hy);
        bin_range(s.sphZ[i], rBin[i], s.p.binSize, s.p.nbZ, lz, hz);
        for (uint32_t k = lz; k <= hz && k < s.p.nbZ; k++)
            for (uint32_t j = ly; j <= hy && j < s.p.nbY; j++)
                for (uint32_t ii = lx; ii <= hx && ii < s.p.nbX; ii++) {
                    ub.push_back(ii + j * s.p.nbX + k * s.p.nbX * s.p.nbY);  // binIDFrom3Indices, DEMHelperKernels.cuh:339-347
                    us.push_back(i);
                }
    }
#endif
    // stable sort by bin: DEMCubContactDetection.cu:181 (radix sort => stable)
    const size_t P = ub.size();
    std::vector<uint32_t> perm(P);
    for (size_t i = 0; i < P; i++)
        perm[i] = (uint32_t)i;
    ORC_STABLE_SORT(perm.begin(), perm.end(), [&](uint32_t x, uint32_t y) { return ub[x] < ub[y]; });
    s.incBin.resize(P);
    s.incSph.resize(P);
#ifdef ORC_PERF
#pragma omp parallel for schedule(static)
#endif
    for (int64_t i = 0; i < (int64_t)P; i++) {
        s.incBin[i] = ub[perm[i]];
        s.incSph[i] = us[perm[i]];
    }
    // bin segments
    std::vector<size_t> segStart;
    for (size_t i = 0; i < P; i++)
        if (i == 0 || s.incBin[i] != s.incBin[i - 1])
            segStart.push_back(i);
    segStart.push_back(P);
    s.nActiveBins = segStart.size() - 1;
    s.maxInBin = 0;
    for (size_t b = 0; b + 1 < segStart.size(); b++)
        s.maxInBin = std::max<uint32_t>(s.maxInBin, (uint32_t)(segStart[b + 1] - segStart[b]));
    if (s.maxInBin > s.p.errOutBinSphNum)
        return DEME_ERR_BIN_TOO_FULL;

    std::vector<Key> keys;
    // sphere vs analytical: DEMBinSphereKernels.cu:76-130 / 216-276 (margin-inflated on both sides)
    for (uint32_t i = 0; i < nS; i++) {
        const uint32_t o = s.ownerOfSphere[i];
        const unsigned famS = s.familyID[o];
        for (uint32_t ob = 0; ob < s.nAnal; ob++) {
            const uint32_t bo = s.objOwner[ob];
            const unsigned famO = s.familyID[bo];
            if (s.masks[mask_pair(famS, famO)] != 0)
                continue;
            double oX, oY, oZ;
            decode_pos(s.voxelID[bo], s.locX[bo], s.locY[bo], s.locZ[bo], s.p.nvXp2, s.p.nvYp2, s.p.voxelSize, s.p.l,
                       oX, oY, oZ);
            const RotM m = rot_coeffs(s.oriQw[bo], s.oriQx[bo], s.oriQy[bo], s.oriQz[bo]);
            const V3f rp = rotate_f(m, {s.objRelX[ob], s.objRelY[ob], s.objRelZ[ob]});
            const V3f rd = rotate_f(m, {s.objRotX[ob], s.objRotY[ob], s.objRotZ[ob]});
            const V3d B{oX + (double)rp.x, oY + (double)rp.y, oZ + (double)rp.z};
            V3d cp;
            V3f nr;
            double depth;
            const uint8_t t = sphere_entity({s.sphX[i], s.sphY[i], s.sphZ[i]}, (float)rBin[i], s.objType[ob], B, rd,
                                            s.objS1[ob], s.objS2[ob], s.objS3[ob], s.objNormal[ob], s.margin[bo], cp,
                                            nr, depth);
            const double thres = (s.famExtra[famS] < s.famExtra[famO]) ? s.famExtra[famS] : s.famExtra[famO];
            if (t && depth > thres)
                keys.push_back({i, ob, t});
        }
    }
    // per-bin all-pairs sweep with the contact-point-in-this-bin rule:
    // DEMContactKernels_SphereSphere.cu:57-89 (calcContactPoint), :153-216
    const size_t nBins = segStart.size() - 1;
    std::vector<std::vector<Key>> perThread;
    int nthreads = 1;
#ifdef _OPENMP
    nthreads = omp_get_max_threads();
#endif
    perThread.resize(nthreads);
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t b = 0; b < (int64_t)nBins; b++) {
        int tid = 0;
#ifdef _OPENMP
        tid = omp_get_thread_num();
#endif
        const size_t s0 = segStart[b], s1 = segStart[b + 1];
        const uint32_t bin = s.incBin[s0];
        for (size_t x = s0; x < s1; x++) {
            const uint32_t A = s.incSph[x];
            const uint32_t oA = s.ownerOfSphere[A];
            const unsigned fA = s.familyID[oA];
            for (size_t y = x + 1; y < s1; y++) {
                const uint32_t Bs = s.incSph[y];
                const uint32_t oB = s.ownerOfSphere[Bs];
                if (oA == oB)
                    continue;
                const unsigned fB = s.familyID[oB];
                if (!s.ghost.empty() && (s.ghost[oA] & 1) && (s.ghost[oB] & 1))
                    continue;  // both are copies of clumps other ranks own: the pair is theirs
                if (s.masks[mask_pair(fA, fB)] != 0)
                    continue;
                double cx, cy, cz, depth;
                float nx, ny, nz;
                bool in = spheres_overlap(s.sphX[A], s.sphY[A], s.sphZ[A], (double)s.sphR[A], s.sphX[Bs], s.sphY[Bs],
                                          s.sphZ[Bs], (double)s.sphR[Bs], cx, cy, cz, nx, ny, nz, depth) != 0;
                const float am = (s.famExtra[fA] < s.famExtra[fB]) ? s.famExtra[fA] : s.famExtra[fB];
                in = in && (depth > (double)am);
                if (in && point_bin(cx, cy, cz, s.p.binSize, s.p.nbX, s.p.nbY) == bin)
                    perThread[tid].push_back({A, Bs, DEME_SPHERE_SPHERE_CONTACT});  // A < B: stable sort keeps sphere order
            }
        }
    }
    // ---- sphere vs triangle (kernel/DEMBinTriangleKernels.cu:22-201, DEMContactKernels_SphereTriangle.cu:116-258)
    s.triIncBin.clear();
    s.triIncTri.clear();
    if (s.nTri) {
        struct TriW {
            T3f a1, a2, a3, b1, b2, b3;
        };
        std::vector<TriW> tw(s.nTri);
        std::vector<uint32_t> tb, tt;
        const float bhs = (float)(s.p.binSize / 2. + 0.001 * s.p.binSize);  // DEME_BIN_ENLARGE_RATIO_FOR_FACETS
        const float bh[3] = {bhs, bhs, bhs};
        for (uint32_t t = 0; t < s.nTri; t++) {
            const uint32_t o = s.ownerMesh[t];
            const T3f p1{s.tri1[3 * t], s.tri1[3 * t + 1], s.tri1[3 * t + 2]};
            const T3f p2{s.tri2[3 * t], s.tri2[3 * t + 1], s.tri2[3 * t + 2]};
            const T3f p3{s.tri3[3 * t], s.tri3[3 * t + 1], s.tri3[3 * t + 2]};
            const T3f inc = tri_incenter(p1, p2, p3);
            const T3f n = vnormalize(vcross(vsub(p2, p1), vsub(p3, p1)));  // face_normal, DEMHelperKernels.cuh:352
            const T3f nn{-n.x, -n.y, -n.z};
            const float beta = s.margin[o];
            T3f loc[6] = {sandwich_vertex(p1, inc, vsub(p2, p1), n, beta), sandwich_vertex(p2, inc, vsub(p3, p2), n, beta),
                          sandwich_vertex(p3, inc, vsub(p1, p3), n, beta), sandwich_vertex(p1, inc, vsub(p2, p1), nn, beta),
                          sandwich_vertex(p3, inc, vsub(p1, p3), nn, beta), sandwich_vertex(p2, inc, vsub(p3, p2), nn, beta)};
            double oX, oY, oZ;
            decode_pos(s.voxelID[o], s.locX[o], s.locY[o], s.locZ[o], s.p.nvXp2, s.p.nvYp2, s.p.voxelSize, s.p.l, oX, oY, oZ);
            const RotM m = rot_coeffs(s.oriQw[o], s.oriQx[o], s.oriQy[o], s.oriQz[o]);
            T3f w[6];
            for (int k = 0; k < 6; k++) {
                const V3f r = rotate_f(m, {loc[k].x, loc[k].y, loc[k].z});
                w[k] = {(float)(oX + r.x), (float)(oY + r.y), (float)(oZ + r.z)};  // double3 + float3 -> float3
            }
            tw[t] = {w[0], w[1], w[2], w[3], w[4], w[5]};
            // boundingBoxIntersectBin (DEMHelperKernels.cuh:528-565) on both sandwich triangles, merged
            int L[3] = {INT32_MAX, INT32_MAX, INT32_MAX}, U[3] = {-1, -1, -1};
            const int nbm[3] = {(int)s.p.nbX - 1, (int)s.p.nbY - 1, (int)s.p.nbZ - 1};
            for (int half = 0; half < 2; half++) {
                int l3[3], u3[3];
                tri_bbox_bins(w + 3 * half, s.p.binSize, nbm, l3, u3);
                for (int d = 0; d < 3; d++) {
                    L[d] = std::min(L[d], l3[d]);
                    U[d] = std::max(U[d], u3[d]);
                }
            }
            for (int i = L[0]; i <= U[0]; i++)
                for (int j = L[1]; j <= U[1]; j++)
                    for (int k = L[2]; k <= U[2]; k++) {
                        const float bc[3] = {(float)(s.p.binSize * i + s.p.binSize / 2.), (float)(s.p.binSize * j + s.p.binSize / 2.),
                                             (float)(s.p.binSize * k + s.p.binSize / 2.)};
                        if (tri_box_overlap(bc, bh, {w[0].x, w[0].y, w[0].z}, {w[1].x, w[1].y, w[1].z}, {w[2].x, w[2].y, w[2].z}) ||
                            tri_box_overlap(bc, bh, {w[3].x, w[3].y, w[3].z}, {w[4].x, w[4].y, w[4].z}, {w[5].x, w[5].y, w[5].z})) {
                            tb.push_back((uint32_t)i + (uint32_t)j * s.p.nbX + (uint32_t)k * s.p.nbX * s.p.nbY);
                            tt.push_back(t);
                        }
                    }
        }
        const size_t TP = tb.size();
        std::vector<uint32_t> tperm(TP);
        for (size_t i = 0; i < TP; i++)
            tperm[i] = (uint32_t)i;
        std::stable_sort(tperm.begin(), tperm.end(), [&](uint32_t x, uint32_t y) { return tb[x] < tb[y]; });
        s.triIncBin.resize(TP);
        s.triIncTri.resize(TP);
        for (size_t i = 0; i < TP; i++) {
            s.triIncBin[i] = tb[tperm[i]];
            s.triIncTri[i] = tt[tperm[i]];
        }
        // per (bin, triangle) incidence: every sphere registered in that bin
        for (size_t e = 0; e < TP; e++) {
            const uint32_t bin = s.triIncBin[e], t = s.triIncTri[e];
            const auto lo = std::lower_bound(s.incBin.begin(), s.incBin.end(), bin) - s.incBin.begin();
            const auto hi = std::upper_bound(s.incBin.begin(), s.incBin.end(), bin) - s.incBin.begin();
            const uint32_t oT = s.ownerMesh[t];
            const unsigned fT = s.familyID[oT];
            for (auto x = lo; x < hi; x++) {
                const uint32_t sp = s.incSph[x];
                const uint32_t oS = s.ownerOfSphere[sp];
                if (oS == oT)
                    continue;
                const unsigned fS = s.familyID[oS];
                if (s.masks[mask_pair(fS, fT)] != 0)
                    continue;
                const float am = (s.famExtra[fS] < s.famExtra[fT]) ? s.famExtra[fS] : s.famExtra[fT];
                const T3f sph{(float)s.sphX[sp], (float)s.sphY[sp], (float)s.sphZ[sp]};
                T3f cp, nr;
                float depth;
                bool inA = tri_sphere_cd<float, true>(tw[t].a1, tw[t].a2, tw[t].a3, sph, s.sphR[sp], nr, depth, cp);
                inA = inA && (-depth > am);
                bool inB = tri_sphere_cd<float, true>(tw[t].b1, tw[t].b2, tw[t].b3, sph, s.sphR[sp], nr, depth, cp);
                inB = inB && (-depth > am);
                if (inA || inB) {
                    snap_to_face<float, double>(tw[t].a1, tw[t].a2, tw[t].a3, sph, cp);
                    if (point_bin((double)cp.x, (double)cp.y, (double)cp.z, s.p.binSize, s.p.nbX, s.p.nbY) == bin)
                        keys.push_back({sp, t, DEME_SPHERE_MESH_CONTACT});
                }
            }
        }
    }
    for (auto& v : perThread)
        keys.insert(keys.end(), v.begin(), v.end());
    keys.insert(keys.end(), s.persist.begin(), s.persist.end());
    ORC_SORT(keys.begin(), keys.end(), key_less);
    if (!s.persist.empty())  // a marked contact the sweep found as well appears once (markDuplicateContacts)
        keys.erase(std::unique(keys.begin(), keys.end(), [](const Key& x, const Key& y) { return x.a == y.a && x.b == y.b && x.t == y.t; }),
                   keys.end());

    // history map: "same (A, B, type) in the previous list", DEMHistoryMappingKernels.cu:17-61
    s.pA.swap(s.cA);
    s.pB.swap(s.cB);
    s.pType.swap(s.cType);
    const size_t nC = keys.size(), nP = s.pA.size();
    s.cA.resize(nC);
    s.cB.resize(nC);
    s.cType.resize(nC);
    s.cMap.assign(nC, DEME_NULL_MAPPING_PARTNER);
#ifdef ORC_PERF
#pragma omp parallel for schedule(static)
    for (int64_t ii = 0; ii < (int64_t)nC; ii++) {  // every new key looks itself up in the previous (sorted) list
        const size_t i = (size_t)ii;
        s.cA[i] = keys[i].a;
        s.cB[i] = keys[i].b;
        s.cType[i] = keys[i].t;
        if (!s.haveList)
            continue;
        size_t lo = 0, hi = nP;
        while (lo < hi) {
            const size_t mid = (lo + hi) >> 1;
            if (key_less({s.pA[mid], s.pB[mid], s.pType[mid]}, keys[i]))
                lo = mid + 1;
            else
                hi = mid;
        }
        if (lo < nP && s.pA[lo] == keys[i].a && s.pB[lo] == keys[i].b && s.pType[lo] == keys[i].t)
            s.cMap[i] = (uint32_t)lo;
    }
#else
    size_t j = 0;
    for (size_t i = 0; i < nC; i++) {
        s.cA[i] = keys[i].a;
        s.cB[i] = keys[i].b;
        s.cType[i] = keys[i].t;
        if (!s.haveList)
            continue;
        while (j < nP && key_less({s.pA[j], s.pB[j], s.pType[j]}, keys[i]))
            j++;
        if (j < nP && s.pA[j] == keys[i].a && s.pB[j] == keys[i].b && s.pType[j] == keys[i].t)
            s.cMap[i] = (uint32_t)j;
    }
#endif
    s.haveList = true;
    s.seeded = false;
    s.nDetections++;
    return DEME_OK;
}

// kernel/DEMPrepForceKernels.cu:46-68 (rearrangeContactWildcards)
void migrate(Sim& s) {
    const size_t nC = s.cA.size();
    for (uint32_t w = 0; w < s.p.nContactWildcards; w++) {
        std::vector<float> nw(nC, 0.f);
#ifdef ORC_PERF
#pragma omp parallel for schedule(static)
#endif
        for (int64_t i = 0; i < (int64_t)nC; i++)
            if (s.cMap[i] != DEME_NULL_MAPPING_PARTNER && s.cMap[i] < s.wc[w].size())
                nw[i] = s.wc[w][s.cMap[i]];
        s.wc[w].swap(nw);
    }
}

// Owner pose as the force kernel sees it: kernel/DEMCalcForceKernels.cu:19-42 (equipOwnerPosRot)
inline void owner_pose(const Sim& s, uint32_t o, V3d& pos, Q4& q) {
    decode_pos(s.voxelID[o], s.locX[o], s.locY[o], s.locZ[o], s.p.nvXp2, s.p.nvYp2, s.p.voxelSize, s.p.l, pos.x, pos.y,
               pos.z);
    pos.x += s.p.LBFX;
    pos.y += s.p.LBFY;
    pos.z += s.p.LBFZ;
    q = {s.oriQw[o], s.oriQx[o], s.oriQy[o], s.oriQz[o]};
}

// kernel/DEMCalcForceKernels.cu:44-267 (calculateContactForces) followed by the
// accumulation of kernel/DEMCustomizablePolicies/ForceInKernelReductionStrat.cu
// (same arithmetic as DEMCollectForceKernels_Compact.cu:13-101), and the
// clearing of kernel/DEMPrepForceKernels.cu:15-37.
void calc_forces(Sim& s, bool record) {
    const size_t nC = s.cA.size();
    std::fill(s.aX.begin(), s.aX.end(), 0.f);
    std::fill(s.aY.begin(), s.aY.end(), 0.f);
    std::fill(s.aZ.begin(), s.aZ.end(), 0.f);
    std::fill(s.alX.begin(), s.alX.end(), 0.f);
    std::fill(s.alY.begin(), s.alY.end(), 0.f);
    std::fill(s.alZ.begin(), s.alZ.end(), 0.f);
    // per-contact results kept so accumulation can run in list order (deterministic)
    std::vector<float> F(nC * 3, 0.f), T(nC * 3, 0.f), PA(nC * 3, 0.f), PB(nC * 3, 0.f);
    std::vector<uint8_t> live(nC, 0);
    std::vector<uint32_t> ownA(nC), ownB(nC);
    const bool hist = (s.p.forceModel == DEME_FORCE_HERTZIAN);
    const bool cohesive = (s.p.forceModel == DEME_FORCE_CUSTOM && s.customKind == 1);
    if (s.p.forceModel == DEME_FORCE_CUSTOM && !cohesive)
        throw std::runtime_error("oracle: DEME_FORCE_CUSTOM needs a parametric model (orc_sim_set_custom_model)");
    if (hist)
        for (int w = 0; w < 4; w++)
            s.wc[w].resize(nC, 0.f);
    if (cohesive)
        s.wc[0].resize(nC, 0.f);

#pragma omp parallel for schedule(static)
    for (int64_t ci = 0; ci < (int64_t)nC; ci++) {
        const size_t c = (size_t)ci;
        uint8_t type = s.cType[c];
        ForceIn in{};
        V3d contactPnt{}, AOwnerPos, BOwnerPos, bodyAPos, bodyBPos;
        in.ts = s.p.h;
        // ---- A is always a sphere
        const uint32_t sA = s.cA[c];
        const uint32_t oA = s.ownerOfSphere[sA];
        ownA[c] = oA;
        {
            const uint16_t cp = s.compOff[sA];
            in.AOwnerMass = s.mass[s.inertiaOff[oA]];
            in.ALinVel = {s.vX[oA], s.vY[oA], s.vZ[oA]};
            in.ARotVel = {s.omgX[oA], s.omgY[oA], s.omgZ[oA]};
            owner_pose(s, oA, AOwnerPos, in.AOriQ);
            const V3f rel = rotate_q(in.AOriQ, {s.relX[cp], s.relY[cp], s.relZ[cp]});
            bodyAPos = {AOwnerPos.x + (double)rel.x, AOwnerPos.y + (double)rel.y, AOwnerPos.z + (double)rel.z};
            in.ARadius = s.Radii[cp];
        }
        const unsigned famA = s.familyID[oA];
        float extraMargin = s.famExtra[famA];
        uint16_t matA = s.sphMat[sA], matB = 0;
        uint32_t oB = 0;
        if (type == DEME_SPHERE_SPHERE_CONTACT) {
            const uint32_t sB = s.cB[c];
            oB = s.ownerOfSphere[sB];
            const uint16_t cp = s.compOff[sB];
            in.BOwnerMass = s.mass[s.inertiaOff[oB]];
            in.BLinVel = {s.vX[oB], s.vY[oB], s.vZ[oB]};
            in.BRotVel = {s.omgX[oB], s.omgY[oB], s.omgZ[oB]};
            owner_pose(s, oB, BOwnerPos, in.BOriQ);
            const V3f rel = rotate_q(in.BOriQ, {s.relX[cp], s.relY[cp], s.relZ[cp]});
            bodyBPos = {BOwnerPos.x + (double)rel.x, BOwnerPos.y + (double)rel.y, BOwnerPos.z + (double)rel.z};
            in.BRadius = s.Radii[cp];
            matB = s.sphMat[sB];
            const float eB = s.famExtra[s.familyID[oB]];
            extraMargin = (extraMargin > eB) ? extraMargin : eB;
            spheres_overlap(bodyAPos.x, bodyAPos.y, bodyAPos.z, (double)in.ARadius, bodyBPos.x, bodyBPos.y, bodyBPos.z,
                            (double)in.BRadius, contactPnt.x, contactPnt.y, contactPnt.z, in.B2A.x, in.B2A.y, in.B2A.z,
                            in.overlapDepth);
            if (in.overlapDepth < -extraMargin)
                type = DEME_NOT_A_CONTACT;
        } else if (type > 10) {  // analytical, DEMCalcForceKernels.cu:184-232
            const uint32_t ob = s.cB[c];
            oB = s.objOwner[ob];
            matB = s.objMat[ob];
            in.BOwnerMass = s.objMass[ob];
            in.BRadius = kHugeF;
            in.BLinVel = {s.vX[oB], s.vY[oB], s.vZ[oB]};
            in.BRotVel = {s.omgX[oB], s.omgY[oB], s.omgZ[oB]};
            owner_pose(s, oB, BOwnerPos, in.BOriQ);
            const RotM m = rot_coeffs(in.BOriQ.w, in.BOriQ.x, in.BOriQ.y, in.BOriQ.z);
            const V3f rel = rotate_f(m, {s.objRelX[ob], s.objRelY[ob], s.objRelZ[ob]});
            bodyBPos = {BOwnerPos.x + (double)rel.x, BOwnerPos.y + (double)rel.y, BOwnerPos.z + (double)rel.z};
            const float eB = s.famExtra[s.familyID[oB]];
            extraMargin = (extraMargin > eB) ? extraMargin : eB;
            const V3f rot = rotate_f(m, {s.objRotX[ob], s.objRotY[ob], s.objRotZ[ob]});
            sphere_entity(bodyAPos, in.ARadius, s.objType[ob], bodyBPos, rot, s.objS1[ob], s.objS2[ob], s.objS3[ob],
                          s.objNormal[ob], 0.0f, contactPnt, in.B2A, in.overlapDepth);
            if (in.overlapDepth < -extraMargin)
                type = DEME_NOT_A_CONTACT;
        } else if (type == DEME_SPHERE_MESH_CONTACT) {  // DEMCalcForceKernels.cu:138-183
            const uint32_t t = s.cB[c];
            oB = s.ownerMesh[t];
            in.BRadius = kHugeF;
            matB = s.triMat[t];
            const float eB = s.famExtra[s.familyID[oB]];
            extraMargin = (extraMargin > eB) ? extraMargin : eB;
            in.BOwnerMass = s.mass[s.inertiaOff[oB]];
            in.BLinVel = {s.vX[oB], s.vY[oB], s.vZ[oB]};
            in.BRotVel = {s.omgX[oB], s.omgY[oB], s.omgZ[oB]};
            owner_pose(s, oB, BOwnerPos, in.BOriQ);
            const RotM m = rot_coeffs(in.BOriQ.w, in.BOriQ.x, in.BOriQ.y, in.BOriQ.z);
            T3d nd[3];
            const float* src[3] = {&s.tri1[3 * t], &s.tri2[3 * t], &s.tri3[3 * t]};
            for (int k = 0; k < 3; k++) {
                const V3d r = rotate_d(m, {(double)src[k][0], (double)src[k][1], (double)src[k][2]});
                nd[k] = {BOwnerPos.x + r.x, BOwnerPos.y + r.y, BOwnerPos.z + r.z};
            }
            bodyBPos = {(nd[0].x + nd[1].x + nd[2].x) / 3., (nd[0].y + nd[1].y + nd[2].y) / 3., (nd[0].z + nd[1].z + nd[2].z) / 3.};
            T3d cn, cpt;
            double depth;
            const bool in_contact = tri_sphere_cd<double, false>(nd[0], nd[1], nd[2], {bodyAPos.x, bodyAPos.y, bodyAPos.z},
                                                                 (double)in.ARadius, cn, depth, cpt);
            in.B2A = {(float)cn.x, (float)cn.y, (float)cn.z};
            contactPnt = {cpt.x, cpt.y, cpt.z};
            if ((depth > extraMargin) || (!in_contact && depth < 0.))
                type = DEME_NOT_A_CONTACT;
            in.overlapDepth = -depth;
        } else {
            type = DEME_NOT_A_CONTACT;
        }
        ownB[c] = oB;
        ForceHist h{};
        if (hist)
            h = {s.wc[0][c], s.wc[1][c], s.wc[2][c], s.wc[3][c]};
        float contact_age = cohesive ? s.wc[0][c] : 0.f;
        if (type != DEME_NOT_A_CONTACT) {
            V3f force{0, 0, 0}, torque{0, 0, 0};
            in.locCPA = rotate_q_inv(in.AOriQ, {(float)(contactPnt.x - AOwnerPos.x), (float)(contactPnt.y - AOwnerPos.y),
                                                (float)(contactPnt.z - AOwnerPos.z)});
            in.locCPB = rotate_q_inv(in.BOriQ, {(float)(contactPnt.x - BOwnerPos.x), (float)(contactPnt.y - BOwnerPos.y),
                                                (float)(contactPnt.z - BOwnerPos.z)});
            in.E_A = s.E[matA];
            in.nu_A = s.nu[matA];
            in.E_B = s.E[matB];
            in.nu_B = s.nu[matB];
            in.CoR = s.CoR[matA * s.nMat + matB];
            in.mu = s.mu[matA * s.nMat + matB];
            in.Crr = s.Crr[matA * s.nMat + matB];
            if (s.p.forceModel == DEME_FORCE_HERTZIAN)
                force_hertz_full(in, h, force, torque);
            else if (cohesive) {
                // the user fragment: the frictionless Hertzian statements, then `force += -Cohesion[A][B] * B2A;` and
                // `contact_age += ts;` inside the same `if (overlapDepth > 0)` block
                force_hertz_frictionless(in, force);
                if (in.overlapDepth > 0) {
                    force = force + (-s.cohesion[matA * s.nMat + matB]) * in.B2A;
                    contact_age += in.ts;
                }
            } else
                force_hertz_frictionless(in, force);
            F[c * 3] = force.x, F[c * 3 + 1] = force.y, F[c * 3 + 2] = force.z;
            T[c * 3] = torque.x, T[c * 3 + 1] = torque.y, T[c * 3 + 2] = torque.z;
            PA[c * 3] = in.locCPA.x, PA[c * 3 + 1] = in.locCPA.y, PA[c * 3 + 2] = in.locCPA.z;
            PB[c * 3] = in.locCPB.x, PB[c * 3 + 1] = in.locCPB.y, PB[c * 3 + 2] = in.locCPB.z;
            live[c] = 1;
        } else {
            h = {0, 0, 0, 0};  // _forceModelContactWildcardDestroy_, DEM/Models.h:363-378
            contact_age = 0.f;
        }
        if (cohesive)
            s.wc[0][c] = contact_age;
        if (hist) {
            s.wc[0][c] = h.delta_tan_x;
            s.wc[1][c] = h.delta_tan_y;
            s.wc[2][c] = h.delta_tan_z;
            s.wc[3][c] = h.delta_time;
        }
    }
    // accumulation (the reference uses float atomics: order-nondeterministic).  Fixed order of this build: for every owner
    // first the contacts in which it is the A side, in list order, then those in which it is the B side, in list order --
    // the order of the HIP path's atomics-free gather, so that a / alpha can be compared bit for bit.
#ifdef ORC_PERF
    // per owner, by its own thread: its A run (the list is sorted by sphere A, so by A's owner) and then the contacts that hold
    // it as B, both ascending -- the order of the loop below, so the sums are the same bit for bit
    {
        const size_t nO = s.nOwners;
        std::vector<uint32_t> aStart(nO + 1, 0), bStart(nO + 1, 0);
        for (size_t c = 0; c < nC; c++)
            aStart[ownA[c] + 1]++, bStart[ownB[c] + 1]++;
        for (size_t o = 0; o < nO; o++)
            aStart[o + 1] += aStart[o], bStart[o + 1] += bStart[o];
        std::vector<uint32_t> bIdx(nC), fill(bStart.begin(), bStart.end() - 1);
        for (size_t c = 0; c < nC; c++)
            bIdx[fill[ownB[c]]++] = (uint32_t)c;
#pragma omp parallel for schedule(dynamic, 512)
        for (int64_t oo = 0; oo < (int64_t)nO; oo++) {
            const uint32_t o = (uint32_t)oo;
            const Q4 q{s.oriQw[o], s.oriQx[o], s.oriQy[o], s.oriQz[o]};
            const V3f moi{s.moiX[s.inertiaOff[o]], s.moiY[s.inertiaOff[o]], s.moiZ[s.inertiaOff[o]]};
            float ax = 0.f, ay = 0.f, az = 0.f, lx = 0.f, ly = 0.f, lz = 0.f;
            for (int side = 0; side < 2; side++) {
                const uint32_t i0 = side ? bStart[o] : aStart[o], i1 = side ? bStart[o + 1] : aStart[o + 1];
                for (uint32_t i = i0; i < i1; i++) {
                    const size_t c = side ? bIdx[i] : i;
                    if (!live[c])
                        continue;
                    const V3f force{F[c * 3], F[c * 3 + 1], F[c * 3 + 2]};
                    const V3f tq{T[c * 3], T[c * 3 + 1], T[c * 3 + 2]};
                    const float m = (side && s.cType[c] > 10) ? s.objMass[s.cB[c]] : s.mass[s.inertiaOff[o]];
                    const V3f cp = side ? V3f{PB[c * 3], PB[c * 3 + 1], PB[c * 3 + 2]} : V3f{PA[c * 3], PA[c * 3 + 1], PA[c * 3 + 2]};
                    V3f myF;
                    if (side == 0) {
                        ax += force.x / m, ay += force.y / m, az += force.z / m;
                        myF = force + tq;
                    } else {
                        ax += -force.x / m, ay += -force.y / m, az += -force.z / m;
                        myF = -1.f * (force + tq);
                    }
                    myF = rotate_q_inv(q, myF);
                    const V3f cr = crossf(cp, myF);
                    lx += cr.x / moi.x, ly += cr.y / moi.y, lz += cr.z / moi.z;
                }
            }
            s.aX[o] = ax, s.aY[o] = ay, s.aZ[o] = az, s.alX[o] = lx, s.alY[o] = ly, s.alZ[o] = lz;
        }
    }
#else
    for (int side = 0; side < 2; side++) {
        for (size_t c = 0; c < nC; c++) {
            if (!live[c])
                continue;
            const V3f force{F[c * 3], F[c * 3 + 1], F[c * 3 + 2]};
            const V3f tq{T[c * 3], T[c * 3 + 1], T[c * 3 + 2]};
            const uint32_t o = side ? ownB[c] : ownA[c];
            const float m = (side && s.cType[c] > 10) ? s.objMass[s.cB[c]] : s.mass[s.inertiaOff[o]];
            const V3f moi{s.moiX[s.inertiaOff[o]], s.moiY[s.inertiaOff[o]], s.moiZ[s.inertiaOff[o]]};
            const V3f cp = side ? V3f{PB[c * 3], PB[c * 3 + 1], PB[c * 3 + 2]} : V3f{PA[c * 3], PA[c * 3 + 1], PA[c * 3 + 2]};
            const Q4 q{s.oriQw[o], s.oriQx[o], s.oriQy[o], s.oriQz[o]};
            V3f myF;
            if (side == 0) {
                s.aX[o] += force.x / m;
                s.aY[o] += force.y / m;
                s.aZ[o] += force.z / m;
                myF = force + tq;
            } else {
                s.aX[o] += -force.x / m;
                s.aY[o] += -force.y / m;
                s.aZ[o] += -force.z / m;
                myF = -1.f * (force + tq);
            }
            myF = rotate_q_inv(q, myF);
            const V3f cr = crossf(cp, myF);
            s.alX[o] += cr.x / moi.x;
            s.alY[o] += cr.y / moi.y;
            s.alZ[o] += cr.z / moi.z;
        }
    }
#endif
    if (record) {
        s.recF.swap(F);
        s.recT.swap(T);
        s.recCPA.swap(PA);
        s.recCPB.swap(PB);
    }
}

// DEMCustomizablePolicies/IntegrationVelPassOn{ForwardEuler,CenteredDiff,ExtendedTaylor}.cu: the velocity the position
// update uses, from the velocity before the step and this step's increment (golden vectors G9)
inline V3f vel_pass_on(uint32_t integrator, const V3f& old_v, const V3f& v_upd) {
    switch (integrator) {
        case DEME_INTEGRATOR_FORWARD_EULER:
            return old_v;
        case DEME_INTEGRATOR_CENTERED_DIFFERENCE:
            return old_v + v_upd;
        default:  // extended Taylor: `v_update * 0.5` narrows 0.5 to float (CUDAMathHelpers.cuh:674)
            return old_v + v_upd * 0.5f;
    }
}

// kernel/DEMIntegrationKernels.cu:100-236 (integrateVelPos) with the velocity pass-on
// of DEMCustomizablePolicies/IntegrationVelPassOn{ForwardEuler,CenteredDiff,ExtendedTaylor}.cu.
// Fixed families (SetFamilyFixed, APIPublic.cpp:980-1011) zero and freeze all six velocity
// components and the pose.
void integrate(Sim& s) {
    const float h = s.p.h;
    if (!s.nextAcc.empty()) {
        for (uint32_t o = 0; o < s.nOwners; o++) {
            s.aX[o] += s.nextAcc[6 * o], s.aY[o] += s.nextAcc[6 * o + 1], s.aZ[o] += s.nextAcc[6 * o + 2];
            s.alX[o] += s.nextAcc[6 * o + 3], s.alY[o] += s.nextAcc[6 * o + 4], s.alZ[o] += s.nextAcc[6 * o + 5];
        }
        s.nextAcc.clear();
    }
#pragma omp parallel for schedule(static)
    for (int64_t oi = 0; oi < (int64_t)s.nOwners; oi++) {
        const uint32_t o = (uint32_t)oi;
        if (!s.ghost.empty() && (s.ghost[o] & 1))
            continue;  // ghost of a clump another rank integrates (slab decomposition)
        const bool fixed = (s.famFlags[s.familyID[o]] & DEME_FAMILY_FIXED) != 0;
        V3f old_v{s.vX[o], s.vY[o], s.vZ[o]};
        V3f old_w{s.omgX[o], s.omgY[o], s.omgZ[o]};
        double X, Y, Z;
        decode_pos(s.voxelID[o], s.locX[o], s.locY[o], s.locZ[o], s.p.nvXp2, s.p.nvYp2, s.p.voxelSize, s.p.l, X, Y, Z);
        X += (double)s.p.LBFX;
        Y += (double)s.p.LBFY;
        Z += (double)s.p.LBFZ;
        // prescribed motion: applyPrescribedVel / applyPrescribedPos / applyAddedAcceleration (DEMIntegrationKernels.cu:8-98)
        uint32_t pf = 0;
        float ext[6] = {0, 0, 0, 0, 0, 0};
        const Sim::Presc& pr = s.presc[s.familyID[o]];
        if (pr.used && (s.famFlags[s.familyID[o]] & DEME_FAMILY_PRESCRIBED)) {
            const float t = (float)s.p.timeElapsed;
            auto val = [&](int k) { return pr.c[k][0] + pr.c[k][1] * t + pr.c[k][2] * sinf(pr.c[k][3] * t); };
            float* tgt[6] = {&s.vX[o], &s.vY[o], &s.vZ[o], &s.omgX[o], &s.omgY[o], &s.omgZ[o]};
            for (int k = 0; k < 6; k++)
                if (pr.has & (1u << k))
                    *tgt[k] = val(k);
            double* ptg[3] = {&X, &Y, &Z};
            for (int k = 0; k < 3; k++)
                if (pr.has & (1u << (6 + k)))
                    *ptg[k] = val(6 + k);
            for (int k = 0; k < 6; k++)
                if (pr.has & (1u << (9 + k)))
                    ext[k] = val(9 + k);
            pf = pr.flags;
        }
        V3f v_upd{0, 0, 0}, w_upd{0, 0, 0};
        if (fixed) {
            s.vX[o] = s.vY[o] = s.vZ[o] = 0.f;
            s.omgX[o] = s.omgY[o] = s.omgZ[o] = 0.f;
            old_v = {0, 0, 0};
            old_w = {0, 0, 0};
        } else {
            if (!(pf & 1u)) {
                v_upd.x = (s.aX[o] + ext[0] + s.p.Gx) * h;
                s.vX[o] += v_upd.x;
            } else {
                old_v.x = s.vX[o];
            }
            if (!(pf & 2u)) {
                v_upd.y = (s.aY[o] + ext[1] + s.p.Gy) * h;
                s.vY[o] += v_upd.y;
            } else {
                old_v.y = s.vY[o];
            }
            if (!(pf & 4u)) {
                v_upd.z = (s.aZ[o] + ext[2] + s.p.Gz) * h;
                s.vZ[o] += v_upd.z;
            } else {
                old_v.z = s.vZ[o];
            }
            if (!(pf & 8u)) {
                w_upd.x = (s.alX[o] + ext[3]) * h;
                s.omgX[o] += w_upd.x;
            } else {
                old_w.x = s.omgX[o];
            }
            if (!(pf & 16u)) {
                w_upd.y = (s.alY[o] + ext[4]) * h;
                s.omgY[o] += w_upd.y;
            } else {
                old_w.y = s.omgY[o];
            }
            if (!(pf & 32u)) {
                w_upd.z = (s.alZ[o] + ext[5]) * h;
                s.omgZ[o] += w_upd.z;
            } else {
                old_w.z = s.omgZ[o];
            }
        }
        const V3f v = vel_pass_on(s.p.integrator, old_v, v_upd), w = vel_pass_on(s.p.integrator, old_w, w_upd);
        if (!fixed) {
            if (!(pf & 64u))
                X += (double)v.x * h;
            if (!(pf & 128u))
                Y += (double)v.y * h;
            if (!(pf & 256u))
                Z += (double)v.z * h;
        }
        X -= (double)s.p.LBFX;
        Y -= (double)s.p.LBFY;
        Z -= (double)s.p.LBFZ;
        encode_pos(X, Y, Z, s.p.nvXp2, s.p.nvYp2, s.p.voxelSize, s.p.l, s.voxelID[o], s.locX[o], s.locY[o], s.locZ[o]);
        if (!fixed && !(pf & 512u)) {
            // ha = 0.5 * h * omgBar : (double)(0.5*h) narrowed to float, then float*float3
            const float hh = (float)(0.5 * h);
            const V3f ha = hh * w;
            float qw, qx, qy, qz;
            hamilton(qw, qx, qy, qz, s.oriQw[o], s.oriQx[o], s.oriQy[o], s.oriQz[o], 1.0f, ha.x, ha.y, ha.z);
            const float len = sqrtf(qx * qx + qy * qy + qz * qz + qw * qw);  // dot(float4): x,y,z,w order
            s.oriQw[o] = qw / len;
            s.oriQx[o] = qx / len;
            s.oriQy[o] = qy / len;
            s.oriQz[o] = qz / len;
        }
    }
}

// kernel/DEMModeratorKernels.cu:10-60 (applyFamilyChanges) with the rules in the parametric test form; the rules are
// applied in order to the family code read at entry (each `if (family_code == A)` tests the original code).
void apply_family_rules(Sim& s) {
    if (s.famRules.empty())
        return;
    for (uint32_t o = 0; o < s.nOwners; o++) {
        const uint8_t code = s.familyID[o];
        double X, Y, Z;
        decode_pos(s.voxelID[o], s.locX[o], s.locY[o], s.locZ[o], s.p.nvXp2, s.p.nvYp2, s.p.voxelSize, s.p.l, X, Y, Z);
        X += s.p.LBFX;
        Y += s.p.LBFY;
        Z += s.p.LBFZ;
        const double q[10] = {X, Y, Z, s.vX[o], s.vY[o], s.vZ[o], s.aX[o], s.aY[o], s.aZ[o], (double)(float)s.p.timeElapsed};
        for (const Sim::FamRule& r : s.famRules)
            if (code == r.from && (r.op == 0 ? q[r.q] > r.thr : q[r.q] < r.thr))
                s.familyID[o] = (uint8_t)r.to;
    }
}

// Stepping policy of this build (lock-step; SURVEY App. C #10 explains why the
// reference's own "sync" mode cannot be reproduced step for step): every K steps
// (K = cdUpdateFreq, K = 0 means every step with zero drift) refresh margins from the
// current velocities, detect on the current positions, migrate history; then forces
// and integration use that list.
int step(Sim& s, uint32_t n) {
    const uint32_t K = s.p.cdUpdateFreq;
    for (uint32_t i = 0; i < n; i++) {
        if (!s.haveList || s.seeded || K == 0 || s.stepsSinceCD >= K) {
            compute_margins(s, K);
            const int rc = detect(s);
            if (rc)
                return rc;
            migrate(s);
            s.stepsSinceCD = 0;
        }
        calc_forces(s, false);
        apply_family_rules(s);  // routineChecks(): between forces and integration (dT.cpp:2437-2443)
        integrate(s);
        s.stepsSinceCD++;
        s.nSteps++;
        s.p.timeElapsed += (double)s.p.h;
    }
    return DEME_OK;
}

}  // namespace

// ===========================================================================
// exported C interface (ctypes)
// ===========================================================================
extern "C" {

int orc_num_threads() {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
void orc_set_num_threads(int n) {
#ifdef _OPENMP
    omp_set_num_threads(n);
#else
    (void)n;
#endif
}

// ---- element-wise functions (pinned against oracle/_ref) --------------------
void orc_el_decode(size_t n, const uint64_t* id, const uint16_t* sx, const uint16_t* sy, const uint16_t* sz,
                   unsigned nvXp2, unsigned nvYp2, double voxelSize, double l, double* X, double* Y, double* Z) {
    for (size_t i = 0; i < n; i++)
        decode_pos(id[i], sx[i], sy[i], sz[i], nvXp2, nvYp2, voxelSize, l, X[i], Y[i], Z[i]);
}
void orc_el_encode(size_t n, const double* X, const double* Y, const double* Z, unsigned nvXp2, unsigned nvYp2,
                   double voxelSize, double l, uint64_t* id, uint16_t* sx, uint16_t* sy, uint16_t* sz) {
    for (size_t i = 0; i < n; i++)
        encode_pos(X[i], Y[i], Z[i], nvXp2, nvYp2, voxelSize, l, id[i], sx[i], sy[i], sz[i]);
}
void orc_el_rotate(size_t n, float* x, float* y, float* z, const float* qw, const float* qx, const float* qy,
                   const float* qz) {
    for (size_t i = 0; i < n; i++) {
        const V3f r = rotate_q({qw[i], qx[i], qy[i], qz[i]}, {x[i], y[i], z[i]});
        x[i] = r.x, y[i] = r.y, z[i] = r.z;
    }
}
void orc_el_rotate_d(size_t n, double* x, double* y, double* z, const float* qw, const float* qx, const float* qy,
                     const float* qz) {
    for (size_t i = 0; i < n; i++) {
        const V3d r = rotate_d(rot_coeffs(qw[i], qx[i], qy[i], qz[i]), {x[i], y[i], z[i]});
        x[i] = r.x, y[i] = r.y, z[i] = r.z;
    }
}
void orc_el_hamilton(size_t n, const float* q1, const float* q2, float* out) {  // wxyz quadruples
    for (size_t i = 0; i < n; i++)
        hamilton(out[4 * i], out[4 * i + 1], out[4 * i + 2], out[4 * i + 3], q1[4 * i], q1[4 * i + 1], q1[4 * i + 2],
                 q1[4 * i + 3], q2[4 * i], q2[4 * i + 1], q2[4 * i + 2], q2[4 * i + 3]);
}
void orc_el_mask_pair(size_t n, const uint32_t* i, const uint32_t* j, uint32_t* out) {
    for (size_t k = 0; k < n; k++)
        out[k] = mask_pair(i[k], j[k]);
}
void orc_el_point_bin(size_t n, const double* X, const double* Y, const double* Z, double binSize, uint32_t nbX,
                      uint32_t nbY, uint32_t* out) {
    for (size_t i = 0; i < n; i++)
        out[i] = point_bin(X[i], Y[i], Z[i], binSize, nbX, nbY);
}
void orc_el_spheres_overlap(size_t n, const double* A, const double* rA, const double* B, const double* rB,
                            uint8_t* type, double* CP, float* nrm, double* depth) {
    for (size_t i = 0; i < n; i++)
        type[i] = spheres_overlap(A[3 * i], A[3 * i + 1], A[3 * i + 2], rA[i], B[3 * i], B[3 * i + 1], B[3 * i + 2],
                                  rB[i], CP[3 * i], CP[3 * i + 1], CP[3 * i + 2], nrm[3 * i], nrm[3 * i + 1],
                                  nrm[3 * i + 2], depth[i]);
}
void orc_el_sphere_entity(size_t n, const double* A, const float* radA, const uint8_t* typeB, const double* B,
                          const float* dirB, const float* size1, const float* normal_sign, const float* beta,
                          uint8_t* type, double* CP, float* nrm, double* depth) {
    for (size_t i = 0; i < n; i++) {
        V3d cp{0, 0, 0};
        V3f nr{0, 0, 0};
        double d = 0;
        type[i] = sphere_entity({A[3 * i], A[3 * i + 1], A[3 * i + 2]}, radA[i], typeB[i],
                                {B[3 * i], B[3 * i + 1], B[3 * i + 2]}, {dirB[3 * i], dirB[3 * i + 1], dirB[3 * i + 2]},
                                size1[i], 0.f, 0.f, normal_sign[i], beta[i], cp, nr, d);
        CP[3 * i] = cp.x, CP[3 * i + 1] = cp.y, CP[3 * i + 2] = cp.z;
        nrm[3 * i] = nr.x, nrm[3 * i + 1] = nr.y, nrm[3 * i + 2] = nr.z;
        depth[i] = d;
    }
}
void orc_el_mat_proxy(size_t n, const float* Y1, const float* nu1, const float* Y2, const float* nu2, float* E,
                      float* G) {
    for (size_t i = 0; i < n; i++)
        mat_proxy(E[i], G[i], Y1[i], nu1[i], Y2[i], nu2[i]);
}
void orc_el_bin_range(size_t n, const double* pos, const double* radius, double binSize, uint32_t nb, uint32_t* lo,
                      uint32_t* hi) {
    for (size_t i = 0; i < n; i++)
        bin_range(pos[i], radius[i], binSize, nb, lo[i], hi[i]);
}

// Force-model batch.  Layout per item (floats unless noted):
//   depth (double array), fin[36]: B2A[3], mA, mB, rA, rB, AOriQ[wxyz], BOriQ[wxyz], locCPA[3], locCPB[3],
//   ALinVel[3], BLinVel[3], ARotVel[3], BRotVel[3], ts, E_A, nu_A, E_B, nu_B, CoR, mu, Crr  (= 39 floats)
//   hist[4] in/out: delta_tan_x, delta_tan_y, delta_tan_z, delta_time
//   out[6]: force[3], torque_only_force[3]
#define ORC_FORCE_NF 39
static ForceIn unpack_force(double depth, const float* f) {
    ForceIn in{};
    in.overlapDepth = depth;
    in.B2A = {f[0], f[1], f[2]};
    in.AOwnerMass = f[3], in.BOwnerMass = f[4], in.ARadius = f[5], in.BRadius = f[6];
    in.AOriQ = {f[7], f[8], f[9], f[10]};
    in.BOriQ = {f[11], f[12], f[13], f[14]};
    in.locCPA = {f[15], f[16], f[17]};
    in.locCPB = {f[18], f[19], f[20]};
    in.ALinVel = {f[21], f[22], f[23]};
    in.BLinVel = {f[24], f[25], f[26]};
    in.ARotVel = {f[27], f[28], f[29]};
    in.BRotVel = {f[30], f[31], f[32]};
    in.ts = f[33];
    in.E_A = f[34], in.nu_A = f[35], in.E_B = f[36], in.nu_B = f[37];
    in.CoR = f[38];
    return in;
}
void orc_el_force(size_t n, int model, const double* depth, const float* fin, const float* mu, const float* Crr,
                  float* hist, float* out) {
    for (size_t i = 0; i < n; i++) {
        ForceIn in = unpack_force(depth[i], fin + i * ORC_FORCE_NF);
        in.mu = mu[i];
        in.Crr = Crr[i];
        V3f force{0, 0, 0}, tq{0, 0, 0};
        if (model == DEME_FORCE_HERTZIAN) {
            ForceHist h{hist[4 * i], hist[4 * i + 1], hist[4 * i + 2], hist[4 * i + 3]};
            force_hertz_full(in, h, force, tq);
            hist[4 * i] = h.delta_tan_x, hist[4 * i + 1] = h.delta_tan_y, hist[4 * i + 2] = h.delta_tan_z,
                     hist[4 * i + 3] = h.delta_time;
        } else {
            force_hertz_frictionless(in, force);
        }
        out[6 * i] = force.x, out[6 * i + 1] = force.y, out[6 * i + 2] = force.z;
        out[6 * i + 3] = tq.x, out[6 * i + 4] = tq.y, out[6 * i + 5] = tq.z;
    }
}

// ---- mesh helpers (parity unpinned; checked against analytic cases and fenv rounding) -------------
void orc_el_vel_pass_on(size_t n, int scheme, const float* old_v, const float* v_update, float* out) {
    for (size_t i = 0; i < n; i++) {
        const V3f v = vel_pass_on((uint32_t)scheme, {old_v[3 * i], old_v[3 * i + 1], old_v[3 * i + 2]},
                                  {v_update[3 * i], v_update[3 * i + 1], v_update[3 * i + 2]});
        out[3 * i] = v.x, out[3 * i + 1] = v.y, out[3 * i + 2] = v.z;
    }
}
void orc_el_rcp_mul_ru(size_t n, const double* x, const double* y, double* rcp, double* mul) {
    for (size_t i = 0; i < n; i++) {
        rcp[i] = rcp_ru(x[i]);
        mul[i] = mul_ru(x[i], y[i]);
    }
}
// the same two operations done by the FPU in round-upward mode
void orc_el_rcp_mul_fenv(size_t n, const double* x, const double* y, double* rcp, double* mul) {
    const int old = fegetround();
    fesetround(FE_UPWARD);
    for (size_t i = 0; i < n; i++) {
        volatile double a = x[i], b = y[i], one = 1.0;
        volatile double r = one / a, m = a * b;
        rcp[i] = r;
        mul[i] = m;
    }
    fesetround(old);
}
// triangle_sphere_CD<double3,double> (directional = 0) or the fp32 directional flavour (directional = 1)
void orc_el_tri_sphere(size_t n, int directional, const double* A, const double* B, const double* C, const double* P,
                       const double* r, uint8_t* hit, double* normal, double* depth, double* pt) {
    for (size_t i = 0; i < n; i++) {
        if (!directional) {
            T3d nr, cp;
            double d;
            hit[i] = tri_sphere_cd<double, false>({A[3 * i], A[3 * i + 1], A[3 * i + 2]}, {B[3 * i], B[3 * i + 1], B[3 * i + 2]},
                                                  {C[3 * i], C[3 * i + 1], C[3 * i + 2]}, {P[3 * i], P[3 * i + 1], P[3 * i + 2]},
                                                  r[i], nr, d, cp);
            normal[3 * i] = nr.x, normal[3 * i + 1] = nr.y, normal[3 * i + 2] = nr.z;
            pt[3 * i] = cp.x, pt[3 * i + 1] = cp.y, pt[3 * i + 2] = cp.z;
            depth[i] = d;
        } else {
            T3f nr, cp;
            float d;
            hit[i] = tri_sphere_cd<float, true>({(float)A[3 * i], (float)A[3 * i + 1], (float)A[3 * i + 2]},
                                                {(float)B[3 * i], (float)B[3 * i + 1], (float)B[3 * i + 2]},
                                                {(float)C[3 * i], (float)C[3 * i + 1], (float)C[3 * i + 2]},
                                                {(float)P[3 * i], (float)P[3 * i + 1], (float)P[3 * i + 2]}, (float)r[i], nr, d, cp);
            normal[3 * i] = nr.x, normal[3 * i + 1] = nr.y, normal[3 * i + 2] = nr.z;
            pt[3 * i] = cp.x, pt[3 * i + 1] = cp.y, pt[3 * i + 2] = cp.z;
            depth[i] = d;
        }
    }
}
void orc_el_tri_box(size_t n, const float* center, const float* half, const float* A, const float* B, const float* C,
                    uint8_t* out) {
    for (size_t i = 0; i < n; i++) {
        const float bh[3] = {half[i], half[i], half[i]};
        out[i] = tri_box_overlap(center + 3 * i, bh, {A[3 * i], A[3 * i + 1], A[3 * i + 2]}, {B[3 * i], B[3 * i + 1], B[3 * i + 2]},
                                 {C[3 * i], C[3 * i + 1], C[3 * i + 2]});
    }
}
void orc_el_tri_bbox(size_t n, const float* A, const float* B, const float* C, double binSize, uint32_t nbX, uint32_t nbY,
                     uint32_t nbZ, int32_t* L, int32_t* U) {
    const int nbm[3] = {(int)nbX - 1, (int)nbY - 1, (int)nbZ - 1};
    for (size_t i = 0; i < n; i++) {
        const T3f v[3] = {{A[3 * i], A[3 * i + 1], A[3 * i + 2]}, {B[3 * i], B[3 * i + 1], B[3 * i + 2]}, {C[3 * i], C[3 * i + 1], C[3 * i + 2]}};
        tri_bbox_bins(v, binSize, nbm, L + 3 * i, U + 3 * i);
    }
}
size_t orc_sim_get_tri_incidence(void* h, uint32_t* bins, uint32_t* tris, size_t cap) {
    Sim* s = (Sim*)h;
    const size_t n = std::min(cap, s->triIncBin.size());
    std::copy(s->triIncBin.begin(), s->triIncBin.begin() + n, bins);
    std::copy(s->triIncTri.begin(), s->triIncTri.begin() + n, tris);
    return s->triIncBin.size();
}
void orc_sim_set_tri_nodes(void* h, const float* n1, const float* n2, const float* n3) {
    Sim* s = (Sim*)h;
    s->tri1.assign(n1, n1 + 3 * (size_t)s->nTri);
    s->tri2.assign(n2, n2 + 3 * (size_t)s->nTri);
    s->tri3.assign(n3, n3 + 3 * (size_t)s->nTri);
}

// ---- simulation object ------------------------------------------------------
void* orc_sim_create(const DemeParams* p, const DemeScene* sc) {
    Sim* s = new Sim();
    s->p = *p;
    s->nOwners = sc->nOwners, s->nOwnerClumps = sc->nOwnerClumps, s->nSpheres = sc->nSpheres, s->nAnal = sc->nAnal;
    s->nMat = sc->nMat, s->nComp = sc->nComp, s->nMassProps = sc->nMassProps;
    const size_t nO = sc->nOwners, nS = sc->nSpheres;
    assign(s->voxelID, sc->voxelID, nO);
    assign(s->locX, sc->locX, nO), assign(s->locY, sc->locY, nO), assign(s->locZ, sc->locZ, nO);
    assign(s->oriQw, sc->oriQw, nO), assign(s->oriQx, sc->oriQx, nO), assign(s->oriQy, sc->oriQy, nO),
        assign(s->oriQz, sc->oriQz, nO);
    assign(s->vX, sc->vX, nO), assign(s->vY, sc->vY, nO), assign(s->vZ, sc->vZ, nO);
    assign(s->omgX, sc->omgBarX, nO), assign(s->omgY, sc->omgBarY, nO), assign(s->omgZ, sc->omgBarZ, nO);
    s->aX.assign(nO, 0), s->aY.assign(nO, 0), s->aZ.assign(nO, 0);
    s->alX.assign(nO, 0), s->alY.assign(nO, 0), s->alZ.assign(nO, 0);
    assign(s->familyID, sc->familyID, nO);
    assign(s->inertiaOff, sc->inertiaPropOffsets, nO);
    s->margin.assign(nO, 0.f);
    assign(s->ownerOfSphere, sc->ownerClumpBody, nS);
    assign(s->compOff, sc->clumpComponentOffset, nS);
    assign(s->sphMat, sc->sphereMaterialOffset, nS);
    assign(s->Radii, sc->Radii, sc->nComp), assign(s->relX, sc->CDRelPosX, sc->nComp),
        assign(s->relY, sc->CDRelPosY, sc->nComp), assign(s->relZ, sc->CDRelPosZ, sc->nComp);
    assign(s->mass, sc->MassProperties, sc->nMassProps), assign(s->moiX, sc->moiX, sc->nMassProps),
        assign(s->moiY, sc->moiY, sc->nMassProps), assign(s->moiZ, sc->moiZ, sc->nMassProps);
    const size_t nA = sc->nAnal;
    assign(s->objType, sc->objType, nA), assign(s->objOwner, sc->objOwner, nA), assign(s->objNormal, sc->objNormal, nA);
    assign(s->objMat, sc->objMaterial, nA);
    assign(s->objRelX, sc->objRelPosX, nA), assign(s->objRelY, sc->objRelPosY, nA), assign(s->objRelZ, sc->objRelPosZ, nA);
    assign(s->objRotX, sc->objRotX, nA), assign(s->objRotY, sc->objRotY, nA), assign(s->objRotZ, sc->objRotZ, nA);
    assign(s->objS1, sc->objSize1, nA), assign(s->objS2, sc->objSize2, nA), assign(s->objS3, sc->objSize3, nA);
    assign(s->objMass, sc->objMass, nA);
    const size_t nM = sc->nMat;
    assign(s->E, sc->E, nM), assign(s->nu, sc->nu, nM);
    assign(s->CoR, sc->CoR, nM * nM), assign(s->mu, sc->mu, nM * nM), assign(s->Crr, sc->Crr, nM * nM);
    assign(s->masks, sc->familyMasks, (size_t)DEME_FAMILY_MASK_ENTRIES);
    assign(s->famExtra, sc->familyExtraMarginSize, (size_t)DEME_NUM_FAMILIES);
    assign(s->famFlags, sc->familyFlags, (size_t)DEME_NUM_FAMILIES);
    s->nTri = sc->nTri;
    assign(s->ownerMesh, sc->ownerMesh, (size_t)sc->nTri);
    assign(s->tri1, sc->triNode1, (size_t)sc->nTri * 3), assign(s->tri2, sc->triNode2, (size_t)sc->nTri * 3),
        assign(s->tri3, sc->triNode3, (size_t)sc->nTri * 3);
    assign(s->triMat, sc->triMaterialOffset, (size_t)sc->nTri);
    if (sc->ownerGhost)
        s->ghost.assign(sc->ownerGhost, sc->ownerGhost + sc->nOwners);
    return s;
}
void orc_sim_destroy(void* h) { delete (Sim*)h; }
void orc_sim_set_params(void* h, const DemeParams* p) { ((Sim*)h)->p = *p; }
void orc_sim_set_custom_model(void* h, int kind, const float* cohesion, size_t n) {
    Sim& s = *(Sim*)h;
    s.customKind = kind;
    s.cohesion.assign(cohesion, cohesion + n);
}
void orc_sim_set_margins(void* h, const float* m) {
    Sim* s = (Sim*)h;
    s->margin.assign(m, m + s->nOwners);
}
void orc_sim_compute_margins(void* h, uint32_t drift) { compute_margins(*(Sim*)h, drift); }
void orc_sim_get_margins(void* h, float* m) {
    Sim* s = (Sim*)h;
    std::copy(s->margin.begin(), s->margin.end(), m);
}
int orc_sim_detect(void* h) { return detect(*(Sim*)h); }
void orc_sim_migrate(void* h) { migrate(*(Sim*)h); }
void orc_sim_calc_forces(void* h, int record) { calc_forces(*(Sim*)h, record != 0); }
void orc_sim_integrate(void* h) { integrate(*(Sim*)h); }
int orc_sim_step(void* h, uint32_t n) { return step(*(Sim*)h, n); }

void orc_sim_counts(void* h, DemeCounts* c) {
    Sim* s = (Sim*)h;
    memset(c, 0, sizeof(*c));
    c->nContacts = s->cA.size();
    c->nPrevContacts = s->pA.size();
    c->nBinSphereTouches = s->incBin.size();
    c->nActiveBins = s->nActiveBins;
    c->nSteps = s->nSteps;
    c->nDetections = s->nDetections;
    c->maxSpheresInBin = s->maxInBin;
}
// Inspectors: DEMSphereQueryKernels.cu:13-54 / DEMOwnerQueryKernels.cu:11-63 with the quantity fragments of
// AuxClasses.cpp:19-92.  values (may be null) receives the per-element quantity; the return value is the number
// of elements; *reduced gets max / min / sum (sums accumulated in double, the reference reduces fp32 with CUB).
void orc_sim_add_family_rule(void* h, uint32_t from, uint32_t to, uint32_t q, uint32_t op, double thr) {
    ((Sim*)h)->famRules.push_back({from, to, q, op, thr});
}
// DEM/APIPrivate.cpp:33-117: mode 0 every contact of the current list, 1 either owner family == N1, 2 both == N1,
// 3 the pair (N1, N2) in either order; mark != 0 marks, 0 removes the qualification
void orc_sim_mark_persistent(void* h, int mode, uint32_t N1, uint32_t N2, int mark) {
    Sim& s = *(Sim*)h;
    std::vector<Key> hit;
    for (size_t i = 0; i < s.cA.size(); i++) {
        const uint32_t oA = s.ownerOfSphere[s.cA[i]];
        const uint32_t oB = s.cType[i] == DEME_SPHERE_SPHERE_CONTACT ? s.ownerOfSphere[s.cB[i]]
                            : (s.cType[i] == DEME_SPHERE_MESH_CONTACT ? s.ownerMesh[s.cB[i]] : s.objOwner[s.cB[i]]);
        const uint32_t fA = s.familyID[oA], fB = s.familyID[oB];
        const bool q = mode == 0 || (mode == 1 && (fA == N1 || fB == N1)) || (mode == 2 && fA == N1 && fB == N1) ||
                       (mode == 3 && ((fA == N1 && fB == N2) || (fA == N2 && fB == N1)));
        if (q)
            hit.push_back({s.cA[i], s.cB[i], s.cType[i]});  // the list is in key_less order already
    }
    std::vector<Key> out;
    if (mark)
        std::set_union(s.persist.begin(), s.persist.end(), hit.begin(), hit.end(), std::back_inserter(out), key_less);
    else
        std::set_difference(s.persist.begin(), s.persist.end(), hit.begin(), hit.end(), std::back_inserter(out), key_less);
    s.persist.swap(out);
}
void orc_sim_add_owner_acc(void* h, uint32_t owner, uint32_t n, const float* acc, const float* angAcc) {
    Sim& s = *(Sim*)h;
    if (s.nextAcc.empty())
        s.nextAcc.assign(6 * (size_t)s.nOwners, 0.f);
    for (uint32_t k = 0; k < n; k++)
        for (int d = 0; d < 3; d++) {
            if (acc)
                s.nextAcc[6 * (size_t)(owner + k) + d] = acc[3 * k + d];
            if (angAcc)
                s.nextAcc[6 * (size_t)(owner + k) + 3 + d] = angAcc[3 * k + d];
        }
}
size_t orc_sim_num_persistent(void* h) { return ((Sim*)h)->persist.size(); }
void orc_sim_get_persistent(void* h, uint32_t* a, uint32_t* b, uint8_t* t) {
    Sim& s = *(Sim*)h;
    for (size_t i = 0; i < s.persist.size(); i++)
        a[i] = s.persist[i].a, b[i] = s.persist[i].b, t[i] = s.persist[i].t;
}
void orc_sim_set_persistent(void* h, const uint32_t* a, const uint32_t* b, const uint8_t* t, size_t n) {
    Sim& s = *(Sim*)h;
    s.persist.clear();
    for (size_t i = 0; i < n; i++)
        s.persist.push_back({a[i], b[i], t[i]});
    std::sort(s.persist.begin(), s.persist.end(), key_less);
    s.persist.erase(std::unique(s.persist.begin(), s.persist.end(),
                                [](const Key& x, const Key& y) { return x.a == y.a && x.b == y.b && type_class(x.t) == type_class(y.t); }),
                    s.persist.end());
}
void orc_sim_change_family(void* h, uint32_t from, uint32_t to) {
    Sim& s = *(Sim*)h;
    for (auto& f : s.familyID)
        if (f == from)
            f = (uint8_t)to;
}
// family prescription in the parametric test form (see Sim::Presc); coef = float[15][4]
void orc_sim_set_prescription(void* h, uint32_t family, uint32_t has, uint32_t flags, const float* coef) {
    Sim& s = *(Sim*)h;
    Sim::Presc& p = s.presc[family & 255u];
    p.used = true;
    p.has = has;
    p.flags = flags;
    memcpy(p.c, coef, sizeof(p.c));
}
// restart / re-decomposition: mirror of deme_seed_contacts (include/deme_hip.h)
int orc_sim_seed_contacts(void* h, const uint32_t* idA, const uint32_t* idB, const uint8_t* type, const float* wildcards, size_t n) {
    Sim& s = *(Sim*)h;
    const uint32_t nW = s.p.nContactWildcards;
    std::vector<Key> keys(n);
    std::vector<uint32_t> perm(n);
    for (size_t i = 0; i < n; i++) {
        keys[i] = {idA[i], idB[i], type[i]};
        perm[i] = (uint32_t)i;
    }
    std::sort(perm.begin(), perm.end(), [&](uint32_t x, uint32_t y) { return key_less(keys[x], keys[y]); });
    for (size_t i = 1; i < n; i++) {  // same rule as deme_seed_contacts: a pair may appear once
        const Key &x = keys[perm[i - 1]], &y = keys[perm[i]];
        if (x.a == y.a && x.b == y.b && x.t == y.t)
            return DEME_ERR_INVALID;
    }
    s.cA.resize(n), s.cB.resize(n), s.cType.resize(n);
    s.cMap.assign(n, DEME_NULL_MAPPING_PARTNER);
    for (uint32_t w = 0; w < nW; w++)
        s.wc[w].assign(n, 0.f);
    for (size_t i = 0; i < n; i++) {
        const Key& k = keys[perm[i]];
        s.cA[i] = k.a, s.cB[i] = k.b, s.cType[i] = k.t;
        for (uint32_t w = 0; w < nW; w++)
            s.wc[w][i] = wildcards[(size_t)perm[i] * nW + w];
    }
    s.haveList = true;
    s.seeded = true;
    return DEME_OK;
}
// region (test form of the reference's region string, AuxClasses.cpp:205-223): axis-aligned box lo <= (X, Y, Z) <= hi on
// the float coordinates the query kernels form (DEMSphereQueryKernels.cu:41-47, DEMOwnerQueryKernels.cu:47-53); null = all
size_t orc_sim_inspect_region(void* h, uint32_t q, const float* lo, const float* hi, float* reduced, float* values) {
    Sim& s = *(Sim*)h;
    const bool perSphere = q <= DEME_INSPECT_CLUMP_MAX_ABSV;
    const size_t n = perSphere ? s.nSpheres : s.nOwners;
    const bool isMax = q == DEME_INSPECT_CLUMP_MAX_Z || q == DEME_INSPECT_CLUMP_MAX_ABSV || q == DEME_INSPECT_MAX_ABSV;
    const bool isMin = q == DEME_INSPECT_CLUMP_MIN_Z;
    const float identity = isMax ? -3.402823466e38f : isMin ? 3.402823466e38f : 0.f;
    double sum = 0;
    float best = identity;
    for (size_t i = 0; i < n; i++) {
        const uint32_t o = perSphere ? s.ownerOfSphere[i] : (uint32_t)i;
        float v;
        const bool clumpOnly = q == DEME_INSPECT_CLUMP_MASS || q == DEME_INSPECT_CLUMP_KINETIC_ENERGY || q == DEME_INSPECT_CLUMP_VOLUME;
        if ((!s.ghost.empty() && (s.ghost[o] & 1)) || (clumpOnly && o >= s.nOwnerClumps)) {
            v = identity;
        } else if (perSphere) {
            const uint16_t c = s.compOff[i];
            const RotM m = rot_coeffs(s.oriQw[o], s.oriQx[o], s.oriQy[o], s.oriQz[o]);
            const V3f rel = rotate_f(m, {s.relX[c], s.relY[c], s.relZ[c]});
            double oX, oY, oZ;
            decode_pos(s.voxelID[o], s.locX[o], s.locY[o], s.locZ[o], s.p.nvXp2, s.p.nvYp2, s.p.voxelSize, s.p.l, oX, oY, oZ);
            if (q == DEME_INSPECT_CLUMP_MAX_ABSV) {
                const V3f w{s.omgX[o], s.omgY[o], s.omgZ[o]};
                const V3f cr{w.y * rel.z - w.z * rel.y, w.z * rel.x - w.x * rel.z, w.x * rel.y - w.y * rel.x};
                const V3f pr = rotate_f(m, cr);
                const V3f t{pr.x + s.vX[o], pr.y + s.vY[o], pr.z + s.vZ[o]};
                v = lenf(t);
            } else {
                const float Z = (float)(oZ + (double)rel.z + (double)s.p.LBFZ);
                v = (q == DEME_INSPECT_CLUMP_MAX_Z) ? Z + s.Radii[c] : Z - s.Radii[c];
            }
        } else {
            const uint16_t io = s.inertiaOff[o];
            if (q == DEME_INSPECT_CLUMP_MASS) {
                v = s.mass[io];
            } else if (q == DEME_INSPECT_CLUMP_VOLUME) {
                v = io < s.volume.size() ? s.volume[io] : 0.f;
            } else if (q == DEME_INSPECT_CLUMP_KINETIC_ENERGY) {
                double vx = s.vX[o], vy = s.vY[o], vz = s.vZ[o];
                double ke = 0.5 * s.mass[io] * (vx * vx + vy * vy + vz * vz);
                vx = s.omgX[o], vy = s.omgY[o], vz = s.omgZ[o];
                ke += 0.5 * ((double)s.moiX[io] * vx * vx + (double)s.moiY[io] * vy * vy + (double)s.moiZ[io] * vz * vz);
                v = (float)ke;
            } else {
                const double vx = s.vX[o], vy = s.vY[o], vz = s.vZ[o];
                v = (float)sqrt(vx * vx + vy * vy + vz * vz);
            }
        }
        if (lo && hi) {
            double oX, oY, oZ, rx = 0., ry = 0., rz = 0.;
            decode_pos(s.voxelID[o], s.locX[o], s.locY[o], s.locZ[o], s.p.nvXp2, s.p.nvYp2, s.p.voxelSize, s.p.l, oX, oY, oZ);
            if (perSphere) {
                const uint16_t c = s.compOff[i];
                const V3f rel = rotate_f(rot_coeffs(s.oriQw[o], s.oriQx[o], s.oriQy[o], s.oriQz[o]), {s.relX[c], s.relY[c], s.relZ[c]});
                rx = (double)rel.x, ry = (double)rel.y, rz = (double)rel.z;
            }
            const float X = perSphere ? (float)(oX + rx + (double)s.p.LBFX) : (float)(oX + (double)s.p.LBFX);
            const float Y = perSphere ? (float)(oY + ry + (double)s.p.LBFY) : (float)(oY + (double)s.p.LBFY);
            const float Z = perSphere ? (float)(oZ + rz + (double)s.p.LBFZ) : (float)(oZ + (double)s.p.LBFZ);
            if (!(X >= lo[0] && X <= hi[0] && Y >= lo[1] && Y <= hi[1] && Z >= lo[2] && Z <= hi[2]))
                v = identity;
        }
        if (values)
            values[i] = v;
        sum += v;
        if (isMax ? v > best : v < best)
            best = v;
    }
    if (reduced)
        *reduced = (isMax || isMin) ? best : (float)sum;
    return n;
}
size_t orc_sim_inspect(void* h, uint32_t q, float* reduced, float* values) {
    return orc_sim_inspect_region(h, q, nullptr, nullptr, reduced, values);
}
void orc_sim_set_volumes(void* h, const float* v, size_t n) { ((Sim*)h)->volume.assign(v, v + n); }
void orc_sim_get_state(void* h, DemeOwnerState* st) {
    Sim* s = (Sim*)h;
    auto cp = [](auto& v, auto* dst) {
        if (dst)
            std::copy(v.begin(), v.end(), dst);
    };
    cp(s->voxelID, st->voxelID), cp(s->locX, st->locX), cp(s->locY, st->locY), cp(s->locZ, st->locZ);
    cp(s->oriQw, st->oriQw), cp(s->oriQx, st->oriQx), cp(s->oriQy, st->oriQy), cp(s->oriQz, st->oriQz);
    cp(s->vX, st->vX), cp(s->vY, st->vY), cp(s->vZ, st->vZ);
    cp(s->omgX, st->omgBarX), cp(s->omgY, st->omgBarY), cp(s->omgZ, st->omgBarZ);
    cp(s->aX, st->aX), cp(s->aY, st->aY), cp(s->aZ, st->aZ);
    cp(s->alX, st->alphaX), cp(s->alY, st->alphaY), cp(s->alZ, st->alphaZ);
    cp(s->familyID, st->familyID);
}
void orc_sim_set_state(void* h, const DemeOwnerState* st) {
    Sim* s = (Sim*)h;
    const size_t n = s->nOwners;
    auto cp = [n](auto& v, const auto* src) {
        if (src)
            std::copy(src, src + n, v.begin());
    };
    cp(s->voxelID, st->voxelID), cp(s->locX, st->locX), cp(s->locY, st->locY), cp(s->locZ, st->locZ);
    cp(s->oriQw, st->oriQw), cp(s->oriQx, st->oriQx), cp(s->oriQy, st->oriQy), cp(s->oriQz, st->oriQz);
    cp(s->vX, st->vX), cp(s->vY, st->vY), cp(s->vZ, st->vZ);
    cp(s->omgX, st->omgBarX), cp(s->omgY, st->omgBarY), cp(s->omgZ, st->omgBarZ);
    cp(s->aX, st->aX), cp(s->aY, st->aY), cp(s->aZ, st->aZ);
    cp(s->alX, st->alphaX), cp(s->alY, st->alphaY), cp(s->alZ, st->alphaZ);
    cp(s->familyID, st->familyID);
}
size_t orc_sim_get_incidence(void* h, uint32_t* bins, uint32_t* sph, size_t cap) {
    Sim* s = (Sim*)h;
    const size_t n = std::min(cap, s->incBin.size());
    std::copy(s->incBin.begin(), s->incBin.begin() + n, bins);
    std::copy(s->incSph.begin(), s->incSph.begin() + n, sph);
    return s->incBin.size();
}
size_t orc_sim_get_contacts(void* h, uint32_t* A, uint32_t* B, uint8_t* t, uint32_t* map, size_t cap) {
    Sim* s = (Sim*)h;
    const size_t n = std::min(cap, s->cA.size());
    if (A)
        std::copy(s->cA.begin(), s->cA.begin() + n, A);
    if (B)
        std::copy(s->cB.begin(), s->cB.begin() + n, B);
    if (t)
        std::copy(s->cType.begin(), s->cType.begin() + n, t);
    if (map)
        std::copy(s->cMap.begin(), s->cMap.begin() + n, map);
    return s->cA.size();
}
size_t orc_sim_get_wildcard(void* h, uint32_t w, float* out, size_t cap) {
    Sim* s = (Sim*)h;
    const size_t n = std::min(cap, s->wc[w].size());
    std::copy(s->wc[w].begin(), s->wc[w].begin() + n, out);
    return s->wc[w].size();
}
void orc_sim_set_wildcard(void* h, uint32_t w, const float* in, size_t n) { ((Sim*)h)->wc[w].assign(in, in + n); }
size_t orc_sim_get_records(void* h, float* F, float* T, float* PA, float* PB, size_t cap) {
    Sim* s = (Sim*)h;
    const size_t n = std::min(cap * 3, s->recF.size());
    if (F)
        std::copy(s->recF.begin(), s->recF.begin() + n, F);
    if (T)
        std::copy(s->recT.begin(), s->recT.begin() + n, T);
    if (PA)
        std::copy(s->recCPA.begin(), s->recCPA.begin() + n, PA);
    if (PB)
        std::copy(s->recCPB.begin(), s->recCPB.begin() + n, PB);
    return s->recF.size() / 3;
}
size_t orc_sim_get_sphere_geometry(void* h, double* X, double* Y, double* Z, float* R, size_t cap) {
    Sim* s = (Sim*)h;
    const size_t n = std::min(cap, s->sphX.size());
    std::copy(s->sphX.begin(), s->sphX.begin() + n, X);
    std::copy(s->sphY.begin(), s->sphY.begin() + n, Y);
    std::copy(s->sphZ.begin(), s->sphZ.begin() + n, Z);
    std::copy(s->sphR.begin(), s->sphR.begin() + n, R);
    return s->sphX.size();
}

}  // extern "C"
