// deme_jit.h -- run-time compilation of user force-model fragments (hipRTC, gfx950).
//
// Replaces the reference's jitify/NVRTC path for the force kernel (core/utils/JitHelper.cpp:50-111,
// DEM/APIPrivate.cpp:1381-1574 equipForceModel, DEM/Models.h:219-378).  The user's statement block is
// spliced LITERALLY (no regex: SURVEY App. B notes the reference's regex_replace hazard) into a generated
// deme_user_model() that pre-declares the reference's ingredient names with the reference's types; the
// kernel around it is the same source as the built-in models (deme_force.h), handed to hipRTC as in-memory
// headers.  Compiled code objects are cached by a hash of the generated source.
#pragma once
#include <hip/hiprtc.h>

#include <functional>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include "deme_embed.inc"  // kSrcDeviceH, kSrcForceH: the texts of deme_device.h / deme_force.h

namespace deme_jit {

// Names and helpers user fragments are written against (the reference gets them from DEM/Defines.h,
// kernel/CUDAMathHelpers.cuh and kernel/DEMHelperKernels.cuh).  Own implementation, reference vocabulary.
static const char* kVocabulary = R"DEMEVOC(
#define DEME_TINY_FLOAT 1e-12
#define DEME_HUGE_FLOAT 1e15
#define DEME_MIN(a, b) ((a < b) ? a : b)
#define DEME_MAX(a, b) ((a > b) ? a : b)
namespace deme {
typedef float oriQ_t;
typedef unsigned int bodyID_t;
typedef unsigned short materialsOffset_t;
typedef unsigned char family_t;
typedef unsigned char contact_t;
constexpr double TWO_OVER_THREE = 2. / 3.;
constexpr double FOUR_OVER_THREE = 4. / 3.;
constexpr double FIVE_OVER_THREE = 5. / 3.;
constexpr double TWO_TIMES_SQRT_FIVE_OVER_SIX = 1.825741858350554;
constexpr double PI = 3.1415926535897932385;
constexpr double PI_SQUARED = 9.869604401089358;
const contact_t NOT_A_CONTACT = 0;
const contact_t SPHERE_SPHERE_CONTACT = 1;
const contact_t SPHERE_MESH_CONTACT = 2;
const contact_t SPHERE_ANALYTICAL_CONTACT = 10;
const contact_t SPHERE_PLANE_CONTACT = 11;
const contact_t SPHERE_PLATE_CONTACT = 12;
const contact_t SPHERE_CYL_CONTACT = 13;
}
__device__ inline float3 make_float3(float s) { return make_float3(s, s, s); }
__device__ inline float dot(float3 a, float3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ inline double dot(double3 a, double3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ inline float3 cross(float3 a, float3 b) {
    return make_float3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ inline float length(float3 v) { return sqrtf(dot(v, v)); }
__device__ inline double length(double3 v) { return sqrt(dot(v, v)); }
__device__ inline float3 normalize(float3 v) { const float il = rsqrtf(dot(v, v)); return v * il; }
__device__ inline float3 to_float3(double3 a) { return make_float3((float)a.x, (float)a.y, (float)a.z); }
__device__ inline double3 to_double3(float3 a) { return make_double3(a.x, a.y, a.z); }
template <typename T1, typename T2>
__device__ inline void applyOriQToVector3(T1& X, T1& Y, T1& Z, const T2& Qw, const T2& Qx, const T2& Qy, const T2& Qz) {
    const T1 oX = X, oY = Y, oZ = Z;
    X = ((T2)2.0 * (Qw * Qw + Qx * Qx) - (T2)1.0) * oX + ((T2)2.0 * (Qx * Qy - Qw * Qz)) * oY + ((T2)2.0 * (Qx * Qz + Qw * Qy)) * oZ;
    Y = ((T2)2.0 * (Qx * Qy + Qw * Qz)) * oX + ((T2)2.0 * (Qw * Qw + Qy * Qy) - (T2)1.0) * oY + ((T2)2.0 * (Qy * Qz - Qw * Qx)) * oZ;
    Z = ((T2)2.0 * (Qx * Qz - Qw * Qy)) * oX + ((T2)2.0 * (Qy * Qz + Qw * Qx)) * oY + ((T2)2.0 * (Qw * Qw + Qz * Qz) - (T2)1.0) * oZ;
}
template <typename T1>
__device__ inline void matProxy2ContactParam(T1& E_eff, T1& G_eff, const T1& Y1, const T1& nu1, const T1& Y2, const T1& nu2) {
    const T1 invE = ((T1)1 - nu1 * nu1) / Y1 + ((T1)1 - nu2 * nu2) / Y2;
    E_eff = (T1)1 / invE;
    const T1 invG = (T1)2 * ((T1)2 - nu1) * ((T1)1 + nu1) / Y1 + (T1)2 * ((T1)2 - nu2) * ((T1)1 + nu2) / Y2;
    G_eff = (T1)1 / invG;
}
template <typename T1>
__device__ inline void matProxy2ContactParam(T1& E_eff, const T1& Y1, const T1& nu1, const T1& Y2, const T1& nu2) {
    const T1 invE = ((T1)1 - nu1 * nu1) / Y1 + ((T1)1 - nu2 * nu2) / Y2;
    E_eff = (T1)1 / invE;
}
)DEMEVOC";

struct MaterialTables {
    uint32_t nMat = 0;
    std::vector<float> E, nu, CoR, mu, Crr;  // CoR/mu/Crr are nMat*nMat
};

inline std::string float_lit(float v) {
    char b[64];
    snprintf(b, sizeof(b), "%.9g", (double)v);
    std::string s(b);
    if (s.find_first_of(".eEn") == std::string::npos)  // "inf"/"nan" contain 'n'
        s += ".0";
    return s + "f";
}

inline void emit_array(std::ostringstream& o, const char* name, const std::vector<float>& v, uint32_t n, bool pair) {
    if (v.empty())
        return;
    if (!pair) {
        o << "__device__ const float " << name << "[] = {";
        for (uint32_t i = 0; i < n; i++)
            o << (i ? ", " : "") << float_lit(v[i]);
        o << "};\n";
    } else {  // _materialDefs_ pairwise form: float name[][nMat] (APIPrivate.cpp:1877-2026)
        o << "__device__ const float " << name << "[][" << n << "] = {";
        for (uint32_t a = 0; a < n; a++) {
            o << (a ? ", {" : "{");
            for (uint32_t b = 0; b < n; b++)
                o << (b ? ", " : "") << float_lit(v[(size_t)a * n + b]);
            o << "}";
        }
        o << "};\n";
    }
}

// ingredient names a fragment may not redefine as wildcards (reference check: APIPrivate.cpp:1425-1465)
inline bool reserved_name(const std::string& n) {
    static const char* r[] = {"overlapDepth", "B2A", "contactPnt", "AOwnerPos", "BOwnerPos", "bodyAPos", "bodyBPos",
                              "AOwnerMass", "BOwnerMass", "ARadius", "BRadius", "AOriQ", "BOriQ", "bodyAMatType",
                              "bodyBMatType", "ContactType", "locCPA", "locCPB", "force", "torque_only_force",
                              "AOwnerFamily", "BOwnerFamily", "ts", "time", "ALinVel", "BLinVel", "ARotVel", "BRotVel",
                              "AOwner", "BOwner", "AGeo", "BGeo", "AOwnerMOI", "BOwnerMOI", "myContactID", "granData",
                              "simParams", "E", "nu", "CoR", "mu", "Crr"};
    for (const char* k : r)
        if (n == k)
            return true;
    return false;
}

inline bool valid_identifier(const std::string& n) {
    if (n.empty() || !(isalpha((unsigned char)n[0]) || n[0] == '_'))
        return false;
    for (char ch : n)
        if (!(isalnum((unsigned char)ch) || ch == '_'))
            return false;
    return true;
}

inline int generate_source(const std::string& user, const std::vector<std::string>& wildcards, const std::string& prereq,
                           const MaterialTables& mt, std::string& out, std::string& err,
                           const std::vector<std::string>& ownerWildcards = {}, const std::vector<std::string>& geoWildcards = {}) {
    for (const auto* set : {&ownerWildcards, &geoWildcards})
        for (const auto& n : *set)
            if (!valid_identifier(n) || reserved_name(n)) {
                err = "wildcard name '" + n + "' is not a usable identifier (it clashes with a force-model ingredient)";
                return 1;
            }
    if (ownerWildcards.size() > 8 || geoWildcards.size() > 8) {
        err = "at most 8 owner and 8 geometry wildcards are supported";
        return 1;
    }
    for (size_t i = 0; i < wildcards.size(); i++) {
        if (!valid_identifier(wildcards[i]) || reserved_name(wildcards[i])) {
            err = "contact wildcard name '" + wildcards[i] + "' is not a usable identifier (it clashes with a force-model ingredient)";
            return 1;
        }
        for (size_t j = 0; j < i; j++)
            if (wildcards[i] == wildcards[j]) {
                err = "contact wildcard '" + wildcards[i] + "' is declared twice";
                return 1;
            }
    }
    std::ostringstream o;
    // the owner-tile form of the force pass (deme_tile.h) is compiled with the user's model too: DEME_JIT_NW wildcards ride in the
    // kernel's history stream
    o << "#define DEME_JIT 1\n#define DEME_TILE_OCC 4\n#define DEME_JIT_NW " << std::max<size_t>(wildcards.size(), 1) << "\n#define DEME_JIT_HAS_WC " << (wildcards.empty() ? 0 : 1)
      << "\n#include \"deme_tile.h\"\n" << kVocabulary << "\n";
    emit_array(o, "E", mt.E, mt.nMat, false);
    emit_array(o, "nu", mt.nu, mt.nMat, false);
    emit_array(o, "CoR", mt.CoR, mt.nMat, true);
    emit_array(o, "mu", mt.mu, mt.nMat, true);
    emit_array(o, "Crr", mt.Crr, mt.nMat, true);
    o << "// ---- _forceModelPrerequisites_\n" << prereq << "\n";
    o << "namespace deme_dev {\n__device__ void deme_user_model(UserModelIO& io) {\n";
    o << "    double overlapDepth = io.overlapDepth; float3 B2A = io.B2A; double3 contactPnt = io.contactPnt;\n"
         "    double3 AOwnerPos = io.AOwnerPos, BOwnerPos = io.BOwnerPos, bodyAPos = io.bodyAPos, bodyBPos = io.bodyBPos;\n"
         "    float AOwnerMass = io.AOwnerMass, BOwnerMass = io.BOwnerMass, ARadius = io.ARadius, BRadius = io.BRadius;\n"
         "    float4 AOriQ = io.AOriQ, BOriQ = io.BOriQ;\n"
         "    deme::materialsOffset_t bodyAMatType = io.bodyAMatType, bodyBMatType = io.bodyBMatType;\n"
         "    deme::contact_t ContactType = io.ContactType; deme::family_t AOwnerFamily = io.AOwnerFamily, BOwnerFamily = io.BOwnerFamily;\n"
         "    float3 locCPA = io.locCPA, locCPB = io.locCPB, force = io.force, torque_only_force = io.torque_only_force;\n"
         "    float ts = io.ts; float time = io.time;\n"
         "    float3 ALinVel = io.ALinVel, BLinVel = io.BLinVel, ARotVel = io.ARotVel, BRotVel = io.BRotVel;\n"
         "    float3 AOwnerMOI = io.AOwnerMOI, BOwnerMOI = io.BOwnerMOI;\n"
         "    deme::bodyID_t AOwner = io.AOwner, BOwner = io.BOwner, AGeo = io.AGeo, BGeo = io.BGeo; unsigned int myContactID = io.myContactID;\n"
         "    (void)overlapDepth; (void)B2A; (void)contactPnt; (void)AOwnerPos; (void)BOwnerPos; (void)bodyAPos; (void)bodyBPos;\n"
         "    (void)AOwnerMass; (void)BOwnerMass; (void)ARadius; (void)BRadius; (void)AOriQ; (void)BOriQ; (void)bodyAMatType;\n"
         "    (void)bodyBMatType; (void)ContactType; (void)AOwnerFamily; (void)BOwnerFamily; (void)locCPA; (void)locCPB; (void)ts;\n"
         "    (void)time; (void)ALinVel; (void)BLinVel; (void)ARotVel; (void)BRotVel; (void)AOwnerMOI; (void)BOwnerMOI; (void)AOwner;\n"
         "    (void)BOwner; (void)AGeo; (void)BGeo; (void)myContactID;\n";
    for (size_t i = 0; i < wildcards.size(); i++)  // _forceModelContactWildcardAcq_
        o << "    float " << wildcards[i] << " = io.wc[" << i << "];\n";
    // owner wildcards are aliases of the per-owner arrays, geometry wildcards come as name_A / name_B (equip_owner_wildcards,
    // equip_geo_wildcards, Models.h:319-360): updating them -- atomically if need be -- is the fragment's business
    for (size_t i = 0; i < ownerWildcards.size(); i++)
        o << "    float* " << ownerWildcards[i] << " = io.ownerWc[" << i << "]; float* " << ownerWildcards[i] << "_A = io.ownerWc[" << i
          << "]; float* " << ownerWildcards[i] << "_B = io.ownerWc[" << i << "]; (void)" << ownerWildcards[i] << "; (void)"
          << ownerWildcards[i] << "_A; (void)" << ownerWildcards[i] << "_B;\n";
    for (size_t i = 0; i < geoWildcards.size(); i++)
        o << "    float* " << geoWildcards[i] << "_A = io.geoWcA[" << i << "]; float* " << geoWildcards[i] << "_B = io.geoWcB[" << i
          << "]; (void)" << geoWildcards[i] << "_A; (void)" << geoWildcards[i] << "_B;\n";
    o << "    // ---- _DEMForceModel_ (user statement block, spliced literally)\n    {\n" << user << "\n    }\n";
    for (size_t i = 0; i < wildcards.size(); i++)  // _forceModelContactWildcardWrite_
        o << "    io.wc[" << i << "] = " << wildcards[i] << ";\n";
    o << "    io.force = force; io.torque_only_force = torque_only_force;\n}\n}  // namespace deme_dev\n";
    const char* ent[2] = {"ss", "sm"};
    for (int cls = 0; cls < 2; cls++)
        o << "extern \"C\" __global__ __launch_bounds__(256) void deme_custom_forces_" << ent[cls]
          << "(const deme_dev::DevParams p, const deme_dev::ForceArgs a) {\n    deme_dev::calc_forces_block<2, " << cls
          << ">(p, a);\n}\n";
    // the tile pass with the user's model where the Hertzian block is: the instances of the kernel templates of deme_tile.h are
    // named through function pointers (hipModuleGetFunction takes the mangled names, listed by deme_tile_entry_names below)
    o << "namespace deme_dev {\n"
         "template __global__ void k_tile_forces<2, false, false>(const DevParams, const TileArgs);\n"
         "template __global__ void k_tile_forces<2, true, false>(const DevParams, const TileArgs);\n"
         "template __global__ void k_tile_forces_big<2, false, false>(const DevParams, const TileArgs);\n"
         "template __global__ void k_tile_forces_big<2, true, false>(const DevParams, const TileArgs);\n}\n";
    out = o.str();
    return 0;
}

// Family motion prescriptions (DEMSolver::SetFamilyPrescribedLinVel & co.; code generation of
// DEM/APIPrivate.cpp:1600-1708 equipFamilyPrescribedMotions).  The three strings are the BODIES of the
// `switch (family)` statements of applyPrescribedVel / applyPrescribedPos / applyAddedAcceleration
// (kernel/DEMIntegrationKernels.cu:8-98), i.e. "case 3: { ... break; }" sequences written against the reference's
// parameter names.  The generated kernel runs over the owners of prescribed families only and leaves one PrescRec
// each for k_integrate.
inline void generate_prescribe_source(const std::string& velCases, const std::string& posCases, const std::string& accCases,
                                      std::string& out) {
    std::ostringstream o;
    o << "#include \"deme_device.h\"\n" << kVocabulary << R"DEMEPRE(
namespace deme_dev {
__device__ inline void applyPrescribedVel(bool& LinVelXPrescribed, bool& LinVelYPrescribed, bool& LinVelZPrescribed,
                                          bool& RotVelXPrescribed, bool& RotVelYPrescribed, bool& RotVelZPrescribed, float& vX,
                                          float& vY, float& vZ, float& omgBarX, float& omgBarY, float& omgBarZ, double X, double Y,
                                          double Z, float oriQw, float oriQx, float oriQy, float oriQz, deme::bodyID_t ownerID,
                                          const deme::family_t& family, const float& t) {
    switch (family) {
)DEMEPRE" << velCases << R"DEMEPRE(
        default:
            return;
    }
}
__device__ inline void applyPrescribedPos(bool& LinXPrescribed, bool& LinYPrescribed, bool& LinZPrescribed, bool& RotPrescribed,
                                          double& X, double& Y, double& Z, float& oriQw, float& oriQx, float& oriQy, float& oriQz,
                                          float vX, float vY, float vZ, float omgBarX, float omgBarY, float omgBarZ,
                                          deme::bodyID_t ownerID, const deme::family_t& family, const float& t) {
    switch (family) {
)DEMEPRE" << posCases << R"DEMEPRE(
        default:
            return;
    }
}
__device__ inline void applyAddedAcceleration(float& accX, float& accY, float& accZ, float& angAccX, float& angAccY, float& angAccZ,
                                              double X, double Y, double Z, float oriQw, float oriQx, float oriQy, float oriQz,
                                              float vX, float vY, float vZ, float omgBarX, float omgBarY, float omgBarZ,
                                              deme::bodyID_t ownerID, const deme::family_t& family, const float& t) {
    switch (family) {
)DEMEPRE" << accCases << R"DEMEPRE(
        default:
            return;
    }
}
}  // namespace deme_dev
extern "C" __global__ __launch_bounds__(256) void deme_prescribe(const deme_dev::DevParams p, const deme_dev::OwnerRec* owners,
                                                                 const uint32_t* list, uint32_t n, deme_dev::PrescRec* out,
                                                                 float t) {
    using namespace deme_dev;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const uint32_t slot = list[i];
    OwnerRec r = load_owner(owners, slot);
    const uint32_t ownerID = p.o2e ? p.o2e[slot] : slot;  // what user code sees is the caller's number of the owner
    const deme::family_t family = (deme::family_t)r.family;
    d3 P = decode_pos(r.voxelID, r.locX, r.locY, r.locZ, p);
    double X = P.x + (double)p.LBFX, Y = P.y + (double)p.LBFY, Z = P.z + (double)p.LBFZ;
    bool lvx = false, lvy = false, lvz = false, rvx = false, rvy = false, rvz = false, lx = false, ly = false, lz = false, rot = false;
    applyPrescribedVel(lvx, lvy, lvz, rvx, rvy, rvz, r.vx, r.vy, r.vz, r.wx, r.wy, r.wz, X, Y, Z, r.qw, r.qx, r.qy, r.qz, ownerID, family, t);
    applyPrescribedPos(lx, ly, lz, rot, X, Y, Z, r.qw, r.qx, r.qy, r.qz, r.vx, r.vy, r.vz, r.wx, r.wy, r.wz, ownerID, family, t);
    PrescRec q;
    q.ax = q.ay = q.az = q.lx = q.ly = q.lz = 0.f;
    applyAddedAcceleration(q.ax, q.ay, q.az, q.lx, q.ly, q.lz, X, Y, Z, r.qw, r.qx, r.qy, r.qz, r.vx, r.vy, r.vz, r.wx, r.wy, r.wz,
                           ownerID, family, t);
    q.X = X, q.Y = Y, q.Z = Z;
    q.vx = r.vx, q.vy = r.vy, q.vz = r.vz, q.wx = r.wx, q.wy = r.wy, q.wz = r.wz;
    q.qw = r.qw, q.qx = r.qx, q.qy = r.qy, q.qz = r.qz;
    q.flags = (lvx ? 1u : 0u) | (lvy ? 2u : 0u) | (lvz ? 4u : 0u) | (rvx ? 8u : 0u) | (rvy ? 16u : 0u) | (rvz ? 32u : 0u) |
              (lx ? 64u : 0u) | (ly ? 128u : 0u) | (lz ? 256u : 0u) | (rot ? 512u : 0u);
    q.pad = 0u;
    out[i] = q;
}
)DEMEPRE";
    out = o.str();
}

// On-the-fly family changes (DEMSolver::ChangeFamilyWhen; code generation equipFamilyOnFlyChanges,
// DEM/APIPrivate.cpp:1576-1598; kernel applyFamilyChanges, kernel/DEMModeratorKernels.cu:10-60).  `rules` is the
// reference's _familyChangeRules_ text: a sequence of
//   if (family_code == A) { bool shouldMakeChange = false; <user code> if (shouldMakeChange) {granData->familyID[myOwner] = B;} }
// written against pos / vel / acc / mass / X..accZ / ts / time.  granData->familyID[myOwner] is an lvalue proxy here.
inline void generate_family_rules_source(const std::string& rules, std::string& out) {
    std::ostringstream o;
    o << "#include \"deme_device.h\"\n" << kVocabulary << R"DEMEFAM(
namespace deme_dev {
struct DemeFamRef {
    deme::family_t* p;
    __device__ deme::family_t& operator[](size_t) const { return *p; }
};
struct DemeGranProxy {
    DemeFamRef familyID;
};
}  // namespace deme_dev
extern "C" __global__ __launch_bounds__(256) void deme_family_changes(const deme_dev::DevParams p, deme_dev::OwnerRec* owners,
                                                                      const deme_dev::AccRec* accArr, uint32_t nOwnerBodies,
                                                                      float timeElapsed) {
    using namespace deme_dev;
    const deme::bodyID_t myOwner = blockIdx.x * blockDim.x + threadIdx.x;
    if (myOwner >= nOwnerBodies)
        return;
    const OwnerRec r = load_owner(owners, myOwner);
    double3 pos;
    float3 vel, acc;
    const float mass = p.massProps[r.inertiaOff].x;
    deme::family_t family_code = (deme::family_t)r.family;
    deme::family_t deme_new_family = family_code;
    const d3 P = decode_pos(r.voxelID, r.locX, r.locY, r.locZ, p);
    pos = make_double3(P.x + p.LBFX, P.y + p.LBFY, P.z + p.LBFZ);
    vel = make_float3(r.vx, r.vy, r.vz);
    acc = make_float3(0.f, 0.f, 0.f);
    if (accArr)
        acc = make_float3(accArr[myOwner].ax, accArr[myOwner].ay, accArr[myOwner].az);
    double X = pos.x, Y = pos.y, Z = pos.z;
    float vX = vel.x, vY = vel.y, vZ = vel.z;
    float accX = acc.x, accY = acc.y, accZ = acc.z;
    float ts = p.h;
    float time = timeElapsed;
    DemeGranProxy deme_gran_proxy{{&deme_new_family}};
    DemeGranProxy* granData = &deme_gran_proxy;
    (void)mass, (void)X, (void)Y, (void)Z, (void)vX, (void)vY, (void)vZ, (void)accX, (void)accY, (void)accZ, (void)ts, (void)time;
    {
)DEMEFAM" << rules << R"DEMEFAM(
    }
    if (deme_new_family != family_code)
        owners[myOwner].family = (r.family & OWNER_FLAG_BITS) | (uint32_t)deme_new_family;  // ghost / replicated-owner bits stay
}
)DEMEFAM";
    out = o.str();
}

// Inspector region filter (DEM/AuxClasses.cpp:205-223 `_inRegionPolicy_` in DEMSphereQueryKernels.cu:41-47 /
// DEMOwnerQueryKernels.cu:47-53): the user's statement block returns a bool from float X, Y, Z (sphere centre or owner CoM,
// domain offset included).  It runs as the body of a lambda, so its own `return` statements work unedited; an element
// outside the region gets the reduction's identity.
inline void generate_region_source(const std::string& code, std::string& out) {
    std::ostringstream o;
    o << "#include \"deme_device.h\"\n" << kVocabulary << R"DEMEREG(
extern "C" __global__ __launch_bounds__(256) void deme_region_filter(const deme_dev::DevParams p, const deme_dev::OwnerRec* owners,
                                                                     const deme_dev::SphereRec* spheres, uint32_t n,
                                                                     uint32_t perSphere, float identity, float* values) {
    using namespace deme_dev;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    uint32_t myOwner = i;
    double rx = 0., ry = 0., rz = 0.;
    if (perSphere) {
        const SphereRec sr = spheres[i];
        myOwner = sr.owner;
        const OwnerRec q = load_owner(owners, myOwner);
        const float4 c = p.comp[sr.comp];
        const f3 rel = rot_apply(rot_coeffs(q.qw, q.qx, q.qy, q.qz), mk3(c.x, c.y, c.z));
        rx = (double)rel.x, ry = (double)rel.y, rz = (double)rel.z;
    }
    const OwnerRec r = load_owner(owners, myOwner);
    const d3 P = decode_pos(r.voxelID, r.locX, r.locY, r.locZ, p);
    const float X = perSphere ? (float)(P.x + rx + (double)p.LBFX) : (float)(P.x + (double)p.LBFX);
    const float Y = perSphere ? (float)(P.y + ry + (double)p.LBFY) : (float)(P.y + (double)p.LBFY);
    const float Z = perSphere ? (float)(P.z + rz + (double)p.LBFZ) : (float)(P.z + (double)p.LBFZ);
    (void)X, (void)Y, (void)Z;
    const bool isInRegion = [&]() -> bool {
)DEMEREG" << code << R"DEMEREG(
    }();
    if (!isInRegion)
        values[i] = identity;
}
)DEMEREG";
    out = o.str();
}

// mangled names of the four tile-pass instances of a compiled model (plain / with mesh records, fitting tiles / the others)
static const char* const kTileEntry[4] = {"_ZN8deme_dev13k_tile_forcesILi2ELb0ELb0EEEvNS_9DevParamsENS_8TileArgsE",
                                          "_ZN8deme_dev13k_tile_forcesILi2ELb1ELb0EEEvNS_9DevParamsENS_8TileArgsE",
                                          "_ZN8deme_dev17k_tile_forces_bigILi2ELb0ELb0EEEvNS_9DevParamsENS_8TileArgsE",
                                          "_ZN8deme_dev17k_tile_forces_bigILi2ELb1ELb0EEEvNS_9DevParamsENS_8TileArgsE"};

// hipRTC: source -> gfx950 code object.  Needs no GPU (used by the CPU test through deme_jit_probe).
inline int compile(const std::string& src, std::vector<char>& code, std::string& log) {
    hiprtcProgram prog;
    const char* hdrs[] = {kSrcDeviceH, kSrcMeshH, kSrcForceH, kSrcForceFastH, kSrcTileH};
    const char* names[] = {"deme_device.h", "deme_mesh.h", "deme_force.h", "deme_force_fast.h", "deme_tile.h"};
    if (hiprtcCreateProgram(&prog, src.c_str(), "deme_custom_force_model.hip", 5, hdrs, names) != HIPRTC_SUCCESS) {
        log = "hiprtcCreateProgram failed";
        return 1;
    }
    // user kernel includes (DEMSolver::AddKernelInclude, DEM/API.h:1362-1367) resolve against the ROCm installation and against the
    // directories of DEME_KERNEL_INCLUDE_PATH (':'-separated), like the reference's jitify include path
    std::vector<std::string> optStr = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize", "-I/opt/rocm/include"};
    if (const char* e = getenv("DEME_KERNEL_INCLUDE_PATH")) {
        std::string all = e;
        size_t pos = 0;
        while (pos <= all.size()) {
            const size_t end = all.find(':', pos);
            const std::string dir = all.substr(pos, end == std::string::npos ? std::string::npos : end - pos);
            if (!dir.empty())
                optStr.push_back("-I" + dir);
            if (end == std::string::npos)
                break;
            pos = end + 1;
        }
    }
    std::vector<const char*> opts;
    for (auto& o : optStr)
        opts.push_back(o.c_str());
    const hiprtcResult rc = hiprtcCompileProgram(prog, (int)opts.size(), opts.data());
    size_t ls = 0;
    hiprtcGetProgramLogSize(prog, &ls);
    if (ls > 1) {
        log.resize(ls);
        hiprtcGetProgramLog(prog, &log[0]);
    }
    if (rc != HIPRTC_SUCCESS) {
        if (log.empty())
            log = hiprtcGetErrorString(rc);
        hiprtcDestroyProgram(&prog);
        return 1;
    }
    size_t cs = 0;
    hiprtcGetCodeSize(prog, &cs);
    code.resize(cs);
    hiprtcGetCode(prog, code.data());
    hiprtcDestroyProgram(&prog);
    if (const char* dump = getenv("DEME_JIT_DUMP")) {  // (debugging: the code object, e.g. for llvm-readelf / tools/co_info.py)
        if (FILE* f = fopen(dump, "wb")) {
            fwrite(code.data(), 1, code.size(), f);
            fclose(f);
        }
    }
    return 0;
}

}  // namespace deme_jit
