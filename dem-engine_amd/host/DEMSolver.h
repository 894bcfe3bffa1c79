// DEMSolver.h -- header-only C++ shell of deme::DEMSolver above the C-ABI (include/deme_hip.h).
//
// Same class / method names and argument meaning as the reference's scripting API for the set-up and
// stepping calls the BASELINE configurations use (reference: src/DEM/API.h:50-1300, APIPublic.cpp,
// BdrsAndObjs.h, Structs.h, AuxClasses.h:422-485), so demo-style programs compile against this header and
// run on libdeme_hip.so.  What it does NOT carry: writers, trackers/inspectors, OBJ/CSV mesh loaders,
// family prescriptions other than "fixed" (SURVEY 8f).  Errors are std::runtime_error thrown from the
// calling thread, as in the reference (Structs.h:285-295).
//
// Host-side preprocessing follows the reference where it defines kernel inputs:
//   voxel bit split / l          APIPrivate.cpp:373-487 (figureOutNV)
//   bin size / counts            APIPrivate.cpp:489-566, HostSideHelpers.hpp:195-207
//   world bounding planes        APIPrivate.cpp:955-1014
//   template order, owner order  APIPrivate.cpp:696-742, dT.cpp:700-800
//   pairwise material matrices   APIPrivate.cpp:1877-2026
#pragma once
#include <algorithm>
#include <array>
#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <iostream>
#include <functional>
#include <map>
#include <memory>
#include <set>
#include <sstream>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "deme_hip.h"

#ifndef DEME_HOST_VECTOR_TYPES
#define DEME_HOST_VECTOR_TYPES
struct float3 {
    float x, y, z;
};
struct float4 {
    float x, y, z, w;
};
inline float3 make_float3(float x, float y, float z) { return {x, y, z}; }
inline float3 make_float3(float s) { return {s, s, s}; }
inline float4 make_float4(float x, float y, float z, float w) { return {x, y, z, w}; }
inline float3 operator*(float3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline float3 operator+(float3 a, float3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline float3 operator-(float3 a, float3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
// the rest of the float3 algebra the reference's scripts use (core/utils/CUDAMathHelpers.cuh on the host side)
inline float3 operator*(float s, float3 a) { return {a.x * s, a.y * s, a.z * s}; }
inline float3 operator*(float3 a, float3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline float3 operator/(float3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
inline float3 operator/(float3 a, float3 b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }
inline float3 operator-(float3 a) { return {-a.x, -a.y, -a.z}; }
inline float3& operator+=(float3& a, float3 b) { a.x += b.x, a.y += b.y, a.z += b.z; return a; }
inline float3& operator-=(float3& a, float3 b) { a.x -= b.x, a.y -= b.y, a.z -= b.z; return a; }
inline float3& operator*=(float3& a, float s) { a.x *= s, a.y *= s, a.z *= s; return a; }
inline float dot(float3 a, float3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline float3 cross(float3 a, float3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline float length(float3 a) { return std::sqrt(dot(a, a)); }
inline float3 normalize(float3 a) { return a / length(a); }
#endif

namespace deme {

constexpr double PI = 3.1415926535897932385;  // DEM/Defines.h:43

// Where the data files of the scripts live (reference: GET_DATA_PATH() = the build's data directory, DEM/Models.h:177;
// GetDEMEDataFile, core/utils/DEMEPaths.h:32).  Here: $DEME_DATA_PATH, else ./data.
inline std::filesystem::path GET_DATA_PATH() {
    const char* e = std::getenv("DEME_DATA_PATH");
    return std::filesystem::path(e ? e : "data");
}
inline std::string GetDEMEDataFile(const std::string& filename) { return (GET_DATA_PATH() / filename).string(); }

enum class TIME_INTEGRATOR { FORWARD_EULER, CENTERED_DIFFERENCE, EXTENDED_TAYLOR };
enum class FORCE_MODEL { HERTZIAN, HERTZIAN_FRICTIONLESS, CUSTOM };
enum VERBOSITY { QUIET = 0, ERR = 10, WARNING = 20, INFO = 30, STEP_METRIC = 35, DEBUG = 40 };
// output content flags and file column names: DEM/Defines.h:152-183, DEM/Structs.h:41-97
enum OUTPUT_CONTENT { XYZ = 0, QUAT = 1, ABSV = 2, VEL = 4, ANG_VEL = 8, ABS_ACC = 16, ACC = 32, ANG_ACC = 64, FAMILY = 128, MAT = 256,
                      OWNER_WILDCARD = 512, GEO_WILDCARD = 1024 };
enum CNT_OUTPUT_CONTENT { CNT_TYPE = 0, FORCE = 1, CNT_POINT = 2, COMPONENT = 4, NORMAL = 8, TORQUE = 16, CNT_WILDCARD = 32, OWNER = 64,
                          GEO_ID = 128, NICKNAME = 256 };
enum class OUTPUT_FORMAT { CSV, BINARY, CHPF };
enum class MESH_FORMAT { VTK, OBJ };
typedef unsigned int bodyID_t;
typedef uint8_t notStupidBool_t;
constexpr float DEME_TINY_FLOAT_HOST = 1e-12f;
constexpr float DEME_TINY_FLOAT = 1e-12f;  // DEM/Defines.h
constexpr float DEME_HUGE_FLOAT = 1e15f;
constexpr unsigned int RESERVED_FAMILY_NUM = 255;
const bool ENTITY_NORMAL_INWARD = false;
const bool ENTITY_NORMAL_OUTWARD = true;

struct DEMMaterial {
    std::unordered_map<std::string, float> mat_prop;
    unsigned int load_order = 0;
};

struct DEMClumpTemplate {
    float mass = 0;
    float3 MOI{0, 0, 0};
    float volume = 0;  // SetVolume: read by the "clump_volume" inspector only
    std::vector<float> radii;
    std::vector<float3> relPos;
    std::vector<std::shared_ptr<DEMMaterial>> materials;
    unsigned int nComp = 0, mark = 0;
    std::string m_name;  // AssignName (Structs.h:697); default "%04d" of the load order (APIPublic.cpp:1751-1755)
    void AssignName(const std::string& n) { m_name = n; }
    void SetVolume(float v) { volume = v; }
    // x,y,z,r rows, '#' comments (data/clumps/*.csv in the reference)
    void ReadComponentFromFile(const std::string& file) {
        std::ifstream in(file);
        if (!in)
            throw std::runtime_error("clump file " + file + " not found");
        std::string line;
        while (std::getline(in, line)) {
            if (line.empty() || line[0] == '#' || line[0] == 'x')
                continue;
            std::replace(line.begin(), line.end(), ',', ' ');
            std::istringstream ss(line);
            float x, y, z, r;
            if (ss >> x >> y >> z >> r) {
                relPos.push_back({x, y, z});
                radii.push_back(r);
            }
        }
        nComp = (unsigned)radii.size();
    }
    void Scale(float s) {  // Structs.h:682-695: lengths*s, mass*s^3, MOI*s^5 -- with the reference's types (the powers in double)
        const double ps = (double)std::abs(s);
        mass *= ps * ps * ps;
        volume *= ps * ps * ps;
        const float s5 = (float)(ps * ps * ps * ps * ps);
        MOI = {MOI.x * s5, MOI.y * s5, MOI.z * s5};
        for (auto& r : radii)
            r *= s;
        for (auto& p : relPos)
            p = p * s;
    }
};

struct DEMClumpBatch {
    std::vector<std::shared_ptr<DEMClumpTemplate>> types;
    std::vector<float3> xyz, vel, angVel;
    std::vector<float4> oriQ;
    std::vector<unsigned int> families;
    size_t nClumps = 0;
    explicit DEMClumpBatch(size_t n) : nClumps(n) {
        vel.assign(n, {0, 0, 0});
        angVel.assign(n, {0, 0, 0});
        oriQ.assign(n, {0, 0, 0, 1});
        families.assign(n, 0);
    }
    void SetTypes(const std::vector<std::shared_ptr<DEMClumpTemplate>>& t) { check_len(t.size(), "SetTypes"), types = t; }
    void SetTypes(const std::shared_ptr<DEMClumpTemplate>& t) { types.assign(nClumps, t); }
    void SetType(const std::shared_ptr<DEMClumpTemplate>& t) { SetTypes(t); }
    void SetPos(const std::vector<float3>& p) { check_len(p.size(), "SetPos"), xyz = p; }
    void SetPos(float3 p) { xyz.assign(nClumps, p); }
    void SetVel(const std::vector<float3>& v) { check_len(v.size(), "SetVel"), vel = v; }
    void SetVel(float3 v) { vel.assign(nClumps, v); }
    void SetAngVel(const std::vector<float3>& v) { check_len(v.size(), "SetAngVel"), angVel = v; }
    void SetAngVel(float3 v) { angVel.assign(nClumps, v); }
    void SetOriQ(const std::vector<float4>& q) { check_len(q.size(), "SetOriQ"), oriQ = q; }
    void SetOriQ(float4 q) { oriQ.assign(nClumps, q); }
    void check_len(size_t n, const char* who) const {
        if (n != nClumps)
            throw std::runtime_error(std::string(who) + ": the input has " + std::to_string(n) + " entries but this batch has " +
                                     std::to_string(nClumps) + " clumps");
    }
    // initial geometry-wildcard values, one per sphere of the batch in clump-major order (Structs.h:908-930)
    void SetGeometryWildcards(const std::unordered_map<std::string, std::vector<float>>& w) {
        for (auto& kv : w)
            AddGeometryWildcard(kv.first, kv.second);
    }
    void AddGeometryWildcard(const std::string& name, const std::vector<float>& vals) {
        if (vals.size() != GetNumSpheres())
            throw std::runtime_error("Input geometry wildcard array in a AddGeometryWildcard call must have the same size as the number "
                                     "of spheres in this batch.");
        geo_wildcards[name] = vals;
    }
    void AddGeometryWildcard(const std::string& name, float val) { AddGeometryWildcard(name, std::vector<float>(GetNumSpheres(), val)); }
    std::unordered_map<std::string, std::vector<float>> geo_wildcards;
    void SetFamilies(const std::vector<unsigned int>& f) { families = f; }
    void SetFamily(unsigned int f) { families.assign(nClumps, f); }
    void SetFamilies(unsigned int f) { SetFamily(f); }
    size_t GetNumClumps() const { return nClumps; }
    size_t GetNumSpheres() const {
        size_t n = 0;
        for (auto& t : types)
            n += t->nComp;
        return n;
    }
    // initial owner-wildcard values of the batch's clumps (Structs.h:885-906); applied once the force model is compiled
    void SetOwnerWildcards(const std::unordered_map<std::string, std::vector<float>>& w) {
        for (auto& kv : w)
            AddOwnerWildcard(kv.first, kv.second);
    }
    void AddOwnerWildcard(const std::string& name, const std::vector<float>& vals) {
        if (vals.size() != nClumps)
            throw std::runtime_error("Input owner wildcard array in a AddOwnerWildcard call must have the same size as the number of "
                                     "clumps in this batch.\nHere, the input array has length " + std::to_string(vals.size()) +
                                     " but this batch has " + std::to_string(nClumps) + " clumps.");
        owner_wildcards[name] = vals;
    }
    void AddOwnerWildcard(const std::string& name, float val) { AddOwnerWildcard(name, std::vector<float>(nClumps, val)); }
    std::unordered_map<std::string, std::vector<float>> owner_wildcards;
    // restart data (Structs.h:857-880): sphere-sphere pairs by geometry id within this batch + their wildcards
    void SetExistingContacts(const std::vector<std::pair<bodyID_t, bodyID_t>>& pairs) { contact_pairs = pairs; }
    void SetExistingContactWildcards(const std::unordered_map<std::string, std::vector<float>>& w) {
        for (auto& kv : w)
            if (kv.second.size() != contact_pairs.size())
                throw std::runtime_error("SetExistingContactWildcards needs to be called after SetExistingContacts, with each "
                                         "wildcard array having the same length as the number of contact pairs.");
        contact_wildcards = w;
    }
    std::vector<std::pair<bodyID_t, bodyID_t>> contact_pairs;
    std::unordered_map<std::string, std::vector<float>> contact_wildcards;
};

struct DEMExternObj {
    struct Comp {
        uint8_t type;
        float3 pos, dir;
        float size1;
        float normal_sign;
        std::shared_ptr<DEMMaterial> mat;
    };
    std::vector<Comp> comps;
    unsigned int family_code = RESERVED_FAMILY_NUM;
    float3 init_pos{0, 0, 0};
    float4 init_oriQ{0, 0, 0, 1};
    float mass = 1e6f;
    float3 MOI{1e6f, 1e6f, 1e6f};
    static float3 unit(float3 n) {  // normalize(): v * rsqrtf(dot)
        const float il = 1.0f / std::sqrt(n.x * n.x + n.y * n.y + n.z * n.z);
        return {n.x * il, n.y * il, n.z * il};
    }
    void AddPlane(float3 pos, float3 normal, const std::shared_ptr<DEMMaterial>& m) {
        comps.push_back({DEME_ANAL_OBJ_TYPE_PLANE, pos, unit(normal), 0.f, 1.f, m});
    }
    void AddCylinder(float3 pos, float3 axis, float rad, const std::shared_ptr<DEMMaterial>& m, bool normal = ENTITY_NORMAL_INWARD) {
        comps.push_back({DEME_ANAL_OBJ_TYPE_CYL_INF, pos, unit(axis), rad, normal == ENTITY_NORMAL_INWARD ? 1.f : -1.f, m});
    }
    void AddZCylinder(float3 pos, float rad, const std::shared_ptr<DEMMaterial>& m, bool normal = ENTITY_NORMAL_INWARD) {
        AddCylinder(pos, {0, 0, 1}, rad, m, normal);
    }
    void SetFamily(unsigned int f) { family_code = f; }
    void SetFamilies(unsigned int f) { family_code = f; }
    void SetInitPos(float3 p) { init_pos = p; }
    void SetMass(float m) { mass = m; }
    void SetMOI(float3 m) { MOI = m; }
};

struct DEMMeshConnected {
    std::vector<float3> vertices;
    std::vector<std::array<int, 3>> faces;
    std::shared_ptr<DEMMaterial> mat;
    unsigned int family_code = RESERVED_FAMILY_NUM;
    float3 init_pos{0, 0, 0};
    float4 init_oriQ{0, 0, 0, 1};
    float mass = 1.f;
    float3 MOI{1, 1, 1};
    size_t GetNumTriangles() const { return faces.size(); }
    size_t GetNumNodes() const { return vertices.size(); }
    /// Minimal Wavefront OBJ reader (the reference uses tinyobjloader through DEMMeshConnected::LoadWavefrontMesh,
    /// BdrsAndObjs.h): `v x y z` and `f a b c [d ...]` with 1-based (or negative, relative) indices in the v, v/vt, v//vn and
    /// v/vt/vn forms; polygons are fanned into triangles; normals / texture coordinates are not needed by the solver.
    bool LoadWavefrontMesh(const std::string& input_file) {
        std::ifstream f(input_file);
        if (!f)
            return false;
        vertices.clear(), faces.clear();
        std::string line;
        while (std::getline(f, line)) {
            std::istringstream ls(line);
            std::string tag;
            if (!(ls >> tag))
                continue;
            if (tag == "v") {
                float3 v{0, 0, 0};
                ls >> v.x >> v.y >> v.z;
                vertices.push_back(v);
            } else if (tag == "f") {
                std::vector<int> idx;
                std::string tok;
                while (ls >> tok) {
                    const int i = std::stoi(tok.substr(0, tok.find('/')));
                    idx.push_back(i > 0 ? i - 1 : (int)vertices.size() + i);
                }
                for (size_t k = 2; k < idx.size(); k++)
                    faces.push_back({idx[0], idx[k - 1], idx[k]});
            }
        }
        return !faces.empty();
    }
    void Scale(float s) {
        for (auto& v : vertices)
            v = v * s;
    }
    void Scale(float3 s) {  // per-axis (BdrsAndObjs.h: Scale(float3))
        for (auto& v : vertices)
            v = v * s;
    }
    void SetFamily(unsigned int f) { family_code = f; }
    void SetFamilies(unsigned int f) { family_code = f; }
    void SetInitPos(float3 p) { init_pos = p; }
    void SetInitQuat(float4 q) { init_oriQ = q; }
    void SetMass(float m) { mass = m; }
    void SetMOI(float3 m) { MOI = m; }
    const std::vector<float3>& GetCoordsVertices() const { return vertices; }
    std::vector<float3>& GetCoordsVertices() { return vertices; }
    const std::vector<std::array<int, 3>>& GetIndicesVertexes() const { return faces; }
    static void rotate_node(float3& v, float4 q) {  // applyOriQToVector3 (DEMHelperKernels.cuh:161-173); q = (x, y, z, w)
        const float w = q.w, x = q.x, y = q.y, z = q.z;
        const float ox = (2.0f * (w * w + x * x) - 1.0f) * v.x + (2.0f * (x * y - w * z)) * v.y + (2.0f * (x * z + w * y)) * v.z;
        const float oy = (2.0f * (x * y + w * z)) * v.x + (2.0f * (w * w + y * y) - 1.0f) * v.y + (2.0f * (y * z - w * x)) * v.z;
        const float oz = (2.0f * (x * z - w * y)) * v.x + (2.0f * (y * z + w * x)) * v.y + (2.0f * (w * w + z * z) - 1.0f) * v.z;
        v = {ox, oy, oz};
    }
    /// InformCentroidPrincipal (BdrsAndObjs.h:420): the nodes were given in a frame whose origin / axes are not the mesh's
    /// centroid / principal axes; move them into that frame (translate by -center, rotate by the inverse of prin_Q)
    void InformCentroidPrincipal(float3 center, float4 prin_Q) {
        for (auto& n : vertices) {
            n = n - center;
            rotate_node(n, make_float4(-prin_Q.x, -prin_Q.y, -prin_Q.z, prin_Q.w));
        }
    }
    /// Move (BdrsAndObjs.h:437): rotate the nodes by rot_Q, then translate by vec
    void Move(float3 vec, float4 rot_Q) {
        for (auto& n : vertices) {
            rotate_node(n, rot_Q);
            n = n + vec;
        }
    }
    void Mirror(float3 plane_point, float3 plane_normal) {  // BdrsAndObjs.h: reflect across a plane, keep the facets' outward side
        const float inv = 1.0f / std::sqrt(plane_normal.x * plane_normal.x + plane_normal.y * plane_normal.y + plane_normal.z * plane_normal.z);
        const float3 nn = plane_normal * inv;
        for (auto& v : vertices) {
            const float3 d = v - plane_point;
            const float t = d.x * nn.x + d.y * nn.y + d.z * nn.z;
            v = v - nn * (2.f * t);
        }
        for (auto& f : faces)
            std::swap(f[1], f[2]);
    }
};

/// Per-contact information of the current contact list, one entry per contact pair (reference: ContactInfoContainer,
/// DEM/Structs.h:1049-1107; filled by DEMSolver::GetContactDetailedInfo).  A field exists when the contact output content
/// (SetContactOutputContent) asks for it -- the contact type and the owners' families always do; asking for an absent field
/// throws like the reference's on_missing_key.
class ContactInfoContainer {
  public:
    explicit ContactInfoContainer(unsigned int cnt_out_content) : m_content(cnt_out_content) {}
    std::vector<std::string>& GetContactType() { return m_type; }
    std::vector<float3>& GetPoint() { return need(CNT_POINT, "Point"), m_point; }
    std::vector<bodyID_t>& GetAOwner() { return need(OWNER, "AOwner"), m_ownerA; }
    std::vector<bodyID_t>& GetBOwner() { return need(OWNER, "BOwner"), m_ownerB; }
    std::vector<bodyID_t>& GetAGeo() { return need(GEO_ID, "AGeo"), m_geoA; }
    std::vector<bodyID_t>& GetBGeo() { return need(GEO_ID, "BGeo"), m_geoB; }
    std::vector<uint8_t>& GetAOwnerFamily() { return m_famA; }
    std::vector<uint8_t>& GetBOwnerFamily() { return m_famB; }
    std::vector<float3>& GetForce() { return need(FORCE, "Force"), m_force; }
    std::vector<float3>& GetTorque() { return need(TORQUE, "Torque"), m_torque; }
    std::vector<float3>& GetNormal() { return need(NORMAL, "Normal"), m_normal; }
    /// a contact wildcard by name (the reference's Get<float>(name))
    std::vector<float>& GetWildcard(const std::string& name) {
        auto it = m_wc.find(name);
        if (it == m_wc.end())
            missing(name);
        return it->second;
    }
    size_t Size() const { return m_type.size(); }

  private:
    friend class DEMSolver;
    void need(unsigned int bit, const char* key) const {
        if (!(m_content & bit))
            missing(key);
    }
    [[noreturn]] static void missing(const std::string& key) {
        throw std::runtime_error("ContactInfoContainer does not have field: '" + key +
                                 "', you may need to turn on the output of this field by correctly calling "
                                 "SetContactOutputContent before Initialize().");
    }
    unsigned int m_content;
    std::vector<std::string> m_type;
    std::vector<float3> m_point, m_force, m_torque, m_normal;
    std::vector<bodyID_t> m_ownerA, m_ownerB, m_geoA, m_geoB;
    std::vector<uint8_t> m_famA, m_famB;
    std::map<std::string, std::vector<float>> m_wc;
};

class DEMForceModel {
  public:
    FORCE_MODEL type = FORCE_MODEL::HERTZIAN;
    std::string code, prerequisites;
    std::set<std::string> contact_wildcards;  // std::set: alphabetical indices, as in the reference (Models.h:363)
    std::set<std::string> pairwise_props{"CoR", "mu", "Crr"};
    void SetForceModelType(FORCE_MODEL t) {
        type = t;
        contact_wildcards.clear();
        if (t == FORCE_MODEL::HERTZIAN)
            contact_wildcards = {"delta_tan_x", "delta_tan_y", "delta_tan_z", "delta_time"};
    }
    void DefineCustomModel(const std::string& model) {
        type = FORCE_MODEL::CUSTOM;
        code = model;
    }
    int ReadCustomModelFile(const std::string& path) {
        std::ifstream in(path);
        if (!in)
            return 1;
        std::stringstream ss;
        ss << in.rdbuf();
        DefineCustomModel(ss.str());
        return 0;
    }
    void DefineCustomModelPrerequisites(const std::string& util) { prerequisites = util; }
    void SetMustPairwiseMatProp(const std::set<std::string>& props) { pairwise_props.insert(props.begin(), props.end()); }
    /// material properties every loaded material must then define (AuxClasses.h: SetMustHaveMatProp); checked at Initialize
    void SetMustHaveMatProp(const std::set<std::string>& props) { must_have_props.insert(props.begin(), props.end()); }
    std::set<std::string> must_have_props;
    void SetPerContactWildcards(const std::set<std::string>& wc) { contact_wildcards = wc; }
    void SetPerOwnerWildcards(const std::set<std::string>& wc) { owner_wildcards = wc; }
    void SetPerGeometryWildcards(const std::set<std::string>& wc) { geo_wildcards = wc; }
    std::set<std::string> owner_wildcards, geo_wildcards;
};

class DEMSolver {
  public:
    /// DEM/API.h:52-56, DEM/APIPublic.cpp:22-110: the reference picks its devices here (kT on one, dT on another).  This engine
    /// replaces that functional split by a spatial one: nGPUs devices = nGPUs x-/y-/z-slabs of the bed, one per device,
    /// exchanging ghost clumps over RCCL every step (deme_multi_*, csrc/deme_decomp.inc).  One device = one plain context.
    /// DEME_SLABS_PER_DEVICE=S (environment) cuts every device's part into S slabs -- the whole multi-device path on one GPU.
    explicit DEMSolver(unsigned int nGPUs = 1) {
        if (nGPUs < 1)
            throw std::runtime_error("DEMSolver: at least one GPU");
        std::vector<int> ids(nGPUs);
        for (unsigned i = 0; i < nGPUs; i++)
            ids[i] = (int)i;
        open_devices(ids);
    }
    explicit DEMSolver(const std::vector<int>& device_ids) { open_devices(device_ids); }
    ~DEMSolver() {
        if (m_multi)
            deme_multi_destroy(m_multi);
        else if (m_ctx)
            deme_ctx_destroy(m_ctx);
    }
    /// slabs per device of a decomposed run (not in the reference; the environment variable DEME_SLABS_PER_DEVICE does the same
    /// for an unchanged script), the clumps' migration interval in steps (0: never) and the ghost layer thickness (0: four clump
    /// reaches); every n-th migration may recompute the slab boundaries from where the clumps are by then (0: the boundaries stay
    /// where Initialize() put them).  Before Initialize().
    void SetSlabRebalanceInterval(unsigned int nthMigration) { m_rebalance_every = nthMigration; }
    void SetSlabsPerDevice(unsigned int s) { m_slabs_per_device = s < 1 ? 1 : s; }
    void SetSlabMigrationInterval(unsigned int steps) { m_migrate_every = steps; }
    void SetSlabHalo(float halo) { m_slab_halo = halo; }
    unsigned int GetNumSlabs() const {
        uint32_t n = 1;
        if (m_multi)
            deme_multi_num_slabs(m_multi, &n);
        return n;
    }
    const std::vector<int>& GetDeviceIDs() const { return m_devices; }
    DEMSolver(const DEMSolver&) = delete;
    DEMSolver& operator=(const DEMSolver&) = delete;

    void SetVerbosity(int) {}
    void SetVerbosity(VERBOSITY) {}
    void SetVerbosity(const std::string& verbose) {  // API.h:1337: the level names; this build only reports errors (exceptions)
        static const std::set<std::string> ok = {"QUIET", "ERROR", "WARNING", "INFO", "STEP_ANOMALY", "STEP_METRIC", "DEBUG", "STEP_DEBUG"};
        if (!ok.count(upper(verbose)))
            throw std::runtime_error("Instruction " + verbose + " is unknown in SetVerbosity call.");
    }
    // knobs of the reference's run-time compiler and of its two-thread scheduler: nothing to configure in this build
    // (templates and mass properties always live in device tables; there is no second thread to drift ahead of)
    void SetJitifyClumpTemplates(bool = true) {}
    void DisableJitifyClumpTemplates() {}
    void SetJitifyMassProperties(bool = true) {}
    void DisableJitifyMassProperties() {}
    void EnsureKernelErrMsgLineNum(bool = true) {}
    std::vector<std::string> GetJitifyOptions() const { return {"--offload-arch=gfx950", "-O3", "-std=c++17"}; }  // what hipRTC is given
    void SetJitifyOptions(const std::vector<std::string>&) {}
    // DEM/API.h:1362-1367, APIPublic.cpp:1638: text put in front of every run-time compiled force model (the reference's
    // _kernelIncludes_).  Its default there is CUDA's <curand_kernel.h>, which has nothing to name here: the default is empty, and a
    // script that adds that header by name gets hipRAND's device header, its counterpart in this toolchain.
    void AddKernelInclude(const std::string& lib_name) {
        m_kernel_includes += "#include <" + std::string(lib_name == "curand_kernel.h" ? "hiprand/hiprand_kernel.h" : lib_name) + ">\n";
    }
    void SetKernelInclude(const std::string& includes) { m_kernel_includes = includes; }
    void RemoveKernelInclude() { m_kernel_includes = " "; }
    void PrintKinematicScratchSpaceUsage() const {}
    void SetCDNumStepsMaxDriftAheadOfAvg(float) {}
    void SetCDNumStepsMaxDriftMultipleOfAvg(float) {}
    void SetCDNumStepsMaxDriftHistorySize(unsigned int) {}
    bool GetInitStatus() const { return m_initialized; }

    // ---- domain (APIPublic.cpp:845-904)
    void InstructBoxDomainDimension(float x, float y, float z) {
        InstructBoxDomainDimension({-x / 2, x / 2}, {-y / 2, y / 2}, {-z / 2, z / 2});
    }
    void InstructBoxDomainDimension(std::pair<float, float> x, std::pair<float, float> y, std::pair<float, float> z) {
        m_user_min = {std::min(x.first, x.second), std::min(y.first, y.second), std::min(z.first, z.second)};
        m_user_max = {std::max(x.first, x.second), std::max(y.first, y.second), std::max(z.first, z.second)};
        const float3 e = {(m_user_max.x - m_user_min.x) * 0.1f, (m_user_max.y - m_user_min.y) * 0.1f, (m_user_max.z - m_user_min.z) * 0.1f};
        m_target_min = m_user_min - e;
        m_target_max = m_user_max + e;
    }
    void InstructBoxDomainBoundingBC(const std::string& inst, const std::shared_ptr<DEMMaterial>& mat) {
        m_bounding = inst;
        m_bounding_mat = mat;
    }

    // ---- materials
    std::shared_ptr<DEMMaterial> LoadMaterial(const std::unordered_map<std::string, float>& props) {
        auto m = std::make_shared<DEMMaterial>();
        m->mat_prop = props;
        m->load_order = (unsigned)m_materials.size();
        m_materials.push_back(m);
        return m;
    }
    std::shared_ptr<DEMMaterial> LoadMaterial(DEMMaterial& a_material) { return LoadMaterial(a_material.mat_prop); }
    /// Duplicate (API.h:402-410, APIPublic.cpp:397-411): a deep copy, loaded as a new object
    std::shared_ptr<DEMMaterial> Duplicate(const std::shared_ptr<DEMMaterial>& ptr) { return LoadMaterial(ptr->mat_prop); }
    std::shared_ptr<DEMClumpTemplate> Duplicate(const std::shared_ptr<DEMClumpTemplate>& ptr) {
        auto t = std::make_shared<DEMClumpTemplate>(*ptr);
        m_templates.push_back(t);
        return t;
    }
    std::shared_ptr<DEMClumpBatch> Duplicate(const std::shared_ptr<DEMClumpBatch>& ptr) {
        auto b = std::make_shared<DEMClumpBatch>(*ptr);
        m_batches.push_back(b);
        return b;
    }
    void SetMaterialPropertyPair(const std::string& name, const std::shared_ptr<DEMMaterial>& a, const std::shared_ptr<DEMMaterial>& b, float v) {
        m_pair_overrides[name][{a->load_order, b->load_order}] = v;
        m_pair_overrides[name][{b->load_order, a->load_order}] = v;
    }

    // ---- templates
    std::shared_ptr<DEMClumpTemplate> LoadClumpType(float mass, float3 moi, const std::vector<float>& radii,
                                                    const std::vector<float3>& relPos, const std::shared_ptr<DEMMaterial>& mat) {
        auto t = std::make_shared<DEMClumpTemplate>();
        t->mass = mass, t->MOI = moi, t->radii = radii, t->relPos = relPos, t->nComp = (unsigned)radii.size();
        t->materials.assign(radii.size(), mat);
        m_templates.push_back(t);
        return t;
    }
    std::shared_ptr<DEMClumpTemplate> LoadClumpType(float mass, float3 moi, const std::string& file, const std::shared_ptr<DEMMaterial>& mat) {
        auto t = std::make_shared<DEMClumpTemplate>();
        t->mass = mass, t->MOI = moi;
        t->ReadComponentFromFile(file);
        t->materials.assign(t->radii.size(), mat);
        m_templates.push_back(t);
        return t;
    }
    std::shared_ptr<DEMClumpTemplate> LoadClumpType(float mass, float3 moi, const std::vector<float>& radii,
                                                    const std::vector<float3>& relPos,
                                                    const std::vector<std::shared_ptr<DEMMaterial>>& mats) {
        if (mats.size() != radii.size() || relPos.size() != radii.size())
            throw std::runtime_error("LoadClumpType: radii, positions and materials must have the same length");
        auto t = std::make_shared<DEMClumpTemplate>();
        t->mass = mass, t->MOI = moi, t->radii = radii, t->relPos = relPos, t->nComp = (unsigned)radii.size(), t->materials = mats;
        m_templates.push_back(t);
        return t;
    }
    std::shared_ptr<DEMClumpTemplate> LoadClumpType(DEMClumpTemplate& clump) {  // API.h:322: a copy of a user-built template
        auto t = std::make_shared<DEMClumpTemplate>(clump);
        t->nComp = (unsigned)t->radii.size();
        if (t->materials.size() != t->radii.size())
            throw std::runtime_error("LoadClumpType: the template's materials do not match its components");
        m_templates.push_back(t);
        return t;
    }
    std::shared_ptr<DEMClumpTemplate> LoadSphereType(float mass, float radius, const std::shared_ptr<DEMMaterial>& mat) {
        const float I = 2.f / 5.f * mass * radius * radius;
        return LoadClumpType(mass, {I, I, I}, std::vector<float>{radius}, std::vector<float3>{{0, 0, 0}}, mat);
    }

    // ---- entities
    std::shared_ptr<DEMClumpBatch> AddClumps(const std::vector<std::shared_ptr<DEMClumpTemplate>>& types, const std::vector<float3>& xyz) {
        if (types.size() != xyz.size())
            throw std::runtime_error("AddClumps: type and position arrays differ in length");
        auto b = std::make_shared<DEMClumpBatch>(xyz.size());
        b->types = types, b->xyz = xyz;
        m_batches.push_back(b);
        return b;
    }
    std::shared_ptr<DEMClumpBatch> AddClumps(const std::shared_ptr<DEMClumpTemplate>& type, const std::vector<float3>& xyz) {
        return AddClumps(std::vector<std::shared_ptr<DEMClumpTemplate>>(xyz.size(), type), xyz);
    }
    // the other input forms of API.h:588-634
    std::shared_ptr<DEMClumpBatch> AddClumps(const std::shared_ptr<DEMClumpTemplate>& type, float3 xyz) {
        return AddClumps(std::vector<std::shared_ptr<DEMClumpTemplate>>(1, type), std::vector<float3>(1, xyz));
    }
    std::shared_ptr<DEMClumpBatch> AddClumps(const std::shared_ptr<DEMClumpTemplate>& type, const std::vector<float>& xyz) {
        if (xyz.size() != 3)
            throw std::runtime_error("AddClumps: input_xyz must have 3 elements");
        return AddClumps(type, make_float3(xyz[0], xyz[1], xyz[2]));
    }
    static std::vector<float3> to_float3(const std::vector<std::vector<float>>& v, const char* who) {
        std::vector<float3> out(v.size());
        for (size_t i = 0; i < v.size(); i++) {
            if (v[i].size() != 3)
                throw std::runtime_error(std::string(who) + ": every position must have 3 elements");
            out[i] = make_float3(v[i][0], v[i][1], v[i][2]);
        }
        return out;
    }
    std::shared_ptr<DEMClumpBatch> AddClumps(const std::shared_ptr<DEMClumpTemplate>& type, const std::vector<std::vector<float>>& xyz) {
        return AddClumps(type, to_float3(xyz, "AddClumps"));
    }
    std::shared_ptr<DEMClumpBatch> AddClumps(const std::vector<std::shared_ptr<DEMClumpTemplate>>& types,
                                             const std::vector<std::vector<float>>& xyz) {
        return AddClumps(types, to_float3(xyz, "AddClumps"));
    }
    std::shared_ptr<DEMClumpBatch> AddClumps(DEMClumpBatch& input_batch) {
        auto b = std::make_shared<DEMClumpBatch>(input_batch);
        m_batches.push_back(b);
        return b;
    }
    std::shared_ptr<DEMExternObj> AddExternalObject() {
        m_ext.push_back(std::make_shared<DEMExternObj>());
        return m_ext.back();
    }
    std::shared_ptr<DEMExternObj> AddBCPlane(float3 pos, float3 normal, const std::shared_ptr<DEMMaterial>& mat) {
        auto o = AddExternalObject();
        o->AddPlane(pos, normal, mat);
        return o;
    }
    std::shared_ptr<DEMMeshConnected> AddMeshObject(const std::vector<float3>& vertices, const std::vector<std::array<int, 3>>& faces,
                                                    const std::shared_ptr<DEMMaterial>& mat) {
        auto m = std::make_shared<DEMMeshConnected>();
        m->vertices = vertices, m->faces = faces, m->mat = mat;
        m_meshes.push_back(m);
        return m;
    }

    // ---- force model (APIPublic.cpp:906-933)
    std::shared_ptr<DEMForceModel> UseFrictionalHertzianModel() {
        m_force_model->SetForceModelType(FORCE_MODEL::HERTZIAN);
        return m_force_model;
    }
    std::shared_ptr<DEMForceModel> UseFrictionlessHertzianModel() {
        m_force_model->SetForceModelType(FORCE_MODEL::HERTZIAN_FRICTIONLESS);
        return m_force_model;
    }
    std::shared_ptr<DEMForceModel> DefineContactForceModel(const std::string& model) {
        m_force_model = std::make_shared<DEMForceModel>();
        m_force_model->DefineCustomModel(model);
        return m_force_model;
    }
    std::shared_ptr<DEMForceModel> ReadContactForceModel(const std::string& file) {
        m_force_model = std::make_shared<DEMForceModel>();
        if (m_force_model->ReadCustomModelFile(file))
            throw std::runtime_error("The force model file " + file + " is not found.");
        return m_force_model;
    }

    // ---- knobs
    void SetInitTimeStep(double h) { m_h = (float)h; }
    void SetGravitationalAcceleration(float3 g) { m_G = g; }
    void SetCDUpdateFreq(int k) { m_cd_freq = k < 0 ? 0 : (unsigned)k; }
    /// initial bin size (API.h:140-153, 1403-1412): explicit, a multiple of the smallest radius, or -- the default -- whatever gives
    /// about the target number of bins (1e6), found by the loop of APIPrivate.cpp:525-541 starting from the multiple (8)
    void SetInitBinSize(double s) { m_bin_size = s, m_bin_num_target = 0; }
    void SetInitBinSizeAsMultipleOfSmallestSphere(float m) { m_bin_multiple = m, m_bin_size = -1, m_bin_num_target = 0; }
    void SetInitBinNumTarget(size_t num) { m_bin_num_target = num, m_bin_size = -1; }
    void SetExpandSafetyMultiplier(float m) { m_safety_multi = m; }
    void SetExpandSafetyAdder(float a) { m_safety_adder = a; }
    void SetMaxVelocity(float v) { m_max_vel = v; }
    void SetMaxSphereInBin(unsigned int n) { m_max_sph_in_bin = n; }  // errOutBinSphNum (API.h:212)
    void SetMaxTriangleInBin(unsigned int) {}  // triangles are swept per (bin, triangle) incidence here: no per-bin staging limit
    void SetErrorOutAvgContacts(float) {}      // the contact arena grows on demand: nothing to cap
    void SetSortContactPairs(bool) {}          // the list is always in (A, class, B) order
    void DisableFamilyOutput(unsigned int fam) { m_no_output_families.insert(fam & 255u); }
    void EnableFamilyOutput(unsigned int fam) { m_no_output_families.erase(fam & 255u); }
    void SetErrorOutVelocity(float v) { m_err_vel = v; }
    void SetIntegrator(TIME_INTEGRATOR i) { m_integrator = i; }
    void SetIntegrator(const std::string& intg) {  // API.h:124, APIPublic.cpp SetIntegrator(string)
        const std::string u = upper(intg);
        if (u == "FORWARD_EULER")
            m_integrator = TIME_INTEGRATOR::FORWARD_EULER;
        else if (u == "CENTERED_DIFFERENCE")
            m_integrator = TIME_INTEGRATOR::CENTERED_DIFFERENCE;
        else if (u == "EXTENDED_TAYLOR")
            m_integrator = TIME_INTEGRATOR::EXTENDED_TAYLOR;
        else
            throw std::runtime_error("Integration type " + intg + " is unknown. Please select another via SetIntegrator.");
    }
    void SetFamilyFixed(unsigned int f) { m_family_flags[f & 255] |= DEME_FAMILY_FIXED; }

    // ---- family motion prescriptions and on-the-fly family changes (API.h:720-838, 1024-1028; APIPublic.cpp:1013-1330)
    void SetFamilyPrescribedLinVel(unsigned int ID, const std::string& velX, const std::string& velY, const std::string& velZ,
                                   bool dictate = true, const std::string& pre = "none") {
        Presc& q = new_presc(ID);
        q.flag[0] = q.flag[1] = q.flag[2] = q.flag[3] = q.flag[4] = q.flag[5] = dictate;  // lin vel prescription also fixes rotation
        q.s["linVelX"] = velX, q.s["linVelY"] = velY, q.s["linVelZ"] = velZ, q.s["linVelPre"] = pre;
        if (velX != "none") q.flag[0] = true;
        if (velY != "none") q.flag[1] = true;
        if (velZ != "none") q.flag[2] = true;
    }
    void SetFamilyPrescribedAngVel(unsigned int ID, const std::string& velX, const std::string& velY, const std::string& velZ,
                                   bool dictate = true, const std::string& pre = "none") {
        Presc& q = new_presc(ID);
        q.flag[0] = q.flag[1] = q.flag[2] = q.flag[3] = q.flag[4] = q.flag[5] = dictate;
        q.s["rotVelX"] = velX, q.s["rotVelY"] = velY, q.s["rotVelZ"] = velZ, q.s["rotVelPre"] = pre;
        if (velX != "none") q.flag[3] = true;
        if (velY != "none") q.flag[4] = true;
        if (velZ != "none") q.flag[5] = true;
    }
    void SetFamilyPrescribedPosition(unsigned int ID, const std::string& X, const std::string& Y, const std::string& Z,
                                     bool dictate = true, const std::string& pre = "none") {
        Presc& q = new_presc(ID);
        q.flag[6] = q.flag[7] = q.flag[8] = q.flag[9] = dictate;
        q.s["linPosX"] = X, q.s["linPosY"] = Y, q.s["linPosZ"] = Z, q.s["linPosPre"] = pre;
        if (X != "none") q.flag[6] = true;
        if (Y != "none") q.flag[7] = true;
        if (Z != "none") q.flag[8] = true;
    }
    void SetFamilyPrescribedQuaternion(unsigned int ID, const std::string& q_formula, bool dictate = true) {
        Presc& q = new_presc(ID);
        q.flag[6] = q.flag[7] = q.flag[8] = q.flag[9] = dictate;
        q.s["oriQ"] = q_formula;
        if (q_formula != "none") q.flag[9] = true;
    }
    // "keep as is": the components are marked prescribed with no expression, so the contact forces stop changing them
    // (API.h:712-778, APIPublic.cpp:1056-1362)
    void SetFamilyPrescribedLinVel(unsigned int ID) { keep_as_is(ID, {0, 1, 2}); }
    void SetFamilyPrescribedLinVelX(unsigned int ID) { keep_as_is(ID, {0}); }
    void SetFamilyPrescribedLinVelY(unsigned int ID) { keep_as_is(ID, {1}); }
    void SetFamilyPrescribedLinVelZ(unsigned int ID) { keep_as_is(ID, {2}); }
    void SetFamilyPrescribedAngVel(unsigned int ID) { keep_as_is(ID, {3, 4, 5}); }
    void SetFamilyPrescribedAngVelX(unsigned int ID) { keep_as_is(ID, {3}); }
    void SetFamilyPrescribedAngVelY(unsigned int ID) { keep_as_is(ID, {4}); }
    void SetFamilyPrescribedAngVelZ(unsigned int ID) { keep_as_is(ID, {5}); }
    void SetFamilyPrescribedPosition(unsigned int ID) { keep_as_is(ID, {6, 7, 8}); }
    void SetFamilyPrescribedPositionX(unsigned int ID) { keep_as_is(ID, {6}); }
    void SetFamilyPrescribedPositionY(unsigned int ID) { keep_as_is(ID, {7}); }
    void SetFamilyPrescribedPositionZ(unsigned int ID) { keep_as_is(ID, {8}); }
    void SetFamilyPrescribedQuaternion(unsigned int ID) { keep_as_is(ID, {9}); }
    void AddFamilyPrescribedAcc(unsigned int ID, const std::string& X, const std::string& Y, const std::string& Z,
                                const std::string& pre = "none") {
        Presc& q = new_presc(ID);
        q.s["accX"] = X, q.s["accY"] = Y, q.s["accZ"] = Z, q.s["accPre"] = pre;
    }
    void AddFamilyPrescribedAngAcc(unsigned int ID, const std::string& X, const std::string& Y, const std::string& Z,
                                   const std::string& pre = "none") {
        Presc& q = new_presc(ID);
        q.s["angAccX"] = X, q.s["angAccY"] = Y, q.s["angAccZ"] = Z, q.s["angAccPre"] = pre;
    }
    void ChangeFamilyWhen(unsigned int ID_from, unsigned int ID_to, const std::string& condition) {
        m_family_rules.push_back({ID_from, ID_to, condition});
    }
    void ChangeFamily(unsigned int ID_from, unsigned int ID_to) {
        each_ctx([&](deme_ctx* c) { return deme_change_family(c, ID_from, ID_to); });
        m_state_fresh = false;
    }
    // ---- wildcard values (API.h:852-868, 936-1014).  Owner and geometry wildcards belong to a user force model
    // (SetPerOwnerWildcards / SetPerGeometryWildcards); contact wildcards to whichever model runs.  Post-Initialize calls.
    void SetOwnerWildcardValue(bodyID_t ownerID, const std::string& name, const std::vector<float>& vals) {
        edit_wildcard(0, m_n_owners, m_force_model->owner_wildcards, name, [&](std::vector<float>& a) {
            for (size_t k = 0; k < vals.size() && ownerID + k < a.size(); k++)
                a[ownerID + k] = vals[k];
        });
    }
    void SetOwnerWildcardValue(bodyID_t ownerID, const std::string& name, float val, size_t n = 1) {
        SetOwnerWildcardValue(ownerID, name, std::vector<float>(n, val));
    }
    void SetFamilyOwnerWildcardValue(unsigned int N, const std::string& name, const std::vector<float>& vals) {
        const std::vector<uint8_t> fam = owner_families();
        edit_wildcard(0, m_n_owners, m_force_model->owner_wildcards, name, [&](std::vector<float>& a) {
            size_t k = 0;  // one value: every member gets it; several: handed out in owner order
            for (size_t o = 0; o < a.size(); o++)
                if (fam[o] == N) {
                    a[o] = vals[std::min(k, vals.size() - 1)];
                    k++;
                }
        });
    }
    void SetFamilyOwnerWildcardValue(unsigned int N, const std::string& name, float val) {
        SetFamilyOwnerWildcardValue(N, name, std::vector<float>(1, val));
    }
    std::vector<float> GetAllOwnerWildcardValue(const std::string& name) {
        return get_wildcard(0, m_n_owners, m_force_model->owner_wildcards, name);
    }
    std::vector<float> GetOwnerWildcardValue(bodyID_t ownerID, const std::string& name, bodyID_t n = 1) {
        const std::vector<float> a = GetAllOwnerWildcardValue(name);
        return std::vector<float>(a.begin() + ownerID, a.begin() + std::min<size_t>(a.size(), (size_t)ownerID + n));
    }
    std::vector<float> GetFamilyOwnerWildcardValue(unsigned int N, const std::string& name) {
        const std::vector<float> a = GetAllOwnerWildcardValue(name);
        const std::vector<uint8_t> fam = owner_families();
        std::vector<float> out;
        for (size_t o = 0; o < a.size(); o++)
            if (fam[o] == N)
                out.push_back(a[o]);
        return out;
    }
    void SetSphereWildcardValue(bodyID_t geoID, const std::string& name, const std::vector<float>& vals) {
        set_geo(1, m_keep.sphOwner.size(), geoID, name, vals);
    }
    void SetTriWildcardValue(bodyID_t geoID, const std::string& name, const std::vector<float>& vals) {
        set_geo(2, m_keep.triOwner.size(), geoID, name, vals);
    }
    void SetAnalWildcardValue(bodyID_t geoID, const std::string& name, const std::vector<float>& vals) {
        set_geo(3, m_keep.objOwner.size(), geoID, name, vals);
    }
    std::vector<float> GetSphereWildcardValue(bodyID_t geoID, const std::string& name, size_t n) {
        return get_geo(1, m_keep.sphOwner.size(), geoID, name, n);
    }
    std::vector<float> GetTriWildcardValue(bodyID_t geoID, const std::string& name, size_t n) {
        return get_geo(2, m_keep.triOwner.size(), geoID, name, n);
    }
    std::vector<float> GetAnalWildcardValue(bodyID_t geoID, const std::string& name, size_t n) {
        return get_geo(3, m_keep.objOwner.size(), geoID, name, n);
    }
    /// every contact of the current list (DEMSolver::SetContactWildcardValue, API.h:868)
    void SetContactWildcardValue(const std::string& name, float val) { set_contact_wc(0, 0, 0, name, val); }
    void SetFamilyContactWildcardValueEither(unsigned int N, const std::string& name, float val) { set_contact_wc(1, N, 0, name, val); }
    void SetFamilyContactWildcardValueBoth(unsigned int N, const std::string& name, float val) { set_contact_wc(2, N, 0, name, val); }
    void SetFamilyContactWildcardValue(unsigned int N1, unsigned int N2, const std::string& name, float val) {
        set_contact_wc(3, N1, N2, name, val);
    }
    // Persistent contacts (DEM/API.h:874-905): contacts of the current list that qualify stay in the list at every later
    // contact detection.  Like the reference these are post-Initialize calls.
    void MarkFamilyPersistentContactEither(unsigned int N) { each_ctx([&](deme_ctx* c) { return deme_mark_persistent_contacts(c, 1, N, 0, 1); }); }
    void MarkFamilyPersistentContactBoth(unsigned int N) { each_ctx([&](deme_ctx* c) { return deme_mark_persistent_contacts(c, 2, N, 0, 1); }); }
    void MarkFamilyPersistentContact(unsigned int N1, unsigned int N2) { each_ctx([&](deme_ctx* c) { return deme_mark_persistent_contacts(c, 3, N1, N2, 1); }); }
    void MarkPersistentContact() { each_ctx([&](deme_ctx* c) { return deme_mark_persistent_contacts(c, 0, 0, 0, 1); }); }
    void RemoveFamilyPersistentContactEither(unsigned int N) { each_ctx([&](deme_ctx* c) { return deme_mark_persistent_contacts(c, 1, N, 0, 0); }); }
    void RemoveFamilyPersistentContactBoth(unsigned int N) { each_ctx([&](deme_ctx* c) { return deme_mark_persistent_contacts(c, 2, N, 0, 0); }); }
    void RemoveFamilyPersistentContact(unsigned int N1, unsigned int N2) { each_ctx([&](deme_ctx* c) { return deme_mark_persistent_contacts(c, 3, N1, N2, 0); }); }
    void RemovePersistentContact() { each_ctx([&](deme_ctx* c) { return deme_mark_persistent_contacts(c, 0, 0, 0, 0); }); }
    void DisableContactBetweenFamilies(unsigned int a, unsigned int b) {
        if (a > b)
            std::swap(a, b);
        m_family_masks[(1 + b) * b / 2 + a] = 1;  // locateMaskPair, DEMHelperKernels.cuh:57-62
    }
    void EnableContactBetweenFamilies(unsigned int a, unsigned int b) {
        if (a > b)
            std::swap(a, b);
        m_family_masks[(1 + b) * b / 2 + a] = 0;
    }
    void SetFamilyExtraMargin(unsigned int f, float m) { m_family_extra[f & 255] = m; }
    // ---- adaptive controllers (API.h:253-309) on device timers: deme_set_adaptive.  The reference switches both on by
    // default; here they are opt-in so that runs are reproducible step for step unless asked otherwise.
    void UseAdaptiveBinSize(bool use = true) { m_adaptive.autoBinSize = use, push_adaptive(); }
    void DisableAdaptiveBinSize() { UseAdaptiveBinSize(false); }
    void UseAdaptiveUpdateFreq(bool use = true) { m_adaptive.autoUpdateFreq = use, push_adaptive(); }
    void DisableAdaptiveUpdateFreq() { UseAdaptiveUpdateFreq(false); }
    void SetAdaptiveBinSizeDelaySteps(unsigned int n) { m_adaptive.binObserveSteps = n >= 1 ? n : 1, push_adaptive(); }
    void SetAdaptiveBinSizeMaxRate(float rate) { m_adaptive.binMaxRate = rate > 0 ? rate : 0, push_adaptive(); }
    void SetAdaptiveBinSizeAcc(float acc) { m_adaptive.binAcc = std::min(1.f, std::max(0.01f, acc)), push_adaptive(); }
    void SetAdaptiveBinSizeUpperProactivity(float r) {  // API.h:282-284 -> APIPrivate.cpp:1106-1107
        m_adaptive.binUpperSafety = 0.01f + (1.f - std::min(1.f, std::max(0.f, r))) * 0.98f, push_adaptive();
    }
    void SetAdaptiveBinSizeLowerProactivity(float r) {
        m_adaptive.binLowerSafety = 0.01f + (1.f - std::min(1.f, std::max(0.f, r))) * 0.98f, push_adaptive();
    }
    void SetCDMaxUpdateFreq(unsigned int max_freq) { m_adaptive.maxUpdateFreq = max_freq, push_adaptive(); }
    double GetBinSize() const {
        double b = 0;
        deme_get_adaptive_state(diag_ctx(), &b, nullptr, nullptr, nullptr);
        return b;
    }
    unsigned int GetUpdateFreq() const {
        uint32_t k = 0;
        deme_get_adaptive_state(diag_ctx(), nullptr, &k, nullptr, nullptr);
        return k;
    }
    void SetNoForceRecord(bool flag = true) {  // per-contact force records are only kept when the contact output needs them
        if (flag)
            m_cnt_out_content &= ~(unsigned)(FORCE | CNT_POINT | NORMAL | TORQUE);
    }
    void SetCollectAccRightAfterForceCalc(bool = true) {}

    // ---- run
    void Initialize(bool dry_run_before_simulation) { (void)dry_run_before_simulation, Initialize(); }
    void Initialize() {
        for (auto& prop : m_force_model->must_have_props)
            for (auto& m : m_materials)
                if (!m->mat_prop.count(prop))
                    throw std::runtime_error("A material is loaded without the property " + prop + " that the force model requires "
                                             "(SetMustHaveMatProp)");
        initialize_impl();
        m_initialized = true;
        push_adaptive();
    }
    void DoDynamics(double t) { step((uint32_t)std::llround(t / (double)m_h)); }
    void DoDynamicsThenSync(double t) {
        DoDynamics(t);
        if (m_multi)
            mcheck(deme_multi_sync(m_multi));
        else
            check(deme_sync(m_ctx));
    }
    void DoStepDynamics(unsigned int n = 1) { step(n); }
    double GetSimTime() const { return m_time; }

    // ---- queries (subset of DEMTracker / GetOwner* getters)
    size_t GetNumClumps() const { return m_n_clumps; }
    size_t GetNumContacts() { return n_contacts(); }
    float3 GetOwnerPosition(unsigned int owner) {
        refresh_state();
        return m_pos.at(owner);
    }
    float3 GetOwnerVelocity(unsigned int owner) {
        refresh_state();
        return {m_st_v[0].at(owner), m_st_v[1].at(owner), m_st_v[2].at(owner)};
    }
    // DEMSolver::Get/SetOwner* (API.h:515-586): whole-state round trips through the ABI -- scripting calls, not for inner loops
    float3 GetOwnerAngVel(unsigned int owner) { return owner_column3(owner, 2); }
    float3 GetOwnerAcc(unsigned int owner) { return owner_column3(owner, 3); }
    float3 GetOwnerAngAcc(unsigned int owner) { return owner_column3(owner, 4); }
    float4 GetOwnerOriQ(unsigned int owner) {
        const Snapshot sn = snapshot(false);
        return sn.q.at(owner);
    }
    unsigned int GetOwnerFamily(unsigned int owner) { return owner_families().at(owner); }
    float GetOwnerMass(unsigned int owner) const { return m_keep.mass.at(m_keep.inert.at(owner)); }
    float3 GetOwnerMOI(unsigned int owner) const {
        const uint16_t k = m_keep.inert.at(owner);
        return {m_keep.moix.at(k), m_keep.moiy.at(k), m_keep.moiz.at(k)};
    }
    void SetOwnerPosition(unsigned int owner, float3 pos) { set_owner(owner, &pos, nullptr, nullptr, nullptr); }
    void SetOwnerVelocity(unsigned int owner, float3 vel) { set_owner(owner, nullptr, &vel, nullptr, nullptr); }
    void SetOwnerAngVel(unsigned int owner, float3 w) { set_owner(owner, nullptr, nullptr, &w, nullptr); }
    void SetOwnerOriQ(unsigned int owner, float4 q) { set_owner(owner, nullptr, nullptr, nullptr, &q); }
    void SetOwnerFamily(unsigned int owner, unsigned int fam, size_t n = 1) {
        std::vector<uint8_t> f = owner_families();
        for (size_t k = 0; k < n && owner + k < f.size(); k++)
            f[owner + k] = (uint8_t)fam;
        DemeOwnerState st{};
        st.familyID = f.data();
        ul_state(&st);
        m_state_fresh = false;
    }
    /// every clump whose CoM lies in the box becomes family `fam_num`; returns how many did (API.h:699-709)
    size_t ChangeClumpFamily(unsigned int fam_num, const std::pair<double, double>& X = {-1e30, 1e30},
                             const std::pair<double, double>& Y = {-1e30, 1e30}, const std::pair<double, double>& Z = {-1e30, 1e30}) {
        refresh_state();
        std::vector<uint8_t> f = owner_families();
        size_t changed = 0;
        for (size_t i = 0; i < m_n_clumps; i++) {
            const float3 c = m_pos[i];
            if (c.x >= X.first && c.x <= X.second && c.y >= Y.first && c.y <= Y.second && c.z >= Z.first && c.z <= Z.second) {
                f[i] = (uint8_t)fam_num;
                changed++;
            }
        }
        DemeOwnerState st{};
        st.familyID = f.data();
        ul_state(&st);
        return changed;
    }
    float GetMaxOwnerSpeed() {
        refresh_state();
        float m = 0;
        for (size_t i = 0; i < m_n_clumps; i++)
            m = std::max(m, std::sqrt(m_st_v[0][i] * m_st_v[0][i] + m_st_v[1][i] * m_st_v[1][i] + m_st_v[2][i] * m_st_v[2][i]));
        return m;
    }
    deme_ctx* GetContext() { return m_ctx; }

    // ---- more of the reference's surface (API.h): mesh loading from OBJ, mesh output, run-time updates, statistics
    std::shared_ptr<DEMMeshConnected> AddWavefrontMeshObject(const std::string& filename, const std::shared_ptr<DEMMaterial>& mat,
                                                             bool load_normals = true, bool load_uv = false) {
        (void)load_normals, (void)load_uv;
        auto m = std::make_shared<DEMMeshConnected>();
        if (!m->LoadWavefrontMesh(filename))
            throw std::runtime_error("Failed to load in mesh file " + filename + ".");
        m->mat = mat;
        m_meshes.push_back(m);
        return m;
    }
    using MESH_FORMAT = deme::MESH_FORMAT;
    void SetMeshOutputFormat(MESH_FORMAT f) {
        if (f != MESH_FORMAT::VTK)
            throw std::runtime_error("only MESH_FORMAT::VTK is implemented");
    }
    void SetMeshOutputFormat(const std::string& format) {  // API.h:1354
        const std::string u = upper(format);
        if (u == "VTK")
            SetMeshOutputFormat(MESH_FORMAT::VTK);
        else if (u == "OBJ")
            SetMeshOutputFormat(MESH_FORMAT::OBJ);
        else
            throw std::runtime_error("Instruction " + format + " is unknown in SetMeshOutputFormat call.");
    }
    /// writeMeshesAsVtk (dT.cpp:1850-1935): all meshes in one legacy-VTK unstructured grid, nodes in the global frame
    void WriteMeshFile(const std::string& outfilename) {
        const Snapshot sn = snapshot(false);
        std::ostringstream o;
        o << "# vtk DataFile Version 2.0\nVTK from DEM simulation\nASCII\n\n\nDATASET UNSTRUCTURED_GRID\n";
        size_t total_v = 0, total_f = 0;
        std::vector<size_t> voff(m_meshes.size() + 1, 0);
        for (size_t i = 0; i < m_meshes.size(); i++) {
            voff[i + 1] = voff[i] + m_meshes[i]->vertices.size();
            total_v += m_meshes[i]->vertices.size(), total_f += m_meshes[i]->faces.size();
        }
        o << "POINTS " << total_v << " float" << std::endl;
        for (size_t i = 0; i < m_meshes.size(); i++) {
            const size_t owner = m_n_owners - m_meshes.size() + i;
            for (float3 v : m_meshes[i]->vertices) {  // applyFrameTransformLocalToGlobal: rotate, then translate
                rotate(v, sn.q[owner]);
                v = v + sn.com[owner];
                o << v.x << " " << v.y << " " << v.z << std::endl;
            }
        }
        o << "\n\nCELLS " << total_f << " " << 4 * total_f << std::endl;
        for (size_t i = 0; i < m_meshes.size(); i++)
            for (auto& f : m_meshes[i]->faces)
                o << "3 " << (size_t)f[0] + voff[i] << " " << (size_t)f[1] + voff[i] << " " << (size_t)f[2] + voff[i] << std::endl;
        o << "\n\nCELL_TYPES " << total_f << std::endl;
        for (size_t j = 0; j < total_f; j++)
            o << "5 " << std::endl;
        flush(outfilename, o);
    }
    /// UpdateClumps (API.h:1267): clumps added with AddClumps after Initialize() join the running simulation; old owners keep
    /// their state and the contact list keeps its history (new clumps are appended, so sphere ids are stable)
    void UpdateClumps() {
        const size_t oldClumps = m_n_clumps, oldOwners = m_n_owners;
        // state of the old owners
        std::vector<uint64_t> vid(oldOwners);
        std::vector<uint16_t> lx(oldOwners), ly(oldOwners), lz(oldOwners);
        std::vector<float> f[10];
        for (auto& v : f)
            v.resize(oldOwners);
        std::vector<uint8_t> fam(oldOwners);
        DemeOwnerState st{};
        st.voxelID = vid.data(), st.locX = lx.data(), st.locY = ly.data(), st.locZ = lz.data();
        st.oriQw = f[0].data(), st.oriQx = f[1].data(), st.oriQy = f[2].data(), st.oriQz = f[3].data();
        st.vX = f[4].data(), st.vY = f[5].data(), st.vZ = f[6].data();
        st.omgBarX = f[7].data(), st.omgBarY = f[8].data(), st.omgBarZ = f[9].data();
        st.familyID = fam.data();
        dl_state(&st);
        // contact list + wildcards
        const size_t nc = n_contacts();
        const uint32_t nW = m_p.nContactWildcards;
        std::vector<uint32_t> a(nc), b(nc), map(nc);
        std::vector<uint8_t> ty(nc);
        dl_contacts(a.data(), b.data(), ty.data(), map.data(), nc);
        std::vector<float> W(nc * nW), col(nc);
        for (uint32_t w = 0; w < nW; w++) {
            dl_contact_wc(w, col.data(), nc);
            for (size_t i = 0; i < nc; i++)
                W[i * nW + w] = col[i];
        }
        const double t = m_time;
        const SavedWildcards saved = save_user_wildcards();
        // (a context keeps its persistent marks across a re-upload; a decomposed run gets new contexts: the marks go with the list)
        size_t nP = 0;
        std::vector<uint32_t> pa, pb;
        std::vector<uint8_t> pt;
        if (m_multi) {
            mc_num_persistent_contacts(&nP);
            pa.resize(nP), pb.resize(nP), pt.resize(nP);
            if (nP)
                mc_download_persistent_contacts(pa.data(), pb.data(), pt.data(), nP);
        }
        ReuploadGuard reuploadGuard{this};  // (st and the mapping live on this frame: whatever leaves it, an exception included, clears the members)
        m_reupload_state = &st, m_reupload_n = oldOwners;
        m_reupload_dst = [=](size_t o, size_t newClumps) { return o < oldClumps ? o : newClumps + (o - oldClumps); };
        initialize_impl();  // rebuilds and uploads the scene with all batches
        if (m_n_clumps < oldClumps || m_n_owners - m_n_clumps != oldOwners - oldClumps)
            throw std::runtime_error("UpdateClumps can only append clumps");
        restore_user_wildcards(saved, [&](size_t o) { return o < oldClumps ? o : m_n_clumps + (o - oldClumps); },
                               [](size_t i) { return i; });  // new spheres are appended
        // put the old owners' state back: old clumps keep their slots, the other owners moved behind the new clumps
        const size_t n = m_n_owners;
        std::vector<uint64_t> vid2(n);
        std::vector<uint16_t> lx2(n), ly2(n), lz2(n);
        std::vector<float> g[10];
        for (auto& v : g)
            v.resize(n);
        std::vector<uint8_t> fam2(n);
        DemeOwnerState s2{};
        s2.voxelID = vid2.data(), s2.locX = lx2.data(), s2.locY = ly2.data(), s2.locZ = lz2.data();
        s2.oriQw = g[0].data(), s2.oriQx = g[1].data(), s2.oriQy = g[2].data(), s2.oriQz = g[3].data();
        s2.vX = g[4].data(), s2.vY = g[5].data(), s2.vZ = g[6].data();
        s2.omgBarX = g[7].data(), s2.omgBarY = g[8].data(), s2.omgBarZ = g[9].data();
        s2.familyID = fam2.data();
        dl_state(&s2);
        for (size_t o = 0; o < oldOwners; o++) {
            const size_t dst = o < oldClumps ? o : m_n_clumps + (o - oldClumps);
            vid2[dst] = vid[o], lx2[dst] = lx[o], ly2[dst] = ly[o], lz2[dst] = lz[o], fam2[dst] = fam[o];
            for (int k = 0; k < 10; k++)
                g[k][dst] = f[k][o];
        }
        ul_state(&s2);
        m_time = t;
        m_p.timeElapsed = t;
        each_ctx([&](deme_ctx* c) { return deme_set_params(c, &m_p); });
        if (nc)
            mc_seed_contacts(a.data(), b.data(), ty.data(), nW ? W.data() : nullptr, nc);
        if (nP)
            mc_upload_persistent_contacts(pa.data(), pb.data(), pt.data(), nP);
        m_state_fresh = false;
        m_steps_since_replan = 0;
    }
    /// A decomposed run cut anew by where the clumps are now: fresh slabs in the engine's own order (the migration appends arrivals
    /// to a slab, so its tiles loosen over a long run of mixing), state, contact history, marks and wildcard arrays carried by global
    /// id.  No reference equivalent.  SetSlabReplanInterval(n): every n steps inside DoDynamics (0, the default: never); a replan
    /// opens the slabs' contexts and communicators again -- seconds at 10^6 clumps: an interval of 10^5 steps or more.
    void ReplanSlabs() {
        if (m_multi)
            UpdateClumps();  // (nothing appended: the re-upload of the same batches with the owners' current state)
    }
    void SetSlabReplanInterval(unsigned int steps) { m_replan_every = steps; }
    /// Restore spatial order in a running simulation (no reference equivalent: its owner ids never change).  The clumps of every
    /// batch are renumbered along a Z-order curve of their current positions; state, contact list, contact history and
    /// persistent marks follow.  Mixing destroys the locality the load order had and the engine's gathers slow down with it
    /// (28 % for a random order).  Host-side, about a second per million clumps.  Returns the new owner id
    /// of every old owner; trackers of clump batches keep following their batch, not individual clumps.
    std::vector<bodyID_t> ResortClumps(float cell = 0.f) {
        refresh_state();
        const size_t nC = m_n_clumps, nO = m_n_owners;
        // old state, contacts, history, marks
        std::vector<uint64_t> vid(nO);
        std::vector<uint16_t> lx(nO), ly(nO), lz(nO);
        std::vector<float> f[10];
        for (auto& v : f)
            v.resize(nO);
        std::vector<uint8_t> fam(nO);
        DemeOwnerState st{};
        st.voxelID = vid.data(), st.locX = lx.data(), st.locY = ly.data(), st.locZ = lz.data();
        st.oriQw = f[0].data(), st.oriQx = f[1].data(), st.oriQy = f[2].data(), st.oriQz = f[3].data();
        st.vX = f[4].data(), st.vY = f[5].data(), st.vZ = f[6].data();
        st.omgBarX = f[7].data(), st.omgBarY = f[8].data(), st.omgBarZ = f[9].data();
        st.familyID = fam.data();
        dl_state(&st);
        const size_t nc = n_contacts();
        const uint32_t nW = m_p.nContactWildcards;
        std::vector<uint32_t> a(nc), b(nc), map(nc);
        std::vector<uint8_t> ty(nc);
        dl_contacts(a.data(), b.data(), ty.data(), map.data(), nc);
        std::vector<float> W(nc * nW), col(nc);
        for (uint32_t w = 0; w < nW; w++) {
            dl_contact_wc(w, col.data(), nc);
            for (size_t i = 0; i < nc; i++)
                W[i * nW + w] = col[i];
        }
        size_t nP = 0;
        mc_num_persistent_contacts(&nP);
        std::vector<uint32_t> pa(nP), pb(nP);
        std::vector<uint8_t> pt(nP);
        if (nP)
            mc_download_persistent_contacts(pa.data(), pb.data(), pt.data(), nP);
        // Z-order of the current positions, batch by batch (a batch keeps its id range)
        float rmax = 0.f;
        for (auto& t : m_templates)
            for (size_t k = 0; k < t->radii.size(); k++)
                rmax = std::max(rmax, t->radii[k] + std::max({std::fabs(t->relPos[k].x), std::fabs(t->relPos[k].y), std::fabs(t->relPos[k].z)}));
        if (!(cell > 0.f))
            cell = 4.f * (rmax > 0.f ? rmax : 1.f);
        float3 lo = m_pos.empty() ? make_float3(0, 0, 0) : m_pos[0];
        for (size_t i = 0; i < nC; i++)
            lo = {std::min(lo.x, m_pos[i].x), std::min(lo.y, m_pos[i].y), std::min(lo.z, m_pos[i].z)};
        auto morton = [&](float3 p) {
            const uint64_t q[3] = {(uint64_t)((p.x - lo.x) / cell), (uint64_t)((p.y - lo.y) / cell), (uint64_t)((p.z - lo.z) / cell)};
            uint64_t code = 0;
            for (int bit = 0; bit < 21; bit++)
                for (int ax = 0; ax < 3; ax++)
                    code |= ((q[ax] >> bit) & 1ull) << (3 * bit + ax);
            return code;
        };
        std::vector<size_t> old_of_new(nC);
        std::vector<bodyID_t> new_of_old(nO);
        for (size_t o = nC; o < nO; o++)
            new_of_old[o] = (bodyID_t)o;
        size_t o0 = 0;
        for (auto& bt : m_batches) {
            const size_t n = bt->nClumps;
            std::vector<uint64_t> code(n);
            std::vector<size_t> order(n);
            for (size_t i = 0; i < n; i++)
                code[i] = morton(m_pos[o0 + i]), order[i] = i;
            std::stable_sort(order.begin(), order.end(), [&](size_t x, size_t y) { return code[x] < code[y]; });
            auto permute = [&](auto& v) {
                auto old = v;
                for (size_t i = 0; i < n && i < old.size(); i++)
                    v[i] = old[order[i]];
            };
            permute(bt->types), permute(bt->xyz), permute(bt->vel), permute(bt->angVel), permute(bt->oriQ), permute(bt->families);
            bt->contact_pairs.clear(), bt->contact_wildcards.clear();  // restart data referred to the old numbering and is spent
            for (size_t i = 0; i < n; i++)
                old_of_new[o0 + i] = o0 + order[i], new_of_old[o0 + order[i]] = (bodyID_t)(o0 + i);
            o0 += n;
        }
        // sphere ids: clump-major, a clump's spheres keep their order
        const std::vector<uint32_t> oldSphOwner = m_keep.sphOwner;
        std::vector<size_t> firstOld(nC + 1, 0), cnt(nC, 0);
        for (uint32_t ow : oldSphOwner)
            cnt[ow]++;
        for (size_t o = 0; o < nC; o++)
            firstOld[o + 1] = firstOld[o] + cnt[o];
        std::vector<size_t> firstNew(nC + 1, 0);
        for (size_t j = 0; j < nC; j++)
            firstNew[j + 1] = firstNew[j] + cnt[old_of_new[j]];
        auto new_sphere = [&](uint32_t s) {
            const uint32_t ow = oldSphOwner[s];
            return (uint32_t)(firstNew[new_of_old[ow]] + (s - firstOld[ow]));
        };
        const double t = m_time;
        const SavedWildcards saved = save_user_wildcards();
        ReuploadGuard reuploadGuard{this};  // (st, new_of_old live on this frame: whatever leaves it, an exception included, clears the members)
        m_reupload_state = &st, m_reupload_n = nO;
        m_reupload_dst = [&](size_t o, size_t) { return (size_t)new_of_old[o]; };
        initialize_impl();
        restore_user_wildcards(saved, [&](size_t o) { return (size_t)new_of_old[o]; }, [&](size_t i) { return (size_t)new_sphere((uint32_t)i); });
        std::vector<uint64_t> vid2(nO);
        std::vector<uint16_t> lx2(nO), ly2(nO), lz2(nO);
        std::vector<float> g[10];
        for (auto& v : g)
            v.resize(nO);
        std::vector<uint8_t> fam2(nO);
        for (size_t o = 0; o < nO; o++) {
            const size_t dst = new_of_old[o];
            vid2[dst] = vid[o], lx2[dst] = lx[o], ly2[dst] = ly[o], lz2[dst] = lz[o], fam2[dst] = fam[o];
            for (int k = 0; k < 10; k++)
                g[k][dst] = f[k][o];
        }
        DemeOwnerState s2{};
        s2.voxelID = vid2.data(), s2.locX = lx2.data(), s2.locY = ly2.data(), s2.locZ = lz2.data();
        s2.oriQw = g[0].data(), s2.oriQx = g[1].data(), s2.oriQy = g[2].data(), s2.oriQz = g[3].data();
        s2.vX = g[4].data(), s2.vY = g[5].data(), s2.vZ = g[6].data();
        s2.omgBarX = g[7].data(), s2.omgBarY = g[8].data(), s2.omgBarZ = g[9].data();
        s2.familyID = fam2.data();
        ul_state(&s2);
        m_time = t;
        m_p.timeElapsed = t;
        each_ctx([&](deme_ctx* c) { return deme_set_params(c, &m_p); });
        const bool hertz = m_force_model->type == FORCE_MODEL::HERTZIAN;
        auto remap = [&](std::vector<uint32_t>& A, std::vector<uint32_t>& B, const std::vector<uint8_t>& T, std::vector<float>* wc) {
            for (size_t i = 0; i < A.size(); i++) {
                A[i] = new_sphere(A[i]);
                if (T[i] != DEME_SPHERE_SPHERE_CONTACT)
                    continue;
                B[i] = new_sphere(B[i]);
                if (A[i] > B[i]) {  // a sphere-sphere pair is stored smaller id first; the Hertzian history points from B to A
                    std::swap(A[i], B[i]);
                    if (wc && hertz)
                        for (uint32_t w = 0; w < 3 && w < nW; w++)
                            (*wc)[i * nW + w] = -(*wc)[i * nW + w];
                }
            }
        };
        if (nc) {
            remap(a, b, ty, &W);
            mc_seed_contacts(a.data(), b.data(), ty.data(), nW ? W.data() : nullptr, nc);
        }
        if (nP) {
            remap(pa, pb, pt, nullptr);
            mc_upload_persistent_contacts(pa.data(), pb.data(), pt.data(), nP);
        }
        m_state_fresh = false;
        return new_of_old;
    }
    /// UpdateStepSize (API.h:1274): takes effect from the next step
    void UpdateStepSize(double ts) {
        m_h = (float)ts;
        m_p.h = m_h;
        m_p.timeElapsed = m_time;
        each_ctx([&](deme_ctx* c) { return deme_set_params(c, &m_p); });
    }
    double GetTimeStepSize() const { return m_h; }
    void UpdateSimParams() {
        m_p.timeElapsed = m_time;
        each_ctx([&](deme_ctx* c) { return deme_set_params(c, &m_p); });
    }
    /// Sphere-geometry id pairs of the current contact list and their types (GetContacts / contact info getters)
    std::vector<std::pair<bodyID_t, bodyID_t>> GetContacts() {
        const size_t n = n_contacts();
        std::vector<uint32_t> a(n), b(n), map(n);
        std::vector<uint8_t> ty(n);
        dl_contacts(a.data(), b.data(), ty.data(), map.data(), n);
        std::vector<std::pair<bodyID_t, bodyID_t>> out;
        for (size_t i = 0; i < n; i++)
            if (ty[i] == 1)
                out.push_back({m_keep.sphOwner[a[i]], m_keep.sphOwner[b[i]]});
        return out;
    }
    /// Owner-id pairs of the clump--clump contacts of the list (potential contacts, like the reference's GetClumpContacts,
    /// API.h:500-528), sorted by A's owner; optionally only pairs whose two families are both in `family_to_include`
    std::vector<std::pair<bodyID_t, bodyID_t>> GetClumpContacts() { return GetContacts(); }
    std::vector<std::pair<bodyID_t, bodyID_t>> GetClumpContacts(const std::set<unsigned int>& family_to_include) {
        const std::vector<uint8_t> fam = owner_families();
        std::vector<std::pair<bodyID_t, bodyID_t>> out;
        for (auto& pr : GetContacts())
            if (family_to_include.count(fam[pr.first]) && family_to_include.count(fam[pr.second]))
                out.push_back(pr);
        return out;
    }
    std::vector<std::pair<bodyID_t, bodyID_t>> GetClumpContacts(std::vector<std::pair<unsigned int, unsigned int>>& family_pair) {
        const std::vector<uint8_t> fam = owner_families();
        std::vector<std::pair<bodyID_t, bodyID_t>> out = GetContacts();
        family_pair.clear();
        for (auto& pr : out)
            family_pair.push_back({fam[pr.first], fam[pr.second]});
        return out;
    }
    /// getContactForcesConcerningOwners (algorithms/DEMDynamicMisc.cu:14-100): every contact of the list with an owner in
    /// `owners` on either side and a non-negligible force; the force as the tracked owner feels it (A's side first), the contact
    /// point in global coordinates, optionally the contact torque in the owner's local or the global frame.  Needs the
    /// per-contact records (on unless SetNoForceRecord).  Returns the number of pairs.
    size_t GetOwnerContactForces(const std::vector<bodyID_t>& owners, std::vector<float3>& points, std::vector<float3>& forces,
                                 std::vector<float3>* torques = nullptr, bool torque_in_local = false) {
        const Snapshot sn = snapshot(true);
        const size_t nc = sn.idA.size();
        std::vector<float> cpB(3 * nc);
        {
            std::vector<float> f(3 * nc), t(3 * nc), a(3 * nc);
            dl_contact_records(f.data(), t.data(), a.data(), cpB.data(), nc);
        }
        std::vector<bodyID_t> sorted = owners;
        std::sort(sorted.begin(), sorted.end());
        points.clear(), forces.clear();
        if (torques)
            torques->clear();
        for (size_t i = 0; i < nc; i++) {
            const uint8_t ty = sn.type[i];
            const bodyID_t oA = m_keep.sphOwner[sn.idA[i]];
            const bodyID_t oB = ty == 1 ? m_keep.sphOwner[sn.idB[i]] : ty == 2 ? m_keep.triOwner[sn.idB[i]] : m_keep.objOwner[sn.idB[i]];
            bool isA;
            if (std::binary_search(sorted.begin(), sorted.end(), oA))
                isA = true;
            else if (std::binary_search(sorted.begin(), sorted.end(), oB))
                isA = false;
            else
                continue;
            float3 force = {sn.F[3 * i], sn.F[3 * i + 1], sn.F[3 * i + 2]}, torque = {sn.T[3 * i], sn.T[3 * i + 1], sn.T[3 * i + 2]};
            auto len = [](float3 v) { return std::sqrt(v.x * v.x + v.y * v.y + v.z * v.z); };
            if ((torques ? len(force) + len(torque) : len(force)) < DEME_TINY_FLOAT_HOST)
                continue;
            float3 pnt = isA ? make_float3(sn.cpA[3 * i], sn.cpA[3 * i + 1], sn.cpA[3 * i + 2]) : make_float3(cpB[3 * i], cpB[3 * i + 1], cpB[3 * i + 2]);
            const bodyID_t o = isA ? oA : oB;
            if (!isA) {
                force = force * -1.f;
                torque = torque * -1.f;
            }
            const float4 q = sn.q[o];
            if (torques) {  // the torque-only force becomes a torque about the contact point, in the owner's frame
                rotate(torque, {-q.x, -q.y, -q.z, q.w});
                torque = {pnt.y * torque.z - pnt.z * torque.y, pnt.z * torque.x - pnt.x * torque.z, pnt.x * torque.y - pnt.y * torque.x};
                if (!torque_in_local)
                    rotate(torque, q);
                torques->push_back(torque);
            }
            rotate(pnt, q);
            points.push_back(pnt + sn.com[o]);
            forces.push_back(force);
        }
        return points.size();
    }
    /// ShowTimingStats / ShowThreadCollaborationStats (API.h:1290-1300): the kernels' mean times from HIP events
    void ShowTimingStats() {
        for (const char* name : {"calc_forces", "integrate", "detect"}) {
            double ms = 0;
            uint64_t n = 0;
            if (deme_kernel_time_ms(diag_ctx(), name, &ms, &n) == DEME_OK)
                std::printf("%-12s %10.4f ms per launch over %llu launches\n", name, ms, (unsigned long long)n);
        }
    }
    void ShowThreadCollaborationStats() {
        const DemeCounts c = api_counts();
        std::printf("steps %llu, contact detections %llu (every %u steps), contacts %llu\n", (unsigned long long)c.nSteps,
                    (unsigned long long)c.nDetections, m_cd_freq, (unsigned long long)c.nContacts);
    }
    void ShowAnomalies() {}
    std::shared_ptr<DEMForceModel> GetContactForceModel() { return m_force_model; }
    void EnableOwnerWildcardOutput(bool enable = true) { m_out_content = enable ? (m_out_content | OWNER_WILDCARD) : (m_out_content & ~OWNER_WILDCARD); }
    void EnableGeometryWildcardOutput(bool enable = true) { m_out_content = enable ? (m_out_content | GEO_WILDCARD) : (m_out_content & ~GEO_WILDCARD); }
    void EnableContactWildcardOutput(bool enable = true) {
        m_cnt_out_content = enable ? (m_cnt_out_content | CNT_WILDCARD) : (m_cnt_out_content & ~CNT_WILDCARD);
    }
    /// (no reference equivalent) which kernel evaluates the current contact list, whether the engine keeps the clumps in an order of
    /// its own (ids seen here are always load order), and how many owner tiles go through the per-tile fallback
    std::string GetForceKernelName() const {
        char name[64] = {0};
        deme_force_kernel_name(diag_ctx(), name, sizeof(name), nullptr, nullptr);
        return name;
    }
    bool IsEngineReordered() const {
        int r = 0;
        deme_get_order(diag_ctx(), &r, nullptr);
        return r != 0;
    }
    unsigned GetNumFallbackTiles() const {
        uint32_t t[4] = {0, 0, 0, 0};
        deme_tile_stats(diag_ctx(), t);
        return t[1];
    }
    /// ShowMemStats (API.h:584): device memory in use by this process, as the HIP runtime reports it through the library
    void ShowMemStats() const {
        size_t used = 0, total = 0;
        if (deme_device_memory(diag_ctx(), &used, &total) == DEME_OK)
            std::printf("Device memory in use: %.1f MiB of %.1f MiB\n", used / 1048576.0, total / 1048576.0);
    }
    /// average number of contacts per sphere (kT's avgCntsPerSphere, API.h:251): contacts of the current list / spheres
    float GetAvgSphContacts() {
        const DemeCounts c = api_counts();
        return m_keep.sphOwner.empty() ? 0.f : (float)((double)c.nContacts / (double)m_keep.sphOwner.size());
    }
    /// SetFamilyClumpMaterial / SetFamilyMeshMaterial (API.h:970-974, dT::setFamilyClumpMaterial): every sphere (triangle) whose
    /// owner is of family N takes the material, from the next step on
    void SetFamilyClumpMaterial(unsigned int N, const std::shared_ptr<DEMMaterial>& mat) {
        require_init("SetFamilyClumpMaterial");
        m_family_material_log.push_back({N, (unsigned)mat->load_order, 0});
        each_ctx([&](deme_ctx* c) { return deme_set_family_material(c, N, mat->load_order, 0); });
    }
    void SetFamilyMeshMaterial(unsigned int N, const std::shared_ptr<DEMMaterial>& mat) {
        require_init("SetFamilyMeshMaterial");
        m_family_material_log.push_back({N, (unsigned)mat->load_order, 1});
        each_ctx([&](deme_ctx* c) { return deme_set_family_material(c, N, mat->load_order, 1); });
    }
    struct FamilyMaterial {
        unsigned family, material;
        int mesh;
    };
    std::vector<FamilyMaterial> m_family_material_log;  // replayed on the new slab contexts of a re-planned decomposed run
    void require_init(const char* who) const {
        if (!m_initialized)
            throw std::runtime_error(std::string(who) + " can only be called after the simulation system is initialized");
    }
    void ClearTimingStats() { deme_kernel_time_reset(diag_ctx()); }
    void ClearThreadCollaborationStats() {}
    void UseCubForceCollection(bool = true) {}  // accumulation is atomics-free here: nothing to choose
    void SetExpandSafetyType(const std::string& insp_type) {  // DEM/APIPublic.cpp:836-843: only "auto" exists
        if (insp_type != "auto")
            throw std::runtime_error("Unknown string input \"" + insp_type + "\" for SetExpandSafetyType.");
    }

    // ---- inspectors and trackers (API.h:652-679, AuxClasses.h:26-420)
    std::shared_ptr<class DEMInspector> CreateInspector(const std::string& quantity = "clump_max_z");
    /// region: statements returning a bool from X, Y, Z, e.g. "return (X * X + Y * Y <= 0.25 * 0.25) && (Z <= -0.3);"
    std::shared_ptr<class DEMInspector> CreateInspector(const std::string& quantity, const std::string& region);
    std::shared_ptr<class DEMTracker> Track(const std::shared_ptr<DEMClumpBatch>& batch);
    std::shared_ptr<class DEMTracker> Track(const std::shared_ptr<DEMExternObj>& obj);
    std::shared_ptr<class DEMTracker> Track(const std::shared_ptr<DEMMeshConnected>& mesh);
    friend class DEMInspector;
    friend class DEMTracker;

    // ---- output (API.h:1096-1122, 1318-1324); formats of dT.cpp:1254-1405, 1491-1618, 1620-1848
    void SetOutputFormat(OUTPUT_FORMAT f) { require_csv(f); }
    void SetContactOutputFormat(OUTPUT_FORMAT f) { require_csv(f); }
    void SetOutputContent(unsigned int content) { m_out_content = content; }
    void SetContactOutputContent(unsigned int content) { m_cnt_out_content = content; }
    // the string forms (API.h:1340-1352, APIPublic.cpp): format names and lists of content names
    static OUTPUT_FORMAT format_from(const std::string& f, const char* who) {
        const std::string u = upper(f);
        if (u == "CSV")
            return OUTPUT_FORMAT::CSV;
        if (u == "BINARY")
            return OUTPUT_FORMAT::BINARY;
        if (u == "CHPF")
            return OUTPUT_FORMAT::CHPF;
        throw std::runtime_error("Instruction " + f + " is unknown in " + who + " call.");
    }
    void SetOutputFormat(const std::string& format) { SetOutputFormat(format_from(format, "SetOutputFormat")); }
    void SetContactOutputFormat(const std::string& format) { SetContactOutputFormat(format_from(format, "SetContactOutputFormat")); }
    void SetOutputContent(const std::vector<std::string>& content) {
        static const std::map<std::string, unsigned int> names = {{"XYZ", XYZ}, {"QUAT", QUAT}, {"ABSV", ABSV}, {"VEL", VEL},
            {"ANG_VEL", ANG_VEL}, {"ABS_ACC", ABS_ACC}, {"ACC", ACC}, {"ANG_ACC", ANG_ACC}, {"FAMILY", FAMILY}, {"MAT", MAT},
            {"OWNER_WILDCARD", OWNER_WILDCARD}, {"GEO_WILDCARD", GEO_WILDCARD}};
        unsigned int c = XYZ;
        for (auto& n : content) {
            auto it = names.find(upper(n));
            if (it == names.end())
                throw std::runtime_error("Instruction " + n + " is unknown in SetOutputContent call.");
            c |= it->second;
        }
        m_out_content = c;
    }
    void SetContactOutputContent(const std::vector<std::string>& content) {
        static const std::map<std::string, unsigned int> names = {{"CNT_TYPE", CNT_TYPE}, {"FORCE", FORCE}, {"POINT", CNT_POINT},
            {"COMPONENT", COMPONENT}, {"NORMAL", NORMAL}, {"TORQUE", TORQUE}, {"CNT_WILDCARD", CNT_WILDCARD}, {"OWNER", OWNER},
            {"GEO_ID", GEO_ID}, {"NICKNAME", NICKNAME}};
        unsigned int c = CNT_TYPE;
        for (auto& n : content) {
            auto it = names.find(upper(n));
            if (it == names.end())
                throw std::runtime_error("Instruction " + n + " is unknown in SetContactOutputContent call.");
            c |= it->second;
        }
        m_cnt_out_content = c;
    }
    void WriteSphereFile(const std::string& outfilename) {
        Snapshot sn = snapshot(false);
        if (m_out_content & OWNER_WILDCARD)
            sn.ownerWc = wildcard_arrays(0, m_n_owners, m_force_model->owner_wildcards);
        if (m_out_content & GEO_WILDCARD)
            sn.sphereWc = wildcard_arrays(1, m_keep.sphOwner.size(), m_force_model->geo_wildcards);
        std::ostringstream o;
        o << "X,Y,Z,r";
        owner_header(o);
        if (m_out_content & GEO_WILDCARD)  // dT.cpp:1297-1301
            for (auto& n : m_force_model->geo_wildcards)
                o << "," << n;
        o << "\n";
        for (size_t i = 0; i < m_keep.sphOwner.size(); i++) {
            const uint32_t ow = m_keep.sphOwner[i];
            if (m_no_output_families.count(sn.fam[ow]))
                continue;
            const uint16_t cp = m_keep.sphComp[i];
            float3 d = {m_keep.rx[cp], m_keep.ry[cp], m_keep.rz[cp]};
            rotate(d, sn.q[ow]);
            const float3 pos = sn.com[ow] + d;
            o << pos.x << "," << pos.y << "," << pos.z << "," << m_keep.Radii[cp];
            owner_columns(o, sn, ow);  // (a sphere's owner wildcards are its owner's; the reference indexes them with the sphere id)
            for (auto& arr : sn.sphereWc)
                o << "," << arr[i];
            o << "\n";
        }
        flush(outfilename, o);
    }
    void WriteClumpFile(const std::string& outfilename, unsigned int accuracy = 10) {
        Snapshot sn = snapshot(false);
        if (m_out_content & OWNER_WILDCARD)
            sn.ownerWc = wildcard_arrays(0, m_n_owners, m_force_model->owner_wildcards);
        std::ostringstream o;
        o.precision(accuracy);
        o << "X,Y,Z,Qw,Qx,Qy,Qz,clump_type";
        owner_header(o);
        o << "\n";
        for (size_t i = 0; i < m_n_clumps; i++) {
            if (m_no_output_families.count(sn.fam[i]))
                continue;
            o << sn.com[i].x << "," << sn.com[i].y << "," << sn.com[i].z;
            o << "," << sn.q[i].w << "," << sn.q[i].x << "," << sn.q[i].y << "," << sn.q[i].z;
            o << "," << m_keep.templateName.at(m_keep.inert[i]);
            owner_columns(o, sn, i);
            o << "\n";
        }
        flush(outfilename, o);
    }
    /// Contacts whose |force + torque-only force| reaches force_thres; needs SetContactOutputContent before Initialize if the
    /// force / point / torque columns are wanted (per-contact records are switched on at Initialize).
    /// every pair of the list, the potential (not touching) ones included (API.h:1110-1116)
    void WriteContactFileIncludingPotentialPairs(const std::string& outfilename) { WriteContactFile(outfilename, -1.0f); }
    void WriteContactFile(const std::string& outfilename, float force_thres = DEME_TINY_FLOAT_HOST) {
        const Snapshot sn = snapshot(true);
        const unsigned fl = m_cnt_out_content;
        std::ostringstream o;
        o << "contact_type";
        if (fl & OWNER) o << ",A,B";
        if (fl & GEO_ID) o << ",geoA,geoB";
        if (fl & FORCE) o << ",f_x,f_y,f_z";
        if (fl & CNT_POINT) o << ",X,Y,Z";
        if (fl & NORMAL) o << ",n_x,n_y,n_z";
        if (fl & TORQUE) o << ",torque_x,torque_y,torque_z";
        if (fl & CNT_WILDCARD)
            for (auto& n : m_force_model->contact_wildcards) o << "," << n;
        o << "\n";
        for (size_t c = 0; c < sn.idA.size(); c++) {
            const float3 F = {sn.F[3 * c], sn.F[3 * c + 1], sn.F[3 * c + 2]}, T = {sn.T[3 * c], sn.T[3 * c + 1], sn.T[3 * c + 2]};
            const float3 tot = F + T;
            if (std::sqrt(tot.x * tot.x + tot.y * tot.y + tot.z * tot.z) < force_thres)
                continue;
            const uint8_t ty = sn.type[c];
            const uint32_t oA = m_keep.sphOwner[sn.idA[c]];
            const uint32_t oB = ty == 1 ? m_keep.sphOwner[sn.idB[c]] : ty == 2 ? m_keep.triOwner[sn.idB[c]] : m_keep.objOwner[sn.idB[c]];
            o << (ty == 1 ? "SS" : ty == 2 ? "SM" : "SA");
            if (fl & OWNER) o << "," << oA << "," << oB;
            if (fl & GEO_ID) o << "," << sn.idA[c] << "," << sn.idB[c];
            if (fl & FORCE) o << "," << F.x << "," << F.y << "," << F.z;
            const float3 loc = {sn.cpA[3 * c], sn.cpA[3 * c + 1], sn.cpA[3 * c + 2]};
            float3 pnt = loc;
            rotate(pnt, sn.q[oA]);
            pnt = pnt + sn.com[oA];
            if (fl & CNT_POINT) o << "," << pnt.x << "," << pnt.y << "," << pnt.z;
            if (fl & NORMAL) {
                const uint16_t cp = m_keep.sphComp[sn.idA[c]];
                float3 d = {m_keep.rx[cp], m_keep.ry[cp], m_keep.rz[cp]};
                rotate(d, sn.q[oA]);
                const float3 n = pnt - (sn.com[oA] + d);
                const float inv = 1.0f / std::sqrt(n.x * n.x + n.y * n.y + n.z * n.z);
                o << "," << n.x * inv << "," << n.y * inv << "," << n.z * inv;
            }
            if (fl & TORQUE) {
                float3 t = T;
                const float4 qa = sn.q[oA];
                rotate(t, {-qa.x, -qa.y, -qa.z, qa.w});
                t = {loc.y * t.z - loc.z * t.y, loc.z * t.x - loc.x * t.z, loc.x * t.y - loc.y * t.x};
                rotate(t, qa);
                o << "," << t.x << "," << t.y << "," << t.z;
            }
            if (fl & CNT_WILDCARD)
                for (size_t w = 0; w < sn.wc.size(); w++) o << "," << sn.wc[w][c];
            o << "\n";
        }
        flush(outfilename, o);
    }

    /// GetContactDetailedInfo (API.h: the contact list with everything WriteContactFile would print, as vectors): contacts whose
    /// force (incl. the torque-only part) is below force_thres are left out, like in the contact file
    std::shared_ptr<ContactInfoContainer> GetContactDetailedInfo(float force_thres = DEME_TINY_FLOAT_HOST) const {
        DEMSolver* self = const_cast<DEMSolver*>(this);  // queries download device state: logically const
        const Snapshot sn = self->snapshot(true);
        const unsigned fl = m_cnt_out_content;
        auto out = std::make_shared<ContactInfoContainer>(fl);
        const auto fam = self->owner_families();
        std::vector<std::string> wnames(m_force_model->contact_wildcards.begin(), m_force_model->contact_wildcards.end());
        for (size_t c = 0; c < sn.idA.size(); c++) {
            const float3 F = {sn.F[3 * c], sn.F[3 * c + 1], sn.F[3 * c + 2]}, T = {sn.T[3 * c], sn.T[3 * c + 1], sn.T[3 * c + 2]};
            const float3 tot = F + T;
            if (std::sqrt(tot.x * tot.x + tot.y * tot.y + tot.z * tot.z) < force_thres)
                continue;
            const uint8_t ty = sn.type[c];
            const uint32_t oA = m_keep.sphOwner[sn.idA[c]];
            const uint32_t oB = ty == 1 ? m_keep.sphOwner[sn.idB[c]] : ty == 2 ? m_keep.triOwner[sn.idB[c]] : m_keep.objOwner[sn.idB[c]];
            out->m_type.push_back(ty == 1 ? "SS" : ty == 2 ? "SM" : "SA");
            out->m_famA.push_back((uint8_t)fam.at(oA)), out->m_famB.push_back((uint8_t)fam.at(oB));
            if (fl & OWNER)
                out->m_ownerA.push_back(oA), out->m_ownerB.push_back(oB);
            if (fl & GEO_ID)
                out->m_geoA.push_back(sn.idA[c]), out->m_geoB.push_back(sn.idB[c]);
            if (fl & FORCE)
                out->m_force.push_back(F);
            const float3 loc = {sn.cpA[3 * c], sn.cpA[3 * c + 1], sn.cpA[3 * c + 2]};
            float3 pnt = loc;
            rotate(pnt, sn.q[oA]);
            pnt = pnt + sn.com[oA];
            if (fl & CNT_POINT)
                out->m_point.push_back(pnt);
            if (fl & NORMAL) {
                const uint16_t cp = m_keep.sphComp[sn.idA[c]];
                float3 d = {m_keep.rx[cp], m_keep.ry[cp], m_keep.rz[cp]};
                rotate(d, sn.q[oA]);
                const float3 n = pnt - (sn.com[oA] + d);
                const float inv = 1.0f / std::sqrt(n.x * n.x + n.y * n.y + n.z * n.z);
                out->m_normal.push_back({n.x * inv, n.y * inv, n.z * inv});
            }
            if (fl & TORQUE) {
                float3 t = T;
                const float4 qa = sn.q[oA];
                rotate(t, {-qa.x, -qa.y, -qa.z, qa.w});
                t = {loc.y * t.z - loc.z * t.y, loc.z * t.x - loc.x * t.z, loc.x * t.y - loc.y * t.x};
                rotate(t, qa);
                out->m_torque.push_back(t);
            }
            if (fl & CNT_WILDCARD)
                for (size_t w = 0; w < sn.wc.size() && w < wnames.size(); w++)
                    out->m_wc[wnames[w]].push_back(sn.wc[w][c]);
        }
        return out;
    }

    // ---- CSV readers (static members of the reference's DEMSolver, API.h:1153-1250)
    static std::unordered_map<std::string, std::vector<float3>> ReadClumpXyzFromCsv(const std::string& f) {
        return read_float3_by_type(f, "X", "Y", "Z");
    }
    static std::unordered_map<std::string, std::vector<float3>> ReadClumpVelFromCsv(const std::string& f) {
        return read_float3_by_type(f, "v_x", "v_y", "v_z");
    }
    static std::unordered_map<std::string, std::vector<float3>> ReadClumpAngVelFromCsv(const std::string& f) {
        return read_float3_by_type(f, "w_x", "w_y", "w_z");
    }
    static std::unordered_map<std::string, std::vector<float4>> ReadClumpQuatFromCsv(const std::string& f) {
        const Table t = read_table(f);
        const size_t ty = t.col("clump_type"), w = t.col("Qw"), x = t.col("Qx"), y = t.col("Qy"), z = t.col("Qz");
        std::unordered_map<std::string, std::vector<float4>> out;
        for (auto& r : t.rows)
            out[r[ty]].push_back({std::stof(r[x]), std::stof(r[y]), std::stof(r[z]), std::stof(r[w])});
        return out;
    }
    static std::vector<std::pair<bodyID_t, bodyID_t>> ReadContactPairsFromCsv(const std::string& f, const std::string& cntType = "SS",
                                                                             const std::string& cntColName = "contact_type",
                                                                             const std::string& first_name = "geoA",
                                                                             const std::string& second_name = "geoB") {
        const Table t = read_table(f);
        const size_t ty = t.col(cntColName), a = t.col(first_name), b = t.col(second_name);
        std::vector<std::pair<bodyID_t, bodyID_t>> out;
        for (auto& r : t.rows)
            if (r[ty] == cntType)
                out.push_back({(bodyID_t)std::stoul(r[a]), (bodyID_t)std::stoul(r[b])});
        return out;
    }
    /// Every column that is not a known contact-file column is a wildcard -- which includes X, Y, Z, as in the reference
    /// (CNT_FILE_KNOWN_COL_NAMES, Structs.h:75-84).
    static std::unordered_map<std::string, std::vector<float>> ReadContactWildcardsFromCsv(const std::string& f,
                                                                                          const std::string& cntType = "SS",
                                                                                          const std::string& cntColName = "contact_type") {
        static const std::set<std::string> known = {"A", "B", "compA", "compB", "geoA", "geoB", "nameA", "nameB", "contact_type",
                                                    "f_x", "f_y", "f_z", "torque_x", "torque_y", "torque_z", "n_x", "n_y", "n_z",
                                                    "SS", "SA", "SM"};
        const Table t = read_table(f);
        const size_t ty = t.col(cntColName);
        std::unordered_map<std::string, std::vector<float>> out;
        for (size_t c = 0; c < t.header.size(); c++) {
            if (known.count(t.header[c]))
                continue;
            auto& v = out[t.header[c]];
            for (auto& r : t.rows)
                if (r[ty] == cntType)
                    v.push_back(std::stof(r[c]));
        }
        return out;
    }

  private:
    deme_ctx* m_ctx = nullptr;       // the context of a one-device, one-slab run; of a decomposed run: the FIRST slab's (owned by m_multi)
    deme_multi* m_multi = nullptr;   // a decomposed run: several devices and / or several slabs per device
    std::vector<int> m_devices;
    unsigned int m_slabs_per_device = 1, m_migrate_every = 1000, m_rebalance_every = 0, m_replan_every = 0;
    uint64_t m_steps_since_replan = 0;
    // a scene re-upload (UpdateClumps, ResortClumps) hands the owners' CURRENT state to initialize_impl: a decomposed run is cut by
    // where the clumps are now, not by where their batches were loaded (row o of the state goes to owner dst(o, new clump count))
    const DemeOwnerState* m_reupload_state = nullptr;
    struct ReuploadGuard {  // the re-upload hand-over points at its caller's frame: never past that frame's end
        DEMSolver* s;
        ~ReuploadGuard() { s->m_reupload_state = nullptr, s->m_reupload_dst = nullptr; }
    };
    size_t m_reupload_n = 0;
    std::function<size_t(size_t, size_t)> m_reupload_dst;
    float m_slab_halo = 0.f;

    void open_devices(const std::vector<int>& ids) {
        int visible = 0;
        deme_device_count(&visible);
        if (ids.empty())
            throw std::runtime_error("DEMSolver: at least one device id");
        for (int d : ids)
            if (d < 0 || d >= visible)  // GpuManager.cpp:64-68: "more GPUs are requested than available"
                throw std::runtime_error("DEMSolver: device id " + std::to_string(d) + " is not present (" + std::to_string(visible) +
                                         " HIP device(s) visible)");
        m_devices = ids;
        if (const char* e = std::getenv("DEME_SLABS_PER_DEVICE"))
            m_slabs_per_device = (unsigned)std::max(1, atoi(e));
        if (const char* e = std::getenv("DEME_SLAB_MIGRATE_EVERY"))  // (SetSlabMigrationInterval)
            m_migrate_every = (unsigned)std::max(0, atoi(e));
        if (const char* e = std::getenv("DEME_SLAB_REBALANCE_EVERY"))  // (SetSlabRebalanceInterval)
            m_rebalance_every = (unsigned)std::max(0, atoi(e));
        if (const char* e = std::getenv("DEME_SLAB_REPLAN_EVERY"))  // (SetSlabReplanInterval)
            m_replan_every = (unsigned)std::max(0, atoi(e));
        if (const char* e = std::getenv("DEME_SLAB_HALO"))  // ghost layer thickness [m] of an unchanged script (SetSlabHalo)
            m_slab_halo = (float)atof(e);
        if (ids.size() == 1) {  // (a decomposed run on one device opens its deme_multi at Initialize, when the slab count is final)
            if (deme_ctx_create(ids[0], &m_ctx) != DEME_OK)
                throw std::runtime_error("DEMSolver: no usable HIP device");
            deme_set_fused_step(m_ctx, 2);  // small beds (<= 1e5 owners) step in one launch where the scene allows it (include/deme_hip.h)
        } else {
            char err[512];
            if (deme_multi_create(ids.data(), (int)ids.size(), &m_multi, err, sizeof err) != DEME_OK)
                throw std::runtime_error(std::string("DEMSolver: ") + err);
        }
        m_force_model = std::make_shared<DEMForceModel>();
        m_force_model->SetForceModelType(FORCE_MODEL::HERTZIAN);
        m_family_flags[RESERVED_FAMILY_NUM] = DEME_FAMILY_FIXED;
    }
    bool decomposed() const { return m_multi != nullptr; }
    /// diagnostics (kernel names, timers, tile statistics, device memory) of a decomposed run are its first slab's
    deme_ctx* diag_ctx() const {
        if (!m_multi)
            return m_ctx;
        const deme_ctx* c = nullptr;
        deme_multi_slab_ctx_peek(m_multi, 0, &c);  // (asking does not invalidate the run's merged contact list)
        return const_cast<deme_ctx*>(c);
    }
    /// what a decomposed run does not offer yet says so instead of touching one slab only
    // a user model's wildcard arrays: one context's, or by global id across the slabs of a decomposed run
    void ul_wc_array(uint32_t kind, uint32_t j, const float* v, size_t n) {
        if (m_multi)
            mcheck(deme_multi_upload_wildcard_array(m_multi, kind, j, v, n));
        else
            check(deme_upload_wildcard_array(m_ctx, kind, j, v, n));
    }
    void dl_wc_array(uint32_t kind, uint32_t j, float* v, size_t n) {
        if (m_multi)
            mcheck(deme_multi_download_wildcard_array(m_multi, kind, j, v, n));
        else
            check(deme_download_wildcard_array(m_ctx, kind, j, v, n));
    }
    // seeded / marked contacts: one context's, or in global ids across the slabs of a decomposed run
    void mc_seed_contacts(const uint32_t* a, const uint32_t* b, const uint8_t* ty, const float* w, size_t n) {
        if (m_multi)
            mcheck(deme_multi_seed_contacts(m_multi, a, b, ty, w, n));
        else
            check(deme_seed_contacts(m_ctx, a, b, ty, w, n));
    }
    void mc_num_persistent_contacts(size_t* n) {
        if (m_multi)
            mcheck(deme_multi_num_persistent_contacts(m_multi, n));
        else
            check(deme_num_persistent_contacts(m_ctx, n));
    }
    void mc_download_persistent_contacts(uint32_t* a, uint32_t* b, uint8_t* ty, size_t cap) {
        if (m_multi)
            mcheck(deme_multi_download_persistent_contacts(m_multi, a, b, ty, cap));
        else
            check(deme_download_persistent_contacts(m_ctx, a, b, ty, cap));
    }
    void mc_upload_persistent_contacts(const uint32_t* a, const uint32_t* b, const uint8_t* ty, size_t n) {
        if (m_multi)
            mcheck(deme_multi_upload_persistent_contacts(m_multi, a, b, ty, n));
        else
            check(deme_upload_persistent_contacts(m_ctx, a, b, ty, n));
    }
    void single_only(const char* what) const {
        if (m_multi)
            throw std::runtime_error(std::string(what) + " is not available on a decomposed run (several GPUs / DEME_SLABS_PER_DEVICE) yet");
    }
    void mcheck(int rc) {
        if (rc) {
            const std::string msg = deme_multi_last_error(m_multi);
            throw std::runtime_error(msg.empty() ? "a call on the decomposed run failed with status " + std::to_string(rc) : msg);
        }
    }
    /// the same call on the context, or on every slab's context
    template <class F>
    void each_ctx(F f) {
        if (!m_multi) {
            check(f(m_ctx));
            return;
        }
        uint32_t n = 0;
        mcheck(deme_multi_num_slabs(m_multi, &n));
        for (uint32_t i = 0; i < n; i++) {
            deme_ctx* c = nullptr;
            mcheck(deme_multi_slab_ctx(m_multi, i, &c));
            if (int rc = f(c))
                throw std::runtime_error(deme_last_error(c));
        }
    }
    void dl_state(DemeOwnerState* st) {
        if (m_multi)
            mcheck(deme_multi_download_state(m_multi, st, (uint32_t)m_n_owners));
        else
            check(deme_download_owner_state(m_ctx, st));
    }
    void ul_state(const DemeOwnerState* st) {
        if (m_multi)
            mcheck(deme_multi_upload_state(m_multi, st, (uint32_t)m_n_owners));
        else
            check(deme_upload_owner_state(m_ctx, st));
    }
    /// the contact list / per-contact arrays of the run: the context's, or the merged list of a decomposed run in global sphere ids
    /// (deme_multi_download_contacts: a pair that straddles a cut once; there is no history map across slabs: NULL entries)
    size_t n_contacts() {
        if (m_multi) {
            size_t n = 0;
            mcheck(deme_multi_num_contacts(m_multi, &n));
            return n;
        }
        DemeCounts c{};
        check(deme_get_counts(m_ctx, &c));
        return (size_t)c.nContacts;
    }
    void dl_contacts(uint32_t* a, uint32_t* b, uint8_t* ty, uint32_t* map, size_t nc) {
        if (m_multi) {
            mcheck(deme_multi_download_contacts(m_multi, a, b, ty, nc));
            if (map)
                std::fill(map, map + nc, 0xFFFFFFFFu);
        } else {
            check(deme_download_contacts(m_ctx, a, b, ty, map, nc));
        }
    }
    void dl_contact_wc(uint32_t w, float* col, size_t nc) {
        if (m_multi)
            mcheck(deme_multi_download_contact_wildcard(m_multi, w, col, nc));
        else
            check(deme_download_contact_wildcard(m_ctx, w, col, nc));
    }
    void dl_contact_records(float* f, float* t, float* a, float* b, size_t nc) {
        if (m_multi)
            mcheck(deme_multi_download_contact_records(m_multi, f, t, a, b, nc));
        else
            check(deme_download_contact_records(m_ctx, f, t, a, b, nc));
    }
    void add_owner_acc(uint32_t owner, uint32_t n, const float* acc, const float* angAcc) {
        if (m_multi)
            mcheck(deme_multi_add_owner_acc(m_multi, owner, n, acc, angAcc));
        else
            check(deme_add_owner_acc(m_ctx, owner, n, acc, angAcc));
    }
    DemeCounts api_counts() {
        DemeCounts c{};
        if (m_multi)
            mcheck(deme_multi_counts(m_multi, &c, nullptr));
        else
            check(deme_get_counts(m_ctx, &c));
        return c;
    }
    std::vector<std::shared_ptr<DEMMaterial>> m_materials;
    std::map<std::string, std::map<std::pair<unsigned, unsigned>, float>> m_pair_overrides;
    std::vector<std::shared_ptr<DEMClumpTemplate>> m_templates;
    std::vector<std::shared_ptr<DEMClumpBatch>> m_batches;
    std::vector<std::shared_ptr<DEMExternObj>> m_ext;
    std::vector<std::shared_ptr<DEMMeshConnected>> m_meshes;
    std::shared_ptr<DEMForceModel> m_force_model;
    std::string m_kernel_includes;
    float3 m_user_min{-10, -10, -10}, m_user_max{10, 10, 10}, m_target_min{-12, -12, -12}, m_target_max{12, 12, 12};
    std::string m_bounding = "none";
    std::shared_ptr<DEMMaterial> m_bounding_mat;
    float m_h = 1e-5f;
    float3 m_G{0, 0, -9.81f};
    unsigned m_cd_freq = 20;
    double m_bin_size = -1;
    size_t m_bin_num_target = 1000000;
    float m_bin_multiple = 8.0f, m_safety_multi = 1.f, m_safety_adder = 3.f /* API.h:1484 m_expand_base_vel */, m_max_vel = 1e15f, m_err_vel = 1e15f;
    TIME_INTEGRATOR m_integrator = TIME_INTEGRATOR::EXTENDED_TAYLOR;
    uint8_t m_family_masks[DEME_FAMILY_MASK_ENTRIES] = {0};
    float m_family_extra[DEME_NUM_FAMILIES] = {0};
    uint8_t m_family_flags[DEME_NUM_FAMILIES] = {0};
    DemeParams m_p{};
    size_t m_n_clumps = 0, m_n_owners = 0;
    // a user model's owner / geometry wildcard arrays across a scene re-upload (UpdateClumps, ResortClumps)
    struct SavedWildcards {
        std::vector<std::vector<float>> owner, sphere, tri, anal;
    };
    SavedWildcards save_user_wildcards() const {
        SavedWildcards w;
        if (m_force_model->type != FORCE_MODEL::CUSTOM)
            return w;
        w.owner = wildcard_arrays(0, m_n_owners, m_force_model->owner_wildcards);
        w.sphere = wildcard_arrays(1, m_keep.sphOwner.size(), m_force_model->geo_wildcards);
        if (!m_keep.triOwner.empty())
            w.tri = wildcard_arrays(2, m_keep.triOwner.size(), m_force_model->geo_wildcards);
        if (!m_keep.objOwner.empty())
            w.anal = wildcard_arrays(3, m_keep.objOwner.size(), m_force_model->geo_wildcards);
        return w;
    }
    template <typename FO, typename FS>
    void restore_user_wildcards(const SavedWildcards& w, FO&& new_owner, FS&& new_sphere) {
        for (uint32_t j = 0; j < w.owner.size(); j++) {
            std::vector<float> a(m_n_owners, 0.f);
            for (size_t o = 0; o < w.owner[j].size(); o++)
                a[new_owner(o)] = w.owner[j][o];
            ul_wc_array(0, j, a.data(), a.size());
        }
        for (uint32_t j = 0; j < w.sphere.size(); j++) {
            std::vector<float> a(m_keep.sphOwner.size(), 0.f);
            for (size_t i = 0; i < w.sphere[j].size(); i++)
                a[new_sphere(i)] = w.sphere[j][i];
            ul_wc_array(1, j, a.data(), a.size());
        }
        for (uint32_t j = 0; j < w.tri.size(); j++)
            ul_wc_array(2, j, w.tri[j].data(), w.tri[j].size());
        for (uint32_t j = 0; j < w.anal.size(); j++)
            ul_wc_array(3, j, w.anal[j].data(), w.anal[j].size());
    }
    static uint32_t wc_slot(const std::set<std::string>& names, const std::string& name, const char* what) {
        uint32_t j = 0;
        for (auto it = names.begin(); it != names.end(); ++it, ++j)
            if (*it == name)
                return j;
        throw std::runtime_error("no " + std::string(what) + " wildcard is named " + name);
    }
    std::vector<float> get_wildcard(uint32_t kind, size_t n, const std::set<std::string>& names, const std::string& name) {
        std::vector<float> a(n, 0.f);
        dl_wc_array(kind, wc_slot(names, name, kind ? "geometry" : "owner"), a.data(), n);
        return a;
    }
    template <typename F>
    void edit_wildcard(uint32_t kind, size_t n, const std::set<std::string>& names, const std::string& name, F&& edit) {
        std::vector<float> a = get_wildcard(kind, n, names, name);
        edit(a);
        ul_wc_array(kind, wc_slot(names, name, kind ? "geometry" : "owner"), a.data(), n);
    }
    void set_geo(uint32_t kind, size_t n, bodyID_t geoID, const std::string& name, const std::vector<float>& vals) {
        edit_wildcard(kind, n, m_force_model->geo_wildcards, name, [&](std::vector<float>& a) {
            for (size_t k = 0; k < vals.size() && geoID + k < a.size(); k++)
                a[geoID + k] = vals[k];
        });
    }
    std::vector<float> get_geo(uint32_t kind, size_t n_all, bodyID_t geoID, const std::string& name, size_t n) {
        const std::vector<float> a = get_wildcard(kind, n_all, m_force_model->geo_wildcards, name);
        return std::vector<float>(a.begin() + geoID, a.begin() + std::min(a.size(), (size_t)geoID + n));
    }
    std::vector<uint8_t> owner_families() {
        std::vector<uint8_t> fam(m_n_owners);
        DemeOwnerState st{};
        st.familyID = fam.data();
        dl_state(&st);
        return fam;
    }
    // mode 0 all, 1 either owner's family == N1, 2 both, 3 the pair (N1, N2): APIPrivate.cpp's setFamilyContactWildcardValue_impl
    void set_contact_wc(int mode, unsigned int N1, unsigned int N2, const std::string& name, float val) {
        const uint32_t w = wc_slot(m_force_model->contact_wildcards, name, "contact");
        const size_t nc = n_contacts();
        if (!nc)
            return;
        std::vector<uint32_t> a(nc), b(nc), map(nc);
        std::vector<uint8_t> ty(nc);
        dl_contacts(a.data(), b.data(), ty.data(), map.data(), nc);
        std::vector<float> col(nc);
        dl_contact_wc(w, col.data(), nc);
        const std::vector<uint8_t> fam = owner_families();
        for (size_t i = 0; i < nc; i++) {
            const unsigned fA = fam[m_keep.sphOwner[a[i]]];
            const unsigned fB = fam[ty[i] == DEME_SPHERE_SPHERE_CONTACT ? m_keep.sphOwner[b[i]]
                                    : (ty[i] == DEME_SPHERE_MESH_CONTACT ? m_keep.triOwner[b[i]] : m_keep.objOwner[b[i]])];
            const bool q = mode == 0 || (mode == 1 && (fA == N1 || fB == N1)) || (mode == 2 && fA == N1 && fB == N1) ||
                           (mode == 3 && ((fA == N1 && fB == N2) || (fA == N2 && fB == N1)));
            if (q)
                col[i] = val;
        }
        if (m_multi)
            mcheck(deme_multi_upload_contact_wildcard(m_multi, w, col.data(), nc));
        else
            check(deme_upload_contact_wildcard(m_ctx, w, col.data(), nc));
    }
    bool m_initialized = false;
    unsigned int m_max_sph_in_bin = 32768;  // API.h:1483 default
    std::set<unsigned int> m_no_output_families;  // familiesNoOutput (dT.cpp:1309-1312)
    float3 owner_column3(unsigned int owner, int which) {
        const Snapshot sn = snapshot(false);
        const std::vector<float3>& col = which == 2 ? sn.w : which == 3 ? sn.a : sn.al;
        return col.at(owner);
    }
    // reference defaults of the controllers' knobs (DEM/Structs.h:204-216); both switched off until asked for
    DemeAdaptive m_adaptive{0u, 25u, 0.05f, 0.1f, 0.25f, 0.3f, 0u, 2500u, 4u};
    void push_adaptive() {  // the knobs may be turned before or after Initialize
        if (m_initialized)
            each_ctx([&](deme_ctx* c) { return deme_set_adaptive(c, &m_adaptive); });
    }
    double m_time = 0;
    bool m_state_fresh = false;
    std::vector<float3> m_pos;
    std::vector<float> m_st_v[3];


    struct Presc {
        unsigned int family = 0;
        std::map<std::string, std::string> s;  // field name -> expression ("none": absent), names of familyPrescription_t
        bool flag[10] = {false, false, false, false, false, false, false, false, false, false};
    };
    struct FamilyRule {
        unsigned int from, to;
        std::string cond;
    };
    std::vector<Presc> m_presc_inputs;
    std::vector<FamilyRule> m_family_rules;
    void keep_as_is(unsigned int ID, std::initializer_list<int> flags) {
        Presc& q = new_presc(ID);
        for (int f : flags)
            q.flag[f] = true;
    }
    static std::string upper(std::string s) {
        for (auto& ch : s)
            ch = (char)std::toupper((unsigned char)ch);
        return s;
    }
    Presc& new_presc(unsigned int ID) {
        if (ID > 255)
            throw std::runtime_error("You applied prescribed motion to family " + std::to_string(ID) +
                                     ", but family number should not be larger than 255.");
        m_presc_inputs.emplace_back();
        m_presc_inputs.back().family = ID;
        m_family_flags[ID] |= DEME_FAMILY_PRESCRIBED;
        return m_presc_inputs.back();
    }
    static std::string replace_all(std::string s, const std::string& a, const std::string& b) {
        for (size_t p = s.find(a); p != std::string::npos; p = s.find(a, p + b.size()))
            s.replace(p, a.size(), b);
        return s;
    }
    // merge per family (APIPrivate.cpp:843-937) and emit the switch bodies of equipFamilyPrescribedMotions (:1600-1708)
    void compile_prescriptions_and_rules() {
        if (!m_presc_inputs.empty()) {
            std::map<unsigned, Presc> merged;
            for (auto& in : m_presc_inputs) {
                Presc& m = merged[in.family];
                m.family = in.family;
                for (auto& kv : in.s)
                    if (kv.second != "none")
                        m.s[kv.first] = kv.second;
                for (int k = 0; k < 10; k++)
                    m.flag[k] = m.flag[k] || in.flag[k];
            }
            std::string vel = " ", pos = " ", acc = " ";
            auto get = [](const Presc& m, const char* k) {
                auto it = m.s.find(k);
                return it == m.s.end() ? std::string("none") : it->second;
            };
            for (auto& kv : merged) {
                const Presc& m = kv.second;
                const std::string head = "case " + std::to_string(m.family) + ": {";
                auto block = [&](const char* pre, std::initializer_list<std::pair<const char*, const char*>> items) {
                    std::string o = "{";
                    if (get(m, pre) != "none")
                        o += get(m, pre) + ";";
                    for (auto& it : items)
                        if (get(m, it.second) != "none")
                            o += std::string(it.first) + " = " + get(m, it.second) + ";";
                    return o + "}";
                };
                vel += head + block("linVelPre", {{"vX", "linVelX"}, {"vY", "linVelY"}, {"vZ", "linVelZ"}}) +
                       block("rotVelPre", {{"omgBarX", "rotVelX"}, {"omgBarY", "rotVelY"}, {"omgBarZ", "rotVelZ"}});
                const char* vn[6] = {"LinVelXPrescribed", "LinVelYPrescribed", "LinVelZPrescribed", "RotVelXPrescribed",
                                     "RotVelYPrescribed", "RotVelZPrescribed"};
                for (int k = 0; k < 6; k++)
                    vel += std::string(vn[k]) + " = " + std::to_string(m.flag[k]) + ";";
                vel += "break; }";
                pos += head + block("linPosPre", {{"X", "linPosX"}, {"Y", "linPosY"}, {"Z", "linPosZ"}});
                if (get(m, "oriQ") != "none")
                    pos += "{" + replace_all(get(m, "oriQ"), "return", "float4 DEME_Presc_OriQ = ") +
                           ";oriQw = DEME_Presc_OriQ.w; oriQx = DEME_Presc_OriQ.x; oriQy = DEME_Presc_OriQ.y; oriQz = "
                           "DEME_Presc_OriQ.z;}";
                const char* pn[4] = {"LinXPrescribed", "LinYPrescribed", "LinZPrescribed", "RotPrescribed"};
                for (int k = 0; k < 4; k++)
                    pos += std::string(pn[k]) + " = " + std::to_string(m.flag[6 + k]) + ";";
                pos += "break; }";
                acc += head + block("accPre", {{"accX", "accX"}, {"accY", "accY"}, {"accZ", "accZ"}}) +
                       block("angAccPre", {{"angAccX", "angAccX"}, {"angAccY", "angAccY"}, {"angAccZ", "angAccZ"}}) + "break; }";
            }
            each_ctx([&](deme_ctx* c) { return deme_compile_prescriptions(c, vel.c_str(), pos.c_str(), acc.c_str()); });
        }
        if (!m_family_rules.empty()) {  // equipFamilyOnFlyChanges, APIPrivate.cpp:1576-1598
            std::string rules = " ";
            for (auto& r : m_family_rules)
                rules += "if (family_code == " + std::to_string(r.from) + ") { bool shouldMakeChange = false;" +
                         replace_all(r.cond, "return", "shouldMakeChange = ") +
                         "if (shouldMakeChange) {granData->familyID[myOwner] = " + std::to_string(r.to) + ";}}";
            each_ctx([&](deme_ctx* c) { return deme_compile_family_rules(c, rules.c_str()); });
        }
    }
    unsigned int m_out_content = QUAT | ABSV;                                      // API.h:1418
    unsigned int m_cnt_out_content = OWNER | GEO_ID | FORCE | CNT_POINT | CNT_WILDCARD;  // API.h:1422-1424
    struct Keep {  // scene arrays the writers need after Initialize
        std::vector<uint32_t> sphOwner, objOwner, triOwner;
        std::vector<uint16_t> sphComp, inert;
        std::vector<float> Radii, rx, ry, rz, mass, moix, moiy, moiz;
        std::map<unsigned, std::string> templateName;
    } m_keep;
    struct Snapshot {
        std::vector<float3> com, v, w, a, al;
        std::vector<float4> q;
        std::vector<uint8_t> fam;
        std::vector<uint32_t> idA, idB;
        std::vector<uint8_t> type;
        std::vector<float> F, T, cpA;
        std::vector<std::vector<float>> wc;
        std::vector<std::vector<float>> ownerWc, sphereWc;  // user-model owner / sphere wildcards (when the output asks for them)
    };
    static void require_csv(OUTPUT_FORMAT f) {
        if (f != OUTPUT_FORMAT::CSV)
            throw std::runtime_error("only OUTPUT_FORMAT::CSV is implemented");
    }
    // applyOriQToVector3<float, float> (DEMHelperKernels.cuh:162-173); float4 is (x, y, z, w)
    static void rotate(float3& v, float4 q) {
        const float w = q.w, x = q.x, y = q.y, z = q.z;
        const float ox = (2.0f * (w * w + x * x) - 1.0f) * v.x + (2.0f * (x * y - w * z)) * v.y + (2.0f * (x * z + w * y)) * v.z;
        const float oy = (2.0f * (x * y + w * z)) * v.x + (2.0f * (w * w + y * y) - 1.0f) * v.y + (2.0f * (y * z - w * x)) * v.z;
        const float oz = (2.0f * (x * z - w * y)) * v.x + (2.0f * (y * z + w * x)) * v.y + (2.0f * (w * w + z * z) - 1.0f) * v.z;
        v = {ox, oy, oz};
    }
    void owner_header(std::ostringstream& o) const {
        const unsigned fl = m_out_content;
        if (fl & ABSV) o << ",absv";
        if (fl & VEL) o << ",v_x,v_y,v_z";
        if (fl & ANG_VEL) o << ",w_x,w_y,w_z";
        if (fl & ABS_ACC) o << ",abs_acc";
        if (fl & ACC) o << ",a_x,a_y,a_z";
        if (fl & ANG_ACC) o << ",alpha_x,alpha_y,alpha_z";
        if (fl & FAMILY) o << ",family";
        if (fl & OWNER_WILDCARD)  // dT.cpp:1292-1296: named as the force model names them
            for (auto& n : m_force_model->owner_wildcards)
                o << "," << n;
    }
    // owner (kind 0) / sphere (kind 1) wildcard arrays of the user force model, in the model's name order
    std::vector<std::vector<float>> wildcard_arrays(uint32_t kind, size_t n_elem, const std::set<std::string>& names) const {
        std::vector<std::vector<float>> out;
        uint32_t j = 0;
        for (auto it = names.begin(); it != names.end(); ++it, ++j) {
            out.emplace_back(n_elem, 0.f);
            if (m_force_model->type == FORCE_MODEL::CUSTOM)
                const_cast<DEMSolver*>(this)->dl_wc_array(kind, j, out.back().data(), n_elem);
        }
        return out;
    }
    void owner_columns(std::ostringstream& o, const Snapshot& sn, size_t i) const {
        const unsigned fl = m_out_content;
        auto len = [](float3 a) { return std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z); };
        if (fl & ABSV) o << "," << len(sn.v[i]);
        if (fl & VEL) o << "," << sn.v[i].x << "," << sn.v[i].y << "," << sn.v[i].z;
        if (fl & ANG_VEL) o << "," << sn.w[i].x << "," << sn.w[i].y << "," << sn.w[i].z;
        if (fl & ABS_ACC) o << "," << len(sn.a[i]);
        if (fl & ACC) o << "," << sn.a[i].x << "," << sn.a[i].y << "," << sn.a[i].z;
        if (fl & ANG_ACC) o << "," << sn.al[i].x << "," << sn.al[i].y << "," << sn.al[i].z;
        if (fl & FAMILY) o << "," << +sn.fam[i];
        if (fl & OWNER_WILDCARD)
            for (auto& arr : sn.ownerWc)
                o << "," << arr[i];
    }
    static void flush(const std::string& path, const std::ostringstream& o) {
        std::ofstream f(path, std::ios::out);
        if (!f)
            throw std::runtime_error("cannot open " + path + " for writing");
        f << o.str();
    }
    Snapshot snapshot(bool contacts) {
        const size_t n = m_n_owners;
        std::vector<uint64_t> vid(n);
        std::vector<uint16_t> lx(n), ly(n), lz(n);
        std::vector<float> f[16];
        for (auto& v : f)
            v.resize(n);
        Snapshot sn;
        sn.fam.resize(n);
        DemeOwnerState st{};
        st.voxelID = vid.data(), st.locX = lx.data(), st.locY = ly.data(), st.locZ = lz.data();
        st.oriQw = f[0].data(), st.oriQx = f[1].data(), st.oriQy = f[2].data(), st.oriQz = f[3].data();
        st.vX = f[4].data(), st.vY = f[5].data(), st.vZ = f[6].data();
        st.omgBarX = f[7].data(), st.omgBarY = f[8].data(), st.omgBarZ = f[9].data();
        st.aX = f[10].data(), st.aY = f[11].data(), st.aZ = f[12].data();
        st.alphaX = f[13].data(), st.alphaY = f[14].data(), st.alphaZ = f[15].data();
        st.familyID = sn.fam.data();
        dl_state(&st);
        sn.com.resize(n), sn.q.resize(n), sn.v.resize(n), sn.w.resize(n), sn.a.resize(n), sn.al.resize(n);
        const float vs = (float)m_p.voxelSize, l = (float)m_p.l;
        for (size_t i = 0; i < n; i++) {  // voxelIDToPosition<float, ...> then + LBF, all fp32 (dT.cpp:1312-1320)
            const uint64_t vx = vid[i] & ((1ull << m_p.nvXp2) - 1), vy = (vid[i] >> m_p.nvXp2) & ((1ull << m_p.nvYp2) - 1),
                           vz = vid[i] >> (m_p.nvXp2 + m_p.nvYp2);
            sn.com[i] = {((float)vx * vs + (float)lx[i] * l) + m_p.LBFX, ((float)vy * vs + (float)ly[i] * l) + m_p.LBFY,
                         ((float)vz * vs + (float)lz[i] * l) + m_p.LBFZ};
            sn.q[i] = {f[1][i], f[2][i], f[3][i], f[0][i]};
            sn.v[i] = {f[4][i], f[5][i], f[6][i]};
            sn.w[i] = {f[7][i], f[8][i], f[9][i]};
            sn.a[i] = {f[10][i], f[11][i], f[12][i]};
            sn.al[i] = {f[13][i], f[14][i], f[15][i]};
        }
        if (contacts) {
            const size_t nc = n_contacts();
            sn.idA.resize(nc), sn.idB.resize(nc), sn.type.resize(nc);
            std::vector<uint32_t> map(nc);
            dl_contacts(sn.idA.data(), sn.idB.data(), sn.type.data(), map.data(), nc);
            sn.F.assign(3 * nc, 0.f), sn.T.assign(3 * nc, 0.f), sn.cpA.assign(3 * nc, 0.f);
            std::vector<float> cpB(3 * nc);
            dl_contact_records(sn.F.data(), sn.T.data(), sn.cpA.data(), cpB.data(), nc);
            sn.wc.assign(m_p.nContactWildcards, std::vector<float>(nc));
            for (uint32_t w = 0; w < m_p.nContactWildcards; w++)
                dl_contact_wc(w, sn.wc[w].data(), nc);
        }
        return sn;
    }
    struct Table {
        std::vector<std::string> header;
        std::vector<std::vector<std::string>> rows;
        size_t col(const std::string& name) const {
            for (size_t i = 0; i < header.size(); i++)
                if (header[i] == name)
                    return i;
            throw std::runtime_error("column " + name + " not found");
        }
    };
    static Table read_table(const std::string& path) {
        std::ifstream f(path);
        if (!f)
            throw std::runtime_error("cannot open " + path);
        Table t;
        std::string line;
        auto split = [](const std::string& ln) {
            std::vector<std::string> out;
            std::stringstream ss(ln);
            std::string tok;
            while (std::getline(ss, tok, ',')) {
                const size_t a = tok.find_first_not_of(" \t\r"), b = tok.find_last_not_of(" \t\r");
                out.push_back(a == std::string::npos ? std::string() : tok.substr(a, b - a + 1));
            }
            return out;
        };
        while (std::getline(f, line)) {
            if (line.find_first_not_of(" \t\r") == std::string::npos || line[line.find_first_not_of(" \t")] == '#')
                continue;
            if (t.header.empty())
                t.header = split(line);
            else
                t.rows.push_back(split(line));
        }
        return t;
    }
    static std::unordered_map<std::string, std::vector<float3>> read_float3_by_type(const std::string& f, const char* x, const char* y,
                                                                                    const char* z) {
        const Table t = read_table(f);
        const size_t ty = t.col("clump_type"), cx = t.col(x), cy = t.col(y), cz = t.col(z);
        std::unordered_map<std::string, std::vector<float3>> out;
        for (auto& r : t.rows)
            out[r[ty]].push_back({std::stof(r[cx]), std::stof(r[cy]), std::stof(r[cz])});
        return out;
    }

    void check(int rc) {
        if (rc)  // (a decomposed run keeps no single context: a call that has no decomposed form yet arrives here with a null one)
            throw std::runtime_error(m_multi ? "this call is not available on a decomposed run (several GPUs / DEME_SLABS_PER_DEVICE) yet"
                                             : deme_last_error(m_ctx));
    }
    // owners are numbered: clumps (batch load order), analytical objects (+ the bounding box last), meshes
    size_t tracker_first_owner(int kind, size_t index) const {
        if (m_n_owners == 0)
            throw std::runtime_error("trackers can be used after Initialize()");
        size_t first = 0;
        if (kind == 0) {
            for (size_t i = 0; i < index; i++)
                first += m_batches[i]->nClumps;
            return first;
        }
        if (kind == 1)
            return m_n_clumps + index;
        return m_n_owners - m_meshes.size() + index;
    }
    // all triangles' owner-local nodes are re-sent (deme_update_tri_nodes takes the full arrays, mesh-major)
    void update_mesh_nodes(size_t mesh_index, const std::vector<float3>& new_nodes) {
        auto& me = m_meshes.at(mesh_index);
        if (new_nodes.size() != me->vertices.size())
            throw std::runtime_error("UpdateMesh: the node count does not match the mesh");
        me->vertices = new_nodes;
        std::vector<float> t1, t2, t3;
        for (auto& m : m_meshes)
            for (auto& f : m->faces) {
                const float3 a = m->vertices.at(f[0]), b = m->vertices.at(f[1]), c = m->vertices.at(f[2]);
                t1.insert(t1.end(), {a.x, a.y, a.z}), t2.insert(t2.end(), {b.x, b.y, b.z}), t3.insert(t3.end(), {c.x, c.y, c.z});
            }
        each_ctx([&](deme_ctx* c) { return deme_update_tri_nodes(c, t1.data(), t2.data(), t3.data()); });
    }
    // tracker setters: read-modify-write of the affected SoA columns (null columns keep their device values)
    void set_owner(size_t o, const float3* pos, const float3* vel, const float3* angvel, const float4* q) {
        const size_t n = m_n_owners;
        std::vector<uint64_t> vid(n);
        std::vector<uint16_t> lx(n), ly(n), lz(n);
        std::vector<float> f[10];
        for (auto& v : f)
            v.resize(n);
        DemeOwnerState st{};
        st.voxelID = vid.data(), st.locX = lx.data(), st.locY = ly.data(), st.locZ = lz.data();
        st.oriQw = f[0].data(), st.oriQx = f[1].data(), st.oriQy = f[2].data(), st.oriQz = f[3].data();
        st.vX = f[4].data(), st.vY = f[5].data(), st.vZ = f[6].data();
        st.omgBarX = f[7].data(), st.omgBarY = f[8].data(), st.omgBarZ = f[9].data();
        dl_state(&st);
        if (pos) {  // positionToVoxelID of (pos - LBF), as at initialisation
            const double P[3] = {(double)(pos->x - m_p.LBFX), (double)(pos->y - m_p.LBFY), (double)(pos->z - m_p.LBFZ)};
            uint64_t nn[3];
            uint16_t ss[3];
            for (int k = 0; k < 3; k++) {
                nn[k] = (uint64_t)(P[k] / m_p.voxelSize);
                ss[k] = (uint16_t)((P[k] - (double)nn[k] * m_p.voxelSize) / m_p.l);
            }
            vid[o] = nn[0] + (nn[1] << m_p.nvXp2) + (nn[2] << (m_p.nvXp2 + m_p.nvYp2));
            lx[o] = ss[0], ly[o] = ss[1], lz[o] = ss[2];
        }
        if (q)
            f[0][o] = q->w, f[1][o] = q->x, f[2][o] = q->y, f[3][o] = q->z;
        if (vel)
            f[4][o] = vel->x, f[5][o] = vel->y, f[6][o] = vel->z;
        if (angvel)
            f[7][o] = angvel->x, f[8][o] = angvel->y, f[9][o] = angvel->z;
        ul_state(&st);
        m_state_fresh = false;
    }
    void step(uint32_t n) {
        while (n) {
            uint32_t k = n;
            if (m_multi && m_replan_every)
                k = (uint32_t)std::min<uint64_t>(n, m_replan_every - std::min<uint64_t>(m_steps_since_replan, m_replan_every - 1u));
            if (m_multi)
                mcheck(deme_multi_step(m_multi, k));
            else
                check(deme_step(m_ctx, k));
            m_time += (double)k * (double)m_h;
            m_steps_since_replan += k;
            m_state_fresh = false;
            n -= k;
            if (m_multi && m_replan_every && m_steps_since_replan >= m_replan_every)
                ReplanSlabs();  // (after the steps: what a script queued for its next step -- AddAcc -- has been consumed by then)
        }
    }

    void refresh_state() {
        if (m_state_fresh)
            return;
        std::vector<uint64_t> vid(m_n_owners);
        std::vector<uint16_t> lx(m_n_owners), ly(m_n_owners), lz(m_n_owners);
        for (auto& v : m_st_v)
            v.resize(m_n_owners);
        DemeOwnerState st{};
        st.voxelID = vid.data(), st.locX = lx.data(), st.locY = ly.data(), st.locZ = lz.data();
        st.vX = m_st_v[0].data(), st.vY = m_st_v[1].data(), st.vZ = m_st_v[2].data();
        dl_state(&st);
        m_pos.resize(m_n_owners);
        for (size_t i = 0; i < m_n_owners; i++) {  // voxelIDToPosition + LBF, DEMHelperKernels.cuh:116-134
            const uint64_t vx = vid[i] & ((1ull << m_p.nvXp2) - 1), vy = (vid[i] >> m_p.nvXp2) & ((1ull << m_p.nvYp2) - 1),
                           vz = vid[i] >> (m_p.nvXp2 + m_p.nvYp2);
            m_pos[i] = {(float)((double)vx * m_p.voxelSize + (double)lx[i] * m_p.l + m_p.LBFX),
                        (float)((double)vy * m_p.voxelSize + (double)ly[i] * m_p.l + m_p.LBFY),
                        (float)((double)vz * m_p.voxelSize + (double)lz[i] * m_p.l + m_p.LBFZ)};
        }
        m_state_fresh = true;
    }

    void figure_out_nv(unsigned nv[3], double& l, double& voxel) {
        float xyz[3] = {m_target_max.x - m_target_min.x, m_target_max.y - m_target_min.y, m_target_max.z - m_target_min.z};
        int rank[3] = {0, 1, 2};
        for (int i = 0; i < 2; i++)
            for (int j = i + 1; j < 3; j++)
                if (xyz[i] > xyz[j]) {
                    std::swap(xyz[i], xyz[j]);
                    std::swap(rank[i], rank[j]);
                }
        const float user321[3] = {xyz[0], xyz[1], xyz[2]};
        int more[2] = {0, 0};
        while (xyz[0] < xyz[1]) {
            if (std::sqrt(2.) * xyz[0] > xyz[1])
                break;
            more[0]++;
            xyz[0] *= 2.;
        }
        while (xyz[1] < xyz[2]) {
            if (std::sqrt(2.) * xyz[1] > xyz[2])
                break;
            more[1]++;
            xyz[1] *= 2.;
        }
        const int budget = 64 - 2 * more[0] - more[1];
        int b3 = budget / 3, left = budget % 3;
        int b2 = b3 + more[0], b1 = b2 + more[1];
        while (left-- > 0) {
            if (b3 < b2)
                b3++;
            else if (b2 < b1)
                b2++;
            else
                b1++;
        }
        const int bits[3] = {b3, b2, b1};
        l = 0;
        for (int k = 0; k < 3; k++)
            l = std::max(l, (double)user321[k] / std::pow(2., 16) / std::pow(2., bits[k]));
        for (int k = 0; k < 3; k++)
            nv[rank[k]] = (unsigned)bits[k];
        voxel = 65536.0 * l;
    }

    // records "name\0", element size (u32), element count (u64), raw bytes; the first record is the DemeParams block
    static void dump_scene(const char* path, const DemeParams& p, const DemeScene& s) {
        FILE* f = std::fopen(path, "wb");
        if (!f)
            throw std::runtime_error(std::string("DEME_DUMP_SCENE: cannot write ") + path);
        auto rec = [&](const char* name, const void* data, uint32_t esz, uint64_t n) {
            if (!data)
                n = 0;
            std::fwrite(name, 1, std::strlen(name) + 1, f);
            std::fwrite(&esz, 4, 1, f);
            std::fwrite(&n, 8, 1, f);
            if (n)
                std::fwrite(data, esz, n, f);
        };
        rec("DemeParams", &p, (uint32_t)sizeof(DemeParams), 1);
        const uint64_t nO = s.nOwners, nS = s.nSpheres, nK = s.nComp, nP = s.nMassProps, nA = s.nAnal, nM = s.nMat, nT = s.nTri;
        const uint32_t cnt[8] = {s.nOwners, s.nOwnerClumps, s.nSpheres, s.nAnal, s.nTri, s.nMat, s.nComp, s.nMassProps};
        rec("counts", cnt, 4, 8);
#define DEME_DUMP(field, n) rec(#field, s.field, (uint32_t)sizeof(*s.field), n)
        DEME_DUMP(voxelID, nO), DEME_DUMP(locX, nO), DEME_DUMP(locY, nO), DEME_DUMP(locZ, nO);
        DEME_DUMP(oriQw, nO), DEME_DUMP(oriQx, nO), DEME_DUMP(oriQy, nO), DEME_DUMP(oriQz, nO);
        DEME_DUMP(vX, nO), DEME_DUMP(vY, nO), DEME_DUMP(vZ, nO), DEME_DUMP(omgBarX, nO), DEME_DUMP(omgBarY, nO), DEME_DUMP(omgBarZ, nO);
        DEME_DUMP(familyID, nO), DEME_DUMP(inertiaPropOffsets, nO);
        DEME_DUMP(ownerClumpBody, nS), DEME_DUMP(clumpComponentOffset, nS), DEME_DUMP(sphereMaterialOffset, nS);
        DEME_DUMP(Radii, nK), DEME_DUMP(CDRelPosX, nK), DEME_DUMP(CDRelPosY, nK), DEME_DUMP(CDRelPosZ, nK);
        DEME_DUMP(MassProperties, nP), DEME_DUMP(moiX, nP), DEME_DUMP(moiY, nP), DEME_DUMP(moiZ, nP);
        DEME_DUMP(objType, nA), DEME_DUMP(objOwner, nA), DEME_DUMP(objNormal, nA), DEME_DUMP(objMaterial, nA);
        DEME_DUMP(objRelPosX, nA), DEME_DUMP(objRelPosY, nA), DEME_DUMP(objRelPosZ, nA);
        DEME_DUMP(objRotX, nA), DEME_DUMP(objRotY, nA), DEME_DUMP(objRotZ, nA);
        DEME_DUMP(objSize1, nA), DEME_DUMP(objSize2, nA), DEME_DUMP(objSize3, nA), DEME_DUMP(objMass, nA);
        DEME_DUMP(E, nM), DEME_DUMP(nu, nM), DEME_DUMP(CoR, nM * nM), DEME_DUMP(mu, nM * nM), DEME_DUMP(Crr, nM * nM);
        DEME_DUMP(familyMasks, (uint64_t)DEME_FAMILY_MASK_ENTRIES), DEME_DUMP(familyExtraMarginSize, (uint64_t)DEME_NUM_FAMILIES);
        DEME_DUMP(familyFlags, (uint64_t)DEME_NUM_FAMILIES);
        DEME_DUMP(ownerMesh, nT), DEME_DUMP(triNode1, nT * 3), DEME_DUMP(triNode2, nT * 3), DEME_DUMP(triNode3, nT * 3);
        DEME_DUMP(triMaterialOffset, nT);
#undef DEME_DUMP
        std::fclose(f);
    }

    void initialize_impl() {
        unsigned nv[3];
        double l, voxel;
        figure_out_nv(nv, l, voxel);
        const float3 lbf = m_target_min;
        // templates by component count (stable), marks renumbered
        std::vector<std::shared_ptr<DEMClumpTemplate>> ts = m_templates;
        std::stable_sort(ts.begin(), ts.end(), [](const auto& a, const auto& b) { return a->nComp < b->nComp; });
        std::vector<float> Radii, rx, ry, rz, mass, moix, moiy, moiz, volumes;
        std::vector<unsigned> prefix(ts.size());
        float smallest = 1e30f;
        for (size_t i = 0; i < ts.size(); i++) {
            ts[i]->mark = (unsigned)i;
            prefix[i] = (unsigned)Radii.size();
            for (size_t k = 0; k < ts[i]->radii.size(); k++) {
                Radii.push_back(ts[i]->radii[k]);
                rx.push_back(ts[i]->relPos[k].x), ry.push_back(ts[i]->relPos[k].y), rz.push_back(ts[i]->relPos[k].z);
                smallest = std::min(smallest, ts[i]->radii[k]);
            }
            mass.push_back(ts[i]->mass), moix.push_back(ts[i]->MOI.x), moiy.push_back(ts[i]->MOI.y), moiz.push_back(ts[i]->MOI.z);
            volumes.push_back(ts[i]->volume);
        }
        if (Radii.size() > 65535)
            throw std::runtime_error("more than 65535 clump components");
        double bin = m_bin_size > 0 ? m_bin_size : (double)(m_bin_multiple * smallest);  // a float product, like the reference's
        auto nbins = [&](uint32_t nb[3]) {
            for (int k = 0; k < 3; k++)
                nb[k] = (uint32_t)(voxel * (double)(1ull << nv[k]) / bin) + 1;
            return (uint64_t)nb[0] * nb[1] * nb[2];
        };
        uint32_t nb[3];
        if (m_bin_size <= 0 && m_bin_num_target) {  // DEMSolver::decideBinSize, APIPrivate.cpp:525-541
            const double tgt = (double)m_bin_num_target;
            uint64_t num = nbins(nb), prev = num;
            while ((double)num < 0.67 * tgt || (double)num > 1.5 * tgt) {
                bin *= (num < m_bin_num_target) ? 0.8 : 1.2;
                num = nbins(nb);
                if ((prev < m_bin_num_target && num >= m_bin_num_target) || (prev >= m_bin_num_target && num < m_bin_num_target))
                    break;
                prev = num;
            }
        }
        while (nbins(nb) > 0xFFFFFFFEull)
            bin *= 1.5;
        // bounding box planes
        std::vector<std::shared_ptr<DEMExternObj>> ext = m_ext;
        if (m_bounding != "none") {
            const bool bottom = m_bounding == "only_bottom" || m_bounding == "top_open" || m_bounding == "all";
            const bool sides = m_bounding == "only_sides" || m_bounding == "top_open" || m_bounding == "all";
            const bool top = m_bounding == "all";
            if (!bottom && !sides)
                throw std::runtime_error("Domain bounding BC instruction " + m_bounding + " is unknown.");
            auto box = std::make_shared<DEMExternObj>();
            const float3 c = {(m_user_min.x + m_user_max.x) / 2.f, (m_user_min.y + m_user_max.y) / 2.f, (m_user_min.z + m_user_max.z) / 2.f};
            if (bottom)
                box->AddPlane({c.x, c.y, m_user_min.z}, {0, 0, 1}, m_bounding_mat);
            if (sides) {
                box->AddPlane({m_user_min.x, c.y, c.z}, {1, 0, 0}, m_bounding_mat);
                box->AddPlane({m_user_max.x, c.y, c.z}, {-1, 0, 0}, m_bounding_mat);
                box->AddPlane({c.x, m_user_min.y, c.z}, {0, 1, 0}, m_bounding_mat);
                box->AddPlane({c.x, m_user_max.y, c.z}, {0, -1, 0}, m_bounding_mat);
            }
            if (top)
                box->AddPlane({c.x, c.y, m_user_max.z}, {0, 0, -1}, m_bounding_mat);
            ext.push_back(box);
        }
        // owners
        size_t nC = 0;
        for (auto& b : m_batches)
            nC += b->nClumps;
        const size_t nO = nC + ext.size() + m_meshes.size();
        std::vector<uint64_t> vid(nO);
        std::vector<uint16_t> lx(nO), ly(nO), lz(nO), inert(nO);
        std::vector<float> qw(nO, 1.f), qx(nO, 0.f), qy(nO, 0.f), qz(nO, 0.f), vx(nO, 0.f), vy(nO, 0.f), vz(nO, 0.f), wx(nO, 0.f),
            wy(nO, 0.f), wz(nO, 0.f);
        std::vector<uint8_t> fam(nO, 0);
        std::vector<uint32_t> sphOwner;
        std::vector<uint16_t> sphComp, sphMat;
        auto encode = [&](size_t o, float3 pos) {  // positionToVoxelID of (pos - LBF), dT.cpp:745-782
            const float3 d = pos - lbf;
            const double P[3] = {(double)d.x, (double)d.y, (double)d.z};
            uint64_t n[3];
            uint16_t s[3];
            for (int k = 0; k < 3; k++) {
                n[k] = (uint64_t)(P[k] / voxel);
                s[k] = (uint16_t)((P[k] - (double)n[k] * voxel) / l);
            }
            vid[o] = n[0] + (n[1] << nv[0]) + (n[2] << (nv[0] + nv[1]));
            lx[o] = s[0], ly[o] = s[1], lz[o] = s[2];
        };
        size_t o = 0;
        bool warned = false;
        for (auto& b : m_batches)
            for (size_t i = 0; i < b->nClumps; i++, o++) {
                const float3 c = b->xyz[i];
                if (!warned && (c.x < m_user_min.x || c.y < m_user_min.y || c.z < m_user_min.z || c.x > m_user_max.x ||
                                c.y > m_user_max.y || c.z > m_user_max.z)) {  // the reference's courtesy check, dT.cpp:739-744, 887-893
                    std::fprintf(stderr,
                                 "WARNING: At least one clump is initialized with a position out of the box domain you specified.\nIt is "
                                 "found at %.5g, %.5g, %.5g (this message only shows one such example).\nThis simulation is unlikely to "
                                 "go as planned.\n", c.x, c.y, c.z);
                    warned = true;
                }
                encode(o, b->xyz[i]);
                qw[o] = b->oriQ[i].w, qx[o] = b->oriQ[i].x, qy[o] = b->oriQ[i].y, qz[o] = b->oriQ[i].z;
                vx[o] = b->vel[i].x, vy[o] = b->vel[i].y, vz[o] = b->vel[i].z;
                wx[o] = b->angVel[i].x, wy[o] = b->angVel[i].y, wz[o] = b->angVel[i].z;
                fam[o] = (uint8_t)b->families[i];
                const auto& t = b->types[i];
                inert[o] = (uint16_t)t->mark;
                for (unsigned k = 0; k < t->nComp; k++) {
                    sphOwner.push_back((uint32_t)o);
                    sphComp.push_back((uint16_t)(prefix[t->mark] + k));
                    sphMat.push_back((uint16_t)t->materials[k]->load_order);
                }
            }
        std::vector<uint8_t> objType;
        std::vector<uint32_t> objOwner;
        std::vector<uint16_t> objMat;
        std::vector<float> objN, opx, opy, opz, orx, ory, orz, os1, os2, os3, om;
        for (size_t e = 0; e < ext.size(); e++, o++) {
            encode(o, ext[e]->init_pos);
            qw[o] = ext[e]->init_oriQ.w, qx[o] = ext[e]->init_oriQ.x, qy[o] = ext[e]->init_oriQ.y, qz[o] = ext[e]->init_oriQ.z;
            fam[o] = (uint8_t)ext[e]->family_code;
            inert[o] = (uint16_t)(ts.size() + e);
            mass.push_back(ext[e]->mass), moix.push_back(ext[e]->MOI.x), moiy.push_back(ext[e]->MOI.y), moiz.push_back(ext[e]->MOI.z);
            for (auto& cmp : ext[e]->comps) {
                objType.push_back(cmp.type), objOwner.push_back((uint32_t)o), objMat.push_back((uint16_t)cmp.mat->load_order);
                objN.push_back(cmp.normal_sign);
                opx.push_back(cmp.pos.x), opy.push_back(cmp.pos.y), opz.push_back(cmp.pos.z);
                orx.push_back(cmp.dir.x), ory.push_back(cmp.dir.y), orz.push_back(cmp.dir.z);
                os1.push_back(cmp.size1), os2.push_back(0.f), os3.push_back(0.f), om.push_back(ext[e]->mass);
            }
        }
        std::vector<uint32_t> triOwner;
        std::vector<float> t1, t2, t3;
        std::vector<uint16_t> triMat;
        for (size_t mI = 0; mI < m_meshes.size(); mI++, o++) {
            auto& me = m_meshes[mI];
            encode(o, me->init_pos);
            qw[o] = me->init_oriQ.w, qx[o] = me->init_oriQ.x, qy[o] = me->init_oriQ.y, qz[o] = me->init_oriQ.z;
            fam[o] = (uint8_t)me->family_code;
            inert[o] = (uint16_t)(ts.size() + ext.size() + mI);
            mass.push_back(me->mass), moix.push_back(me->MOI.x), moiy.push_back(me->MOI.y), moiz.push_back(me->MOI.z);
            for (auto& f : me->faces) {
                const float3 a = me->vertices.at(f[0]), b = me->vertices.at(f[1]), c = me->vertices.at(f[2]);
                triOwner.push_back((uint32_t)o), triMat.push_back((uint16_t)me->mat->load_order);
                t1.insert(t1.end(), {a.x, a.y, a.z}), t2.insert(t2.end(), {b.x, b.y, b.z}), t3.insert(t3.end(), {c.x, c.y, c.z});
            }
        }
        // materials
        const size_t nM = std::max<size_t>(1, m_materials.size());
        auto prop = [&](const std::string& name) {
            std::vector<float> v(nM, 0.f);
            for (size_t i = 0; i < m_materials.size(); i++) {
                auto it = m_materials[i]->mat_prop.find(name);
                if (it != m_materials[i]->mat_prop.end())
                    v[i] = it->second;
            }
            return v;
        };
        auto pair = [&](const std::string& name) {
            const std::vector<float> v = prop(name);
            std::vector<float> M(nM * nM);
            for (size_t a = 0; a < nM; a++)
                for (size_t b = 0; b < nM; b++)
                    M[a * nM + b] = (a == b) ? v[a] : (v[a] + v[b]) / 2.f;
            auto it = m_pair_overrides.find(name);
            if (it != m_pair_overrides.end())
                for (auto& kv : it->second)
                    M[kv.first.first * nM + kv.first.second] = kv.second;
            return M;
        };
        const std::vector<float> E = prop("E"), nu = prop("nu"), CoR = pair("CoR"), mu = pair("mu"), Crr = pair("Crr");

        DemeParams& p = m_p;
        p = DemeParams{};
        p.nvXp2 = nv[0], p.nvYp2 = nv[1], p.nvZp2 = nv[2];
        p.nbX = nb[0], p.nbY = nb[1], p.nbZ = nb[2];
        p.l = l, p.voxelSize = voxel, p.binSize = bin;
        p.LBFX = lbf.x, p.LBFY = lbf.y, p.LBFZ = lbf.z;
        p.Gx = m_G.x, p.Gy = m_G.y, p.Gz = m_G.z;
        p.h = m_h;
        p.approxMaxVel = m_max_vel, p.expSafetyMulti = m_safety_multi, p.expSafetyAdder = m_safety_adder;
        p.integrator = m_integrator == TIME_INTEGRATOR::FORWARD_EULER ? DEME_INTEGRATOR_FORWARD_EULER
                       : m_integrator == TIME_INTEGRATOR::CENTERED_DIFFERENCE ? DEME_INTEGRATOR_CENTERED_DIFFERENCE
                                                                               : DEME_INTEGRATOR_EXTENDED_TAYLOR;
        p.forceModel = m_force_model->type == FORCE_MODEL::HERTZIAN ? DEME_FORCE_HERTZIAN
                       : m_force_model->type == FORCE_MODEL::HERTZIAN_FRICTIONLESS ? DEME_FORCE_HERTZIAN_FRICTIONLESS
                                                                                     : DEME_FORCE_CUSTOM;
        p.nContactWildcards = (uint32_t)m_force_model->contact_wildcards.size();
        p.cdUpdateFreq = m_cd_freq;
        p.errOutBinSphNum = m_max_sph_in_bin;
        p.errOutVel = m_err_vel;

        if (m_reupload_state) {
            const DemeOwnerState& st = *m_reupload_state;
            for (size_t o = 0; o < m_reupload_n; o++) {
                const size_t d = m_reupload_dst(o, nC);
                if (d >= nO)
                    continue;
                vid[d] = st.voxelID[o], lx[d] = st.locX[o], ly[d] = st.locY[o], lz[d] = st.locZ[o];
                qw[d] = st.oriQw[o], qx[d] = st.oriQx[o], qy[d] = st.oriQy[o], qz[d] = st.oriQz[o];
                vx[d] = st.vX[o], vy[d] = st.vY[o], vz[d] = st.vZ[o];
                wx[d] = st.omgBarX[o], wy[d] = st.omgBarY[o], wz[d] = st.omgBarZ[o];
                fam[d] = st.familyID[o];
            }
            m_reupload_state = nullptr;
        }
        DemeScene s{};
        s.nOwners = (uint32_t)nO, s.nOwnerClumps = (uint32_t)nC, s.nSpheres = (uint32_t)sphOwner.size();
        s.nAnal = (uint32_t)objType.size(), s.nTri = (uint32_t)triOwner.size(), s.nMat = (uint32_t)nM;
        s.nComp = (uint32_t)Radii.size(), s.nMassProps = (uint32_t)mass.size();
        s.voxelID = vid.data(), s.locX = lx.data(), s.locY = ly.data(), s.locZ = lz.data();
        s.oriQw = qw.data(), s.oriQx = qx.data(), s.oriQy = qy.data(), s.oriQz = qz.data();
        s.vX = vx.data(), s.vY = vy.data(), s.vZ = vz.data(), s.omgBarX = wx.data(), s.omgBarY = wy.data(), s.omgBarZ = wz.data();
        s.familyID = fam.data(), s.inertiaPropOffsets = inert.data();
        s.ownerClumpBody = sphOwner.data(), s.clumpComponentOffset = sphComp.data(), s.sphereMaterialOffset = sphMat.data();
        s.Radii = Radii.data(), s.CDRelPosX = rx.data(), s.CDRelPosY = ry.data(), s.CDRelPosZ = rz.data();
        s.MassProperties = mass.data(), s.moiX = moix.data(), s.moiY = moiy.data(), s.moiZ = moiz.data();
        s.objType = objType.data(), s.objOwner = objOwner.data(), s.objNormal = objN.data(), s.objMaterial = objMat.data();
        s.objRelPosX = opx.data(), s.objRelPosY = opy.data(), s.objRelPosZ = opz.data();
        s.objRotX = orx.data(), s.objRotY = ory.data(), s.objRotZ = orz.data();
        s.objSize1 = os1.data(), s.objSize2 = os2.data(), s.objSize3 = os3.data(), s.objMass = om.data();
        s.E = E.data(), s.nu = nu.data(), s.CoR = CoR.data(), s.mu = mu.data(), s.Crr = Crr.data();
        s.familyMasks = m_family_masks, s.familyExtraMarginSize = m_family_extra, s.familyFlags = m_family_flags;
        s.ownerMesh = triOwner.data(), s.triNode1 = t1.data(), s.triNode2 = t2.data(), s.triNode3 = t3.data();
        s.triMaterialOffset = triMat.data();
        if (const char* dump = std::getenv("DEME_DUMP_SCENE"))  // test hook: what this shell hands the engine (tests/test_host_shell.py
            dump_scene(dump, p, s);                              // compares it field by field with model.py's scene)
        if (m_devices.size() == 1 && m_slabs_per_device > 1 && !m_multi) {  // one device cut into slabs: the context opened by the
            deme_ctx_destroy(m_ctx);                                          // constructor gives way to a deme_multi of that device
            m_ctx = nullptr;
            char err[512];
            if (deme_multi_create(m_devices.data(), 1, &m_multi, err, sizeof err) != DEME_OK)
                throw std::runtime_error(std::string("DEMSolver: ") + err);
        }
        if (m_multi) {  // the bed cut along its longest side into devices x slabs-per-device slabs, bin-aligned (csrc/deme_decomp.inc)
            bool freeOwner = false;  // a mesh / analytical body that moves under contact forces: its accelerations are summed across slabs
            for (size_t o = nC; o < nO; o++)
                freeOwner = freeOwner || !(m_family_flags[fam[o]] & (DEME_FAMILY_FIXED | DEME_FAMILY_PRESCRIBED));
            uint32_t built = 0;
            deme_multi_num_slabs(m_multi, &built);
            if (built)  // a scene re-upload (UpdateClumps, ResortClumps): the old slabs go, the new scene is cut anew
                mcheck(deme_multi_reset(m_multi));
            mcheck(deme_multi_build(m_multi, &p, &s, m_slabs_per_device, -1, (double)m_slab_halo, freeOwner ? DEME_DECOMP_SHARED_FREE : 0u, -1,
                                    p.forceModel == DEME_FORCE_HERTZIAN ? 7u : 0u));
            mcheck(deme_multi_set_migration(m_multi, m_migrate_every));
            mcheck(deme_multi_set_rebalance(m_multi, m_rebalance_every));
            if (built && m_initialized) {
                // the slab contexts are new: what the script set on the old ones after Initialize goes onto them again (a single
                // context survives a re-upload and keeps these by itself) -- the controllers' knobs, families' materials
                push_adaptive();
                for (const FamilyMaterial& fm : m_family_material_log)
                    each_ctx([&](deme_ctx* c) { return deme_set_family_material(c, fm.family, fm.material, fm.mesh); });
            }
        } else {
            check(deme_set_params(m_ctx, &p));
            check(deme_upload_scene(m_ctx, &s));
        }
        volumes.resize(mass.size(), 0.f);  // analytical / mesh owners: unused, like the reference (dT.cpp:607-618)
        if (std::any_of(volumes.begin(), volumes.end(), [](float v) { return v != 0.f; }))
            each_ctx([&](deme_ctx* c) { return deme_upload_volumes(c, volumes.data(), volumes.size()); });
        if (m_force_model->type == FORCE_MODEL::CUSTOM) {
            // _materialDefs_ for properties beyond the five built-in ones (APIPrivate.cpp:1877-2026)
            std::set<std::string> extra;
            for (auto& m : m_materials)
                for (auto& kv : m->mat_prop)
                    if (kv.first != "E" && kv.first != "nu" && kv.first != "CoR" && kv.first != "mu" && kv.first != "Crr")
                        extra.insert(kv.first);
            std::ostringstream pre;
            pre.precision(9);
            pre << m_kernel_includes << "\n";  // AddKernelInclude / SetKernelInclude: in front of everything the model declares
            for (auto& name : extra) {
                if (m_force_model->pairwise_props.count(name)) {
                    const std::vector<float> M = pair(name);
                    pre << "__device__ const float " << name << "[][" << nM << "] = {";
                    for (size_t a = 0; a < nM; a++) {
                        pre << (a ? ", {" : "{");
                        for (size_t b = 0; b < nM; b++)
                            pre << (b ? ", " : "") << std::scientific << M[a * nM + b] << "f";
                        pre << "}";
                    }
                    pre << "};\n";
                } else {
                    const std::vector<float> v = prop(name);
                    pre << "__device__ const float " << name << "[] = {";
                    for (size_t a = 0; a < nM; a++)
                        pre << (a ? ", " : "") << std::scientific << v[a] << "f";
                    pre << "};\n";
                }
            }
            pre << m_force_model->prerequisites;
            std::vector<const char*> names;
            for (auto& n : m_force_model->contact_wildcards)
                names.push_back(n.c_str());
            const std::string prereq = pre.str();
            std::vector<const char*> onames, gnames;
            for (auto& n : m_force_model->owner_wildcards)
                onames.push_back(n.c_str());
            for (auto& n : m_force_model->geo_wildcards)
                gnames.push_back(n.c_str());
            each_ctx([&](deme_ctx* c) {
                return deme_compile_force_model_ex(c, m_force_model->code.c_str(), m_force_model->code.size(), names.data(),
                                                   (uint32_t)names.size(), onames.data(), (uint32_t)onames.size(), gnames.data(),
                                                   (uint32_t)gnames.size(), prereq.c_str());
            });
        }
        m_n_clumps = nC, m_n_owners = nO;
        m_state_fresh = false;
        {  // initial owner-wildcard values the batches carry (DEMClumpBatch::AddOwnerWildcard)
            size_t first = 0;
            for (auto& bt : m_batches) {
                for (auto& kv : bt->owner_wildcards) {
                    if (!m_force_model->owner_wildcards.count(kv.first))
                        throw std::runtime_error("Owner wildcard " + kv.first + " is given to a clump batch but the force model does not "
                                                 "declare it (SetPerOwnerWildcards)");
                    const size_t f0 = first;
                    edit_wildcard(0, nO, m_force_model->owner_wildcards, kv.first, [&](std::vector<float>& a) {
                        for (size_t i = 0; i < bt->nClumps; i++)
                            a[f0 + i] = kv.second[i];
                    });
                }
                first += bt->nClumps;
            }
            size_t firstSph = 0;
            for (auto& bt : m_batches) {
                const size_t nSph = bt->GetNumSpheres();
                for (auto& kv : bt->geo_wildcards) {
                    if (!m_force_model->geo_wildcards.count(kv.first))
                        throw std::runtime_error("Geometry wildcard " + kv.first + " is given to a clump batch but the force model does not "
                                                 "declare it (SetPerGeometryWildcards)");
                    const size_t f0 = firstSph;
                    edit_wildcard(1, sphOwner.size(), m_force_model->geo_wildcards, kv.first, [&](std::vector<float>& a) {
                        for (size_t i = 0; i < nSph; i++)
                            a[f0 + i] = kv.second[i];
                    });
                }
                firstSph += nSph;
            }
        }
        compile_prescriptions_and_rules();
        m_keep.sphOwner = sphOwner, m_keep.sphComp = sphComp, m_keep.inert = inert, m_keep.objOwner = objOwner, m_keep.triOwner = triOwner;
        m_keep.Radii = Radii, m_keep.rx = rx, m_keep.ry = ry, m_keep.rz = rz, m_keep.mass = mass;
        m_keep.moix = moix, m_keep.moiy = moiy, m_keep.moiz = moiz;
        for (size_t i = 0; i < m_templates.size(); i++) {
            char nm[32];
            snprintf(nm, sizeof nm, "%04d", (int)i);
            m_keep.templateName[m_templates[i]->mark] = m_templates[i]->m_name.empty() ? std::string(nm) : m_templates[i]->m_name;
        }
        if (m_cnt_out_content & (FORCE | CNT_POINT | NORMAL | TORQUE))
            each_ctx([&](deme_ctx* c) { return deme_set_record_contacts(c, 1); });
        // restart: existing sphere-sphere contacts of the batches (geometry ids are batch-relative, Structs.h:857)
        {
            std::vector<uint32_t> a, b;
            std::vector<float> w;
            const uint32_t nW = p.nContactWildcards;
            size_t sphBase = 0;
            for (auto& bt : m_batches) {
                for (size_t k = 0; k < bt->contact_pairs.size(); k++) {
                    a.push_back((uint32_t)(sphBase + bt->contact_pairs[k].first));
                    b.push_back((uint32_t)(sphBase + bt->contact_pairs[k].second));
                    for (auto& name : m_force_model->contact_wildcards) {
                        auto it = bt->contact_wildcards.find(name);
                        w.push_back(it == bt->contact_wildcards.end() ? 0.f : it->second[k]);
                    }
                }
                for (size_t i = 0; i < bt->nClumps; i++)
                    sphBase += bt->types[i]->nComp;
            }
            if (!a.empty()) {
                const std::vector<uint8_t> ty(a.size(), 1);
                mc_seed_contacts(a.data(), b.data(), ty.data(), nW ? w.data() : nullptr, a.size());
            }
        }
    }
};


/// A named reduction over the spheres / owners, evaluated on the device (deme_inspect).  Quantities: clump_max_z,
/// clump_min_z, clump_max_absv, clump_mass, clump_volume, max_absv, clump_kinetic_energy, absv (AuxClasses.cpp:94-170);
/// optionally limited to a region given as code (AuxClasses.cpp:205-223), compiled at the first GetValue like the
/// reference's lazy Initialize.
class DEMInspector {
  public:
    DEMInspector(DEMSolver* sys, const std::string& quantity, const std::string& region = "") : m_sys(sys), m_region_code(region) {
        static const std::map<std::string, uint32_t> codes = {{"clump_max_z", DEME_INSPECT_CLUMP_MAX_Z},
                                                              {"clump_min_z", DEME_INSPECT_CLUMP_MIN_Z},
                                                              {"clump_max_absv", DEME_INSPECT_CLUMP_MAX_ABSV},
                                                              {"clump_mass", DEME_INSPECT_CLUMP_MASS},
                                                              {"max_absv", DEME_INSPECT_MAX_ABSV},
                                                              {"clump_kinetic_energy", DEME_INSPECT_CLUMP_KINETIC_ENERGY},
                                                              {"absv", DEME_INSPECT_ABSV},
                                                              {"clump_volume", DEME_INSPECT_CLUMP_VOLUME}};
        auto it = codes.find(quantity);
        if (it == codes.end())
            throw std::runtime_error(quantity + " is not a known query type.");
        m_code = it->second;
    }
    float GetValue() {
        if (!m_sys->decomposed() && m_region < 0 && m_region_code.find_first_not_of(" \t\n") != std::string::npos)
            m_sys->check(deme_compile_region(m_sys->m_ctx, m_region_code.c_str(), &m_region));
        float v = 0;
        if (m_sys->decomposed()) {  // every slab reduces over its OWN clumps (ghost copies are left out of inspections): max / min / sum of those
            const bool isMax = m_code == DEME_INSPECT_CLUMP_MAX_Z || m_code == DEME_INSPECT_CLUMP_MAX_ABSV || m_code == DEME_INSPECT_MAX_ABSV;
            const bool isMin = m_code == DEME_INSPECT_CLUMP_MIN_Z;
            if (m_code == DEME_INSPECT_ABSV)  // (one value per owner, not a reduction: GetValues)
                throw std::runtime_error("DEMInspector(\"absv\") has one value per owner: use GetValues()");
            bool first = true;
            m_sys->each_ctx([&](deme_ctx* c) {
                float w = 0;
                int reg = -1;
                if (m_region_code.find_first_not_of(" \t\n") != std::string::npos)
                    if (int rc = deme_compile_region(c, m_region_code.c_str(), &reg))  // (cached by source inside the library)
                        return rc;
                if (int rc = deme_inspect_region(c, m_code, reg, &w))
                    return rc;
                v = first ? w : (isMax ? std::max(v, w) : isMin ? std::min(v, w) : v + w);
                first = false;
                return (int)DEME_OK;
            });
            return v;
        }
        m_sys->check(deme_inspect_region(m_sys->m_ctx, m_code, m_region, &v));
        return v;
    }
    std::vector<float> GetValues() {
        std::vector<float> v(m_code <= DEME_INSPECT_CLUMP_MAX_ABSV ? m_sys->m_keep.sphOwner.size() : m_sys->m_n_owners);
        if (m_sys->decomposed())
            m_sys->mcheck(deme_multi_inspect_values(m_sys->m_multi, m_code, v.data(), v.size()));
        else
            m_sys->check(deme_inspect_values(m_sys->m_ctx, m_code, v.data(), v.size()));
        return v;
    }

  private:
    DEMSolver* m_sys;
    uint32_t m_code = 0;
    std::string m_region_code;
    int m_region = -1;
};

/// Access to the state of the owners of one loaded object (a batch of clumps, an analytical object, a mesh):
/// getters read the current device state, setters write it back (DEMTracker, AuxClasses.h:93-420).
class DEMTracker {
  public:
    /// kind: 0 clump batch, 1 analytical object, 2 mesh; index: load order.  Owner ids are resolved when used, so a tracker
    /// may be created before Initialize() as in the reference's demos.
    DEMTracker(DEMSolver* sys, int kind, size_t index, size_t n) : m_sys(sys), m_kind(kind), m_index(index), m_n(n) {}
    bodyID_t GetOwnerID(size_t offset = 0) const { return (bodyID_t)(m_sys->tracker_first_owner(m_kind, m_index) + in_range(offset)); }
    size_t GetNumOwners() const { return m_n; }
    std::vector<bodyID_t> GetOwnerIDs() const {
        std::vector<bodyID_t> ids(m_n);
        for (size_t i = 0; i < m_n; i++)
            ids[i] = GetOwnerID(i);
        return ids;
    }
    std::vector<unsigned int> GetFamilies() {
        std::vector<unsigned int> f(m_n);
        for (size_t i = 0; i < m_n; i++)
            f[i] = GetFamily(i);
        return f;
    }
    float3 Pos(size_t offset = 0) { return m_sys->GetOwnerPosition(GetOwnerID(offset)); }
    float3 Vel(size_t offset = 0) { return m_sys->GetOwnerVelocity(GetOwnerID(offset)); }
    float3 AngVelLocal(size_t offset = 0) { return column3(offset, 2); }
    float4 OriQ(size_t offset = 0) {
        const DEMSolver::Snapshot sn = m_sys->snapshot(false);
        return sn.q[GetOwnerID(offset)];
    }
    float3 ContactAcc(size_t offset = 0) { return column3(offset, 3); }
    float3 ContactAngAccLocal(size_t offset = 0) { return column3(offset, 4); }
    std::vector<float3> Positions() {
        std::vector<float3> out(m_n);
        for (size_t i = 0; i < m_n; i++)
            out[i] = Pos(i);
        return out;
    }
    void SetPos(float3 pos, size_t offset = 0) { m_sys->set_owner(GetOwnerID(offset), &pos, nullptr, nullptr, nullptr); }
    void SetVel(float3 vel, size_t offset = 0) { m_sys->set_owner(GetOwnerID(offset), nullptr, &vel, nullptr, nullptr); }
    void SetAngVel(float3 w, size_t offset = 0) { m_sys->set_owner(GetOwnerID(offset), nullptr, nullptr, &w, nullptr); }
    void SetOriQ(float4 q, size_t offset = 0) { m_sys->set_owner(GetOwnerID(offset), nullptr, nullptr, nullptr, &q); }
    // the rest of the getters / setters of AuxClasses.h:117-333 (std::vector<float> twins of the float3 getters included)
    float3 AngVelGlobal(size_t offset = 0) {
        float3 w = AngVelLocal(offset);
        DEMSolver::rotate(w, OriQ(offset));
        return w;
    }
    float3 ContactAngAccGlobal(size_t offset = 0) {
        float3 a = ContactAngAccLocal(offset);
        DEMSolver::rotate(a, OriQ(offset));
        return a;
    }
    float Mass(size_t offset = 0) { return m_sys->GetOwnerMass(GetOwnerID(offset)); }
    float3 MOI(size_t offset = 0) { return m_sys->GetOwnerMOI(GetOwnerID(offset)); }
    unsigned int GetFamily(size_t offset = 0) { return m_sys->GetOwnerFamily(GetOwnerID(offset)); }
    void SetFamily(unsigned int fam_num) { m_sys->SetOwnerFamily(GetOwnerID(0), fam_num, m_n); }
    void SetFamily(unsigned int fam_num, size_t offset) { m_sys->SetOwnerFamily(GetOwnerID(offset), fam_num, 1); }
    static std::vector<float> vec3(float3 v) { return {v.x, v.y, v.z}; }
    std::vector<float> GetPos(size_t offset = 0) { return vec3(Pos(offset)); }
    std::vector<float> GetVel(size_t offset = 0) { return vec3(Vel(offset)); }
    std::vector<float> GetAngVelLocal(size_t offset = 0) { return vec3(AngVelLocal(offset)); }
    std::vector<float> GetAngVelGlobal(size_t offset = 0) { return vec3(AngVelGlobal(offset)); }
    std::vector<float> GetContactAcc(size_t offset = 0) { return vec3(ContactAcc(offset)); }
    std::vector<float> GetContactAngAccLocal(size_t offset = 0) { return vec3(ContactAngAccLocal(offset)); }
    std::vector<float> GetContactAngAccGlobal(size_t offset = 0) { return vec3(ContactAngAccGlobal(offset)); }
    std::vector<float> GetMOI(size_t offset = 0) { return vec3(MOI(offset)); }
    std::vector<float> GetOriQ(size_t offset = 0) {
        const float4 q = OriQ(offset);
        return {q.x, q.y, q.z, q.w};
    }
    void SetPos(const std::vector<float3>& pos) {
        for (size_t k = 0; k < pos.size() && k < m_n; k++)
            SetPos(pos[k], k);
    }
    void SetVel(const std::vector<float3>& vel) {
        for (size_t k = 0; k < vel.size() && k < m_n; k++)
            SetVel(vel[k], k);
    }
    void SetAngVel(const std::vector<float3>& w) {
        for (size_t k = 0; k < w.size() && k < m_n; k++)
            SetAngVel(w[k], k);
    }
    void SetOriQ(const std::vector<float4>& q) {
        for (size_t k = 0; k < q.size() && k < m_n; k++)
            SetOriQ(q[k], k);
    }
    std::vector<float> GetOwnerWildcardValues(const std::string& name) {
        const std::vector<float> all = m_sys->GetAllOwnerWildcardValue(name);
        const size_t o0 = GetOwnerID(0);
        return std::vector<float>(all.begin() + o0, all.begin() + o0 + m_n);
    }
    void SetOwnerWildcardValue(const std::string& name, float wc, size_t offset = 0) {
        m_sys->SetOwnerWildcardValue(GetOwnerID(offset), name, wc, 1);
    }
    void SetOwnerWildcardValues(const std::string& name, const std::vector<float>& wc) {
        m_sys->SetOwnerWildcardValue(GetOwnerID(0), name, wc);
    }
    /// geometry wildcards of the tracked object's spheres / analytical components / triangles, in geometry order
    /// (AuxClasses.h:235, 329-333)
    std::vector<float> GetGeometryWildcardValues(const std::string& name) {
        const std::vector<uint32_t>& own = geo_owner_list();
        const std::vector<float> all = m_sys->get_wildcard(geo_kind(), own.size(), m_sys->m_force_model->geo_wildcards, name);
        std::vector<float> out;
        const size_t o0 = GetOwnerID(0);
        for (size_t i = 0; i < own.size(); i++)
            if (own[i] >= o0 && own[i] < o0 + m_n)
                out.push_back(all[i]);
        return out;
    }
    void SetGeometryWildcardValues(const std::string& name, const std::vector<float>& wc) {
        const std::vector<uint32_t>& own = geo_owner_list();
        const size_t o0 = GetOwnerID(0);
        m_sys->edit_wildcard(geo_kind(), own.size(), m_sys->m_force_model->geo_wildcards, name, [&](std::vector<float>& a) {
            size_t k = 0;
            for (size_t i = 0; i < own.size() && k < wc.size(); i++)
                if (own[i] >= o0 && own[i] < o0 + m_n)
                    a[i] = wc[k++];
        });
    }
    void SetGeometryWildcardValue(const std::string& name, float wc, size_t geo_offset = 0) {
        std::vector<float> v = GetGeometryWildcardValues(name);
        v.at(geo_offset) = wc;
        SetGeometryWildcardValues(name, v);
    }
    /// AddAcc / AddAngAcc (AuxClasses.h:264-274): extra acceleration for the coming step only (co-simulation hand-over)
    void AddAcc(float3 acc, size_t offset = 0) {
        const float v[3] = {acc.x, acc.y, acc.z};
        m_sys->add_owner_acc(GetOwnerID(offset), 1, v, nullptr);
    }
    void AddAcc(const std::vector<float3>& acc) { add_many(acc, true); }
    void AddAngAcc(float3 angAcc, size_t offset = 0) {
        const float v[3] = {angAcc.x, angAcc.y, angAcc.z};
        m_sys->add_owner_acc(GetOwnerID(offset), 1, nullptr, v);
    }
    void AddAngAcc(const std::vector<float3>& angAcc) { add_many(angAcc, false); }
    /// every contact force on one tracked owner / on all of them (AuxClasses.h:335-410)
    size_t GetContactForces(std::vector<float3>& points, std::vector<float3>& forces, size_t offset = 0) {
        return m_sys->GetOwnerContactForces({GetOwnerID(offset)}, points, forces);
    }
    size_t GetContactForcesForAll(std::vector<float3>& points, std::vector<float3>& forces) {
        return m_sys->GetOwnerContactForces(all_owner_ids(), points, forces);
    }
    size_t GetContactForcesAndGlobalTorque(std::vector<float3>& points, std::vector<float3>& forces, std::vector<float3>& torques,
                                           size_t offset = 0) {
        return m_sys->GetOwnerContactForces({GetOwnerID(offset)}, points, forces, &torques, false);
    }
    size_t GetContactForcesAndGlobalTorqueForAll(std::vector<float3>& points, std::vector<float3>& forces, std::vector<float3>& torques) {
        return m_sys->GetOwnerContactForces(all_owner_ids(), points, forces, &torques, false);
    }
    size_t GetContactForcesAndLocalTorque(std::vector<float3>& points, std::vector<float3>& forces, std::vector<float3>& torques,
                                          size_t offset = 0) {
        return m_sys->GetOwnerContactForces({GetOwnerID(offset)}, points, forces, &torques, true);
    }
    size_t GetContactForcesAndLocalTorqueForAll(std::vector<float3>& points, std::vector<float3>& forces, std::vector<float3>& torques) {
        return m_sys->GetOwnerContactForces(all_owner_ids(), points, forces, &torques, true);
    }
    /// UpdateMesh (AuxClasses.h): replace the owner-local node coordinates of a tracked mesh (deformable meshes)
    void UpdateMesh(const std::vector<float3>& new_nodes) {
        if (m_kind != 2)
            throw std::runtime_error("UpdateMesh needs a tracker of a mesh");
        m_sys->update_mesh_nodes(m_index, new_nodes);
    }
    /// UpdateMeshByIncrement (AuxClasses.h:306): add a deformation to every node
    void UpdateMeshByIncrement(const std::vector<float3>& deformation) {
        std::vector<float3> nodes = GetMesh()->vertices;
        if (deformation.size() != nodes.size())
            throw std::runtime_error("UpdateMeshByIncrement: the deformation count does not match the mesh");
        for (size_t i = 0; i < nodes.size(); i++)
            nodes[i] = nodes[i] + deformation[i];
        UpdateMesh(nodes);
    }
    /// the mesh a mesh tracker follows (AuxClasses.h:310)
    std::shared_ptr<DEMMeshConnected>& GetMesh() {
        if (m_kind != 2)
            throw std::runtime_error("GetMesh needs a tracker of a mesh");
        return m_sys->m_meshes.at(m_index);
    }
    /// current node positions in the global frame: owner-local nodes rotated by the owner's orientation, moved to its position
    std::vector<float3> GetMeshNodesGlobal() {
        std::vector<float3> nodes = GetMesh()->vertices;
        const float3 pos = Pos();
        const float4 q = OriQ();
        for (auto& n : nodes) {
            DEMSolver::rotate(n, q);
            n = n + pos;
        }
        return nodes;
    }

  private:
    DEMSolver* m_sys;
    int m_kind;
    size_t m_index, m_n;
    uint32_t geo_kind() const { return m_kind == 0 ? 1u : (m_kind == 1 ? 3u : 2u); }  // spheres / analytical / triangles
    const std::vector<uint32_t>& geo_owner_list() const {
        return m_kind == 0 ? m_sys->m_keep.sphOwner : (m_kind == 1 ? m_sys->m_keep.objOwner : m_sys->m_keep.triOwner);
    }
    void add_many(const std::vector<float3>& v, bool linear) {
        if (v.size() != m_n)
            throw std::runtime_error("AddAcc / AddAngAcc: one value per tracked owner is needed");
        std::vector<float> flat(3 * m_n);
        for (size_t k = 0; k < m_n; k++)
            flat[3 * k] = v[k].x, flat[3 * k + 1] = v[k].y, flat[3 * k + 2] = v[k].z;
        m_sys->add_owner_acc(GetOwnerID(0), (uint32_t)m_n, linear ? flat.data() : nullptr,
                                        linear ? nullptr : flat.data());
    }
    std::vector<bodyID_t> all_owner_ids() const {
        std::vector<bodyID_t> ids(m_n);
        for (size_t k = 0; k < m_n; k++)
            ids[k] = GetOwnerID(k);
        return ids;
    }
    size_t in_range(size_t offset) const {
        if (offset >= m_n)
            throw std::runtime_error("tracker offset is out of range");
        return offset;
    }
    float3 column3(size_t offset, int which) {
        const DEMSolver::Snapshot sn = m_sys->snapshot(false);
        const size_t o = GetOwnerID(offset);
        return which == 2 ? sn.w[o] : which == 3 ? sn.a[o] : sn.al[o];
    }
};

inline std::shared_ptr<DEMInspector> DEMSolver::CreateInspector(const std::string& quantity) {
    return std::make_shared<DEMInspector>(this, quantity);
}
inline std::shared_ptr<DEMInspector> DEMSolver::CreateInspector(const std::string& quantity, const std::string& region) {
    return std::make_shared<DEMInspector>(this, quantity, region);
}
inline std::shared_ptr<DEMTracker> DEMSolver::Track(const std::shared_ptr<DEMClumpBatch>& batch) {
    for (size_t i = 0; i < m_batches.size(); i++)
        if (m_batches[i] == batch)
            return std::make_shared<DEMTracker>(this, 0, i, batch->nClumps);
    throw std::runtime_error("Track: this batch was not loaded into this solver");
}
inline std::shared_ptr<DEMTracker> DEMSolver::Track(const std::shared_ptr<DEMExternObj>& obj) {
    for (size_t e = 0; e < m_ext.size(); e++)
        if (m_ext[e] == obj)
            return std::make_shared<DEMTracker>(this, 1, e, 1);
    throw std::runtime_error("Track: this object was not loaded into this solver");
}
inline std::shared_ptr<DEMTracker> DEMSolver::Track(const std::shared_ptr<DEMMeshConnected>& mesh) {
    for (size_t m = 0; m < m_meshes.size(); m++)
        if (m_meshes[m] == mesh)
            return std::make_shared<DEMTracker>(this, 2, m, 1);
    throw std::runtime_error("Track: this mesh was not loaded into this solver");
}

}  // namespace deme
