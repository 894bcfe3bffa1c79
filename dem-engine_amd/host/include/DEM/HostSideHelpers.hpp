// DEM/HostSideHelpers.hpp -- host-side helper functions the reference's scripts use (src/DEM/HostSideHelpers.hpp): frame
// transforms with quaternions, small vector utilities, value formatting.  Own implementations over the shell's float3 / float4.
#pragma once
#include <cmath>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../DEMSolver.h"

namespace deme {

// applyOriQToVector3 (kernel/DEMHelperKernels.cuh:161-173) on the host: rotate (X, Y, Z) by the quaternion (w, x, y, z)
template <typename T1, typename T2>
inline void hostApplyOriQToVector3(T1& X, T1& Y, T1& Z, const T2& Qw, const T2& Qx, const T2& Qy, const T2& Qz) {
    const T1 oX = X, oY = Y, oZ = Z;
    X = ((T2)2.0 * (Qw * Qw + Qx * Qx) - (T2)1.0) * oX + ((T2)2.0 * (Qx * Qy - Qw * Qz)) * oY + ((T2)2.0 * (Qx * Qz + Qw * Qy)) * oZ;
    Y = ((T2)2.0 * (Qx * Qy + Qw * Qz)) * oX + ((T2)2.0 * (Qw * Qw + Qy * Qy) - (T2)1.0) * oY + ((T2)2.0 * (Qy * Qz - Qw * Qx)) * oZ;
    Z = ((T2)2.0 * (Qx * Qz - Qw * Qy)) * oX + ((T2)2.0 * (Qy * Qz + Qw * Qx)) * oY + ((T2)2.0 * (Qw * Qw + Qz * Qz) - (T2)1.0) * oZ;
}
template <typename T1, typename T2>
inline void applyOriQToVector3(T1& X, T1& Y, T1& Z, const T2& Qw, const T2& Qx, const T2& Qy, const T2& Qz) {
    hostApplyOriQToVector3(X, Y, Z, Qw, Qx, Qy, Qz);
}

/// rotate pos by rot_Q, then translate by vec (HostSideHelpers.hpp: applyFrameTransformLocalToGlobal)
template <typename T1, typename T2, typename T3>
inline void applyFrameTransformLocalToGlobal(T1& pos, const T2& vec, const T3& rot_Q) {
    hostApplyOriQToVector3(pos.x, pos.y, pos.z, rot_Q.w, rot_Q.x, rot_Q.y, rot_Q.z);
    pos.x += vec.x;
    pos.y += vec.y;
    pos.z += vec.z;
}
/// translate by -vec, then rotate by the inverse of rot_Q (HostSideHelpers.hpp:609-614)
template <typename T1, typename T2, typename T3>
inline void applyFrameTransformGlobalToLocal(T1& pos, const T2& vec, const T3& rot_Q) {
    pos.x -= vec.x;
    pos.y -= vec.y;
    pos.z -= vec.z;
    hostApplyOriQToVector3(pos.x, pos.y, pos.z, rot_Q.w, -rot_Q.x, -rot_Q.y, -rot_Q.z);
}
inline std::vector<double> FrameTransformLocalToGlobal(const std::vector<double>& pos, const std::vector<double>& vec,
                                                       const std::vector<double>& rot_Q) {
    if (pos.size() != 3 || vec.size() != 3 || rot_Q.size() != 4)
        throw std::runtime_error("FrameTransformLocalToGlobal: pos and vec need 3 elements, rot_Q 4 (x, y, z, w)");
    double x = pos[0], y = pos[1], z = pos[2];
    hostApplyOriQToVector3(x, y, z, rot_Q[3], rot_Q[0], rot_Q[1], rot_Q[2]);
    return {x + vec[0], y + vec[1], z + vec[2]};
}
inline std::vector<double> FrameTransformGlobalToLocal(const std::vector<double>& pos, const std::vector<double>& vec,
                                                       const std::vector<double>& rot_Q) {
    if (pos.size() != 3 || vec.size() != 3 || rot_Q.size() != 4)
        throw std::runtime_error("FrameTransformGlobalToLocal: pos and vec need 3 elements, rot_Q 4 (x, y, z, w)");
    double x = pos[0] - vec[0], y = pos[1] - vec[1], z = pos[2] - vec[2];
    hostApplyOriQToVector3(x, y, z, rot_Q[3], -rot_Q[0], -rot_Q[1], -rot_Q[2]);
    return {x, y, z};
}

/// a unit vector perpendicular to the given one
template <typename T>
inline T findPerpendicular(const T& v) {
    T a = (std::abs(v.x) < 0.9f * length(v)) ? T{1, 0, 0} : T{0, 1, 0};
    return normalize(cross(v, a));
}
/// rotate vec about the unit axis by theta (Rodrigues' formula)
inline float3 Rodrigues(const float3 vec, const float3 axis, const float theta) {
    const float c = std::cos(theta), s = std::sin(theta);
    return vec * c + cross(axis, vec) * s + axis * (dot(axis, vec) * (1.f - c));
}
/// quaternion (x, y, z, w) of a rotation by theta about a unit axis
inline float4 QuatFromAxisAngle(const float3 axis, const float theta) {
    const float s = std::sin(theta / 2);
    return make_float4(axis.x * s, axis.y * s, axis.z * s, std::cos(theta / 2));
}

/// q (x, y, z, w) followed by a rotation of theta about the unit axis (global frame): the Hamilton product rot * q
inline float4 RotateQuat(const float4 q, const float3 axis, const float theta) {
    const float4 r = QuatFromAxisAngle(normalize(axis), theta);
    return make_float4(r.w * q.x + r.x * q.w + r.y * q.z - r.z * q.y, r.w * q.y - r.x * q.z + r.y * q.w + r.z * q.x,
                       r.w * q.z + r.x * q.y - r.y * q.x + r.z * q.w, r.w * q.w - r.x * q.x - r.y * q.y - r.z * q.z);
}
template <typename T>
inline T vector_sum(const std::vector<T>& v) {
    T s = T(0);
    for (const T& e : v)
        s += e;
    return s;
}

inline std::string to_string_with_precision(const double a_value, const unsigned int n = 17) {
    char buf[400];
    std::snprintf(buf, sizeof buf, "%.*f", (int)n, a_value);
    return buf;
}

}  // namespace deme
