/*
 * ref_glue.cpp -- thin extern "C" wrappers around the REFERENCE's own
 * __host__ __device__ helpers and force-model fragments.
 *
 * TEST INFRASTRUCTURE ONLY.  This file contains no reference code: it
 * #includes the reference sources where they lie (/root/reference/src, given
 * with -I by oracle/Makefile) and is compiled with plain g++ against the CUDA
 * toolkit headers that ship inside this image's triton wheel
 * (.../triton/backends/nvidia/include/cuda_runtime.h) -- real headers, no
 * stand-ins.  Output goes to oracle/_ref/libdeme_ref.so (git-ignored; travels to
 * the GPU box as a prebuilt file).  Used to (1) validate oracle/deme_oracle.cpp
 * function by function and (2) generate tests/golden/*.npz
 * (tests/golden/make_golden.py).
 *
 * What is NOT built this way: every __global__ kernel body and every
 * __device__-only function (they contain _placeholder_ tokens that the reference
 * substitutes textually at run time, i.e. generated code), and
 * kernel/DEMCollisionKernels.cu (device-only round-up intrinsics).
 */
#include <cstddef>
#include <cstdint>

#include "kernel/DEMHelperKernels.cuh"
// check_TriangleBoxOverlap (SAT triangle / axis-aligned box test used by the mesh broad phase): the file is
// __device__-only, which the genuine CUDA headers define away for a host compiler
#include "kernel/DEMTriangleBoxIntersect.cu"

extern "C" {

void ref_decode(size_t n, const uint64_t* id, const uint16_t* sx, const uint16_t* sy, const uint16_t* sz,
                unsigned nvXp2, unsigned nvYp2, double voxelSize, double l, double* X, double* Y, double* Z) {
    for (size_t i = 0; i < n; i++)
        voxelIDToPosition<double, deme::voxelID_t, deme::subVoxelPos_t>(X[i], Y[i], Z[i], id[i], sx[i], sy[i], sz[i],
                                                                        (unsigned char)nvXp2, (unsigned char)nvYp2,
                                                                        voxelSize, l);
}
void ref_encode(size_t n, const double* X, const double* Y, const double* Z, unsigned nvXp2, unsigned nvYp2,
                double voxelSize, double l, uint64_t* id, uint16_t* sx, uint16_t* sy, uint16_t* sz) {
    for (size_t i = 0; i < n; i++) {
        deme::voxelID_t v;
        positionToVoxelID<deme::voxelID_t, deme::subVoxelPos_t, double>(v, sx[i], sy[i], sz[i], X[i], Y[i], Z[i],
                                                                        (unsigned char)nvXp2, (unsigned char)nvYp2,
                                                                        voxelSize, l);
        id[i] = v;
    }
}
void ref_rotate(size_t n, float* x, float* y, float* z, const float* qw, const float* qx, const float* qy,
                const float* qz) {
    for (size_t i = 0; i < n; i++)
        applyOriQToVector3<float, deme::oriQ_t>(x[i], y[i], z[i], qw[i], qx[i], qy[i], qz[i]);
}
void ref_rotate_d(size_t n, double* x, double* y, double* z, const float* qw, const float* qx, const float* qy,
                  const float* qz) {
    for (size_t i = 0; i < n; i++)
        applyOriQToVector3(x[i], y[i], z[i], qw[i], qx[i], qy[i], qz[i]);
}
void ref_hamilton(size_t n, const float* q1, const float* q2, float* out) {
    for (size_t i = 0; i < n; i++)
        HamiltonProduct(out[4 * i], out[4 * i + 1], out[4 * i + 2], out[4 * i + 3], q1[4 * i], q1[4 * i + 1],
                        q1[4 * i + 2], q1[4 * i + 3], q2[4 * i], q2[4 * i + 1], q2[4 * i + 2], q2[4 * i + 3]);
}
void ref_mask_pair(size_t n, const uint32_t* i, const uint32_t* j, uint32_t* out) {
    for (size_t k = 0; k < n; k++)
        out[k] = locateMaskPair<unsigned int>(i[k], j[k]);
}
void ref_recover_pair(size_t n, const uint32_t* ind, const uint32_t* cnt, uint32_t* oi, uint32_t* oj) {
    for (size_t k = 0; k < n; k++)
        recoverCntPair<unsigned int>(oi[k], oj[k], ind[k], cnt[k]);
}
void ref_point_bin(size_t n, const double* X, const double* Y, const double* Z, double binSize, uint32_t nbX,
                   uint32_t nbY, uint32_t* out) {
    for (size_t i = 0; i < n; i++)
        out[i] = getPointBinID<deme::binID_t>(X[i], Y[i], Z[i], binSize, nbX, nbY);
}
void ref_bin_from_indices(size_t n, const uint32_t* x, const uint32_t* y, const uint32_t* z, uint32_t nbX,
                          uint32_t nbY, uint32_t nbZ, uint32_t* out) {
    for (size_t i = 0; i < n; i++)
        out[i] = binIDFrom3Indices<deme::binID_t>(x[i], y[i], z[i], nbX, nbY, nbZ);
}
void ref_spheres_overlap(size_t n, const double* A, const double* rA, const double* B, const double* rB,
                         uint8_t* type, double* CP, float* nrm, double* depth) {
    for (size_t i = 0; i < n; i++)
        type[i] = checkSpheresOverlap<double, float>(A[3 * i], A[3 * i + 1], A[3 * i + 2], rA[i], B[3 * i],
                                                     B[3 * i + 1], B[3 * i + 2], rB[i], CP[3 * i], CP[3 * i + 1],
                                                     CP[3 * i + 2], nrm[3 * i], nrm[3 * i + 1], nrm[3 * i + 2],
                                                     depth[i]);
}
void ref_sphere_entity(size_t n, const double* A, const float* radA, const uint8_t* typeB, const double* B,
                       const float* dirB, const float* size1, const float* normal_sign, const float* beta,
                       uint8_t* type, double* CP, float* nrm, double* depth) {
    for (size_t i = 0; i < n; i++) {
        double3 cp = make_double3(0, 0, 0);
        float3 nr = make_float3(0, 0, 0);
        double d = 0;
        type[i] = checkSphereEntityOverlap<double3, float, double>(
            make_double3(A[3 * i], A[3 * i + 1], A[3 * i + 2]), radA[i], typeB[i],
            make_double3(B[3 * i], B[3 * i + 1], B[3 * i + 2]), make_float3(dirB[3 * i], dirB[3 * i + 1], dirB[3 * i + 2]),
            size1[i], 0.f, 0.f, normal_sign[i], beta[i], cp, nr, d);
        CP[3 * i] = cp.x, CP[3 * i + 1] = cp.y, CP[3 * i + 2] = cp.z;
        nrm[3 * i] = nr.x, nrm[3 * i + 1] = nr.y, nrm[3 * i + 2] = nr.z;
        depth[i] = d;
    }
}
void ref_mat_proxy(size_t n, const float* Y1, const float* nu1, const float* Y2, const float* nu2, float* E,
                   float* G) {
    for (size_t i = 0; i < n; i++)
        matProxy2ContactParam<float>(E[i], G[i], Y1[i], nu1[i], Y2[i], nu2[i]);
}

// Force-model fragments, spliced exactly the way the reference's kernel does
// (kernel/DEMCalcForceKernels.cu:233-251): the ingredient variables are declared
// with the reference's names and types, then the fragment file is included as
// the statement block.  Material tables are 1x1 / 2x2 so that "bodyAMatType"
// indexing works: index 0 = A's material, 1 = B's.
#define REF_FORCE_NF 39
static float E[2], nu[2];
static float CoR[2][2], mu[2][2], Crr[2][2];

static void ref_force_one(int model, double overlapDepth, const float* f, float mu_v, float Crr_v, float* hist,
                          float* out) {
    E[0] = f[34], nu[0] = f[35], E[1] = f[36], nu[1] = f[37];
    for (int a = 0; a < 2; a++)
        for (int b = 0; b < 2; b++) {
            CoR[a][b] = f[38];
            mu[a][b] = mu_v;
            Crr[a][b] = Crr_v;
        }
    float3 B2A = make_float3(f[0], f[1], f[2]);
    float AOwnerMass = f[3], BOwnerMass = f[4], ARadius = f[5], BRadius = f[6];
    float4 AOriQ, BOriQ;
    AOriQ.w = f[7], AOriQ.x = f[8], AOriQ.y = f[9], AOriQ.z = f[10];
    BOriQ.w = f[11], BOriQ.x = f[12], BOriQ.y = f[13], BOriQ.z = f[14];
    float3 locCPA = make_float3(f[15], f[16], f[17]);
    float3 locCPB = make_float3(f[18], f[19], f[20]);
    float3 ALinVel = make_float3(f[21], f[22], f[23]);
    float3 BLinVel = make_float3(f[24], f[25], f[26]);
    float3 ARotVel = make_float3(f[27], f[28], f[29]);
    float3 BRotVel = make_float3(f[30], f[31], f[32]);
    float ts = f[33];
    deme::materialsOffset_t bodyAMatType = 0, bodyBMatType = 1;
    float3 force = make_float3(0, 0, 0);
    float3 torque_only_force = make_float3(0, 0, 0);
    if (model == 0) {
        float delta_tan_x = hist[0], delta_tan_y = hist[1], delta_tan_z = hist[2], delta_time = hist[3];
        {
#include "kernel/DEMCustomizablePolicies/FullHertzianForceModel.cu"
        }
        hist[0] = delta_tan_x, hist[1] = delta_tan_y, hist[2] = delta_tan_z, hist[3] = delta_time;
    } else {
        {
#include "kernel/DEMCustomizablePolicies/FrictionlessHertzianForceModel.cu"
        }
    }
    out[0] = force.x, out[1] = force.y, out[2] = force.z;
    out[3] = torque_only_force.x, out[4] = torque_only_force.y, out[5] = torque_only_force.z;
}

void ref_force(size_t n, int model, const double* depth, const float* fin, const float* mu_in, const float* Crr_in,
               float* hist, float* out) {
    for (size_t i = 0; i < n; i++)
        ref_force_one(model, depth[i], fin + i * REF_FORCE_NF, mu_in[i], Crr_in[i], hist + 4 * i, out + 6 * i);
}

// Integrator velocity pass-on fragments (kernel/DEMCustomizablePolicies/IntegrationVelPassOn*.cu)
void ref_vel_pass_on(size_t n, int scheme, const float* old_v_in, const float* v_update_in, float* v_out) {
    for (size_t i = 0; i < n; i++) {
        float3 old_v = make_float3(old_v_in[3 * i], old_v_in[3 * i + 1], old_v_in[3 * i + 2]);
        float3 v_update = make_float3(v_update_in[3 * i], v_update_in[3 * i + 1], v_update_in[3 * i + 2]);
        float3 old_omgBar = old_v, omgBar_update = v_update;
        float3 v, omgBar;
        if (scheme == 0) {
#include "kernel/DEMCustomizablePolicies/IntegrationVelPassOnForwardEuler.cu"
        } else if (scheme == 1) {
#include "kernel/DEMCustomizablePolicies/IntegrationVelPassOnCenteredDiff.cu"
        } else {
#include "kernel/DEMCustomizablePolicies/IntegrationVelPassOnExtendedTaylor.cu"
        }
        (void)omgBar;
        v_out[3 * i] = v.x, v_out[3 * i + 1] = v.y, v_out[3 * i + 2] = v.z;
    }
}

// G7: check_TriangleBoxOverlap (DEMTriangleBoxIntersect.cu:295), cubic box of half-size half[i]
void ref_tri_box(size_t n, const float* center, const float* half, const float* A, const float* B, const float* C,
                 uint8_t* out) {
    for (size_t i = 0; i < n; i++) {
        float bc[3] = {center[3 * i], center[3 * i + 1], center[3 * i + 2]};
        float bh[3] = {half[i], half[i], half[i]};
        out[i] = check_TriangleBoxOverlap(bc, bh, make_float3(A[3 * i], A[3 * i + 1], A[3 * i + 2]),
                                          make_float3(B[3 * i], B[3 * i + 1], B[3 * i + 2]),
                                          make_float3(C[3 * i], C[3 * i + 1], C[3 * i + 2]));
    }
}

// boundingBoxIntersectBin (DEMHelperKernels.cuh:528): bin range of a triangle's enlarged bounding box
void ref_tri_bbox(size_t n, const float* A, const float* B, const float* C, double binSize, uint32_t nbX, uint32_t nbY,
                  uint32_t nbZ, int32_t* L, int32_t* U) {
    deme::DEMSimParams sp{};
    sp.binSize = binSize;
    sp.nbX = nbX;
    sp.nbY = nbY;
    sp.nbZ = nbZ;
    for (size_t i = 0; i < n; i++) {
        deme::binID_t l3[3], u3[3];
        boundingBoxIntersectBin(l3, u3, make_float3(A[3 * i], A[3 * i + 1], A[3 * i + 2]), make_float3(B[3 * i], B[3 * i + 1], B[3 * i + 2]),
                                make_float3(C[3 * i], C[3 * i + 1], C[3 * i + 2]), &sp);
        for (int d = 0; d < 3; d++) {
            L[3 * i + d] = (int32_t)l3[d];
            U[3 * i + d] = (int32_t)u3[d];
        }
    }
}

}  // extern "C"
