/*
 * deme_hip.h -- C-ABI of the MI355X-native DEM hot path (libdeme_hip.so).
 *
 * The reference (projectchrono/DEM-Engine) has no C ABI: its host classes call
 * C++ functions and runtime-compiled CUDA kernels directly.  Each entry point
 * below names the reference interface it replaces (paths relative to the
 * reference tree, src/...).  Plain pointers and sizes only; no C++ or torch
 * types cross this boundary; no exception crosses it (every call returns an
 * int status, 0 = OK, and deme_last_error() gives the message).
 *
 * One context per GPU.  Calls on one context must be serialised by the caller.
 * All work is enqueued on the context's HIP stream (deme_ctx_set_stream lets a
 * caller such as PyTorch hand in its own stream).
 */
#ifndef DEME_HIP_H
#define DEME_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- constants (reference: DEM/Defines.h:74-82, 99-112, 146) ------------- */
#define DEME_NOT_A_CONTACT 0
#define DEME_SPHERE_SPHERE_CONTACT 1
#define DEME_SPHERE_MESH_CONTACT 2
#define DEME_SPHERE_PLANE_CONTACT 11
#define DEME_SPHERE_CYL_CONTACT 13

#define DEME_ANAL_OBJ_TYPE_PLANE 0
#define DEME_ANAL_OBJ_TYPE_PLATE 1
#define DEME_ANAL_OBJ_TYPE_CYL_INF 2

#define DEME_NULL_MAPPING_PARTNER 0xFFFFFFFFu
#define DEME_NULL_BINID 0xFFFFFFFFu
#define DEME_NUM_FAMILIES 256
#define DEME_FAMILY_MASK_ENTRIES 32896 /* 256*257/2, DEM/kT.cpp:609 */
#define DEME_MAX_WILDCARD_NUM 16

#define DEME_INTEGRATOR_FORWARD_EULER 0
#define DEME_INTEGRATOR_CENTERED_DIFFERENCE 1
#define DEME_INTEGRATOR_EXTENDED_TAYLOR 2

#define DEME_FORCE_HERTZIAN 0              /* FullHertzianForceModel.cu: 4 contact wildcards */
#define DEME_FORCE_HERTZIAN_FRICTIONLESS 1 /* FrictionlessHertzianForceModel.cu: 0 wildcards */
#define DEME_FORCE_CUSTOM 2                /* user fragment, see deme_compile_force_model */

/* familyFlags bits: data-driven form of the prescription `switch` the
 * reference JIT-compiles into integrateOwners (DEMIntegrationKernels.cu:26-33,
 * APIPublic.cpp:980-1011 SetFamilyFixed). */
#define DEME_FAMILY_FIXED 1
/* (retired: ghost copies are marked per owner, DemeScene.ownerGhost, and keep their own family) */
#define DEME_FAMILY_GHOST 2
#define DEME_FAMILY_PRESCRIBED 4 /* the family has a compiled motion prescription (deme_compile_prescriptions) */

/* status codes */
#define DEME_OK 0
#define DEME_ERR_INVALID 1
#define DEME_ERR_HIP 2
#define DEME_ERR_OVERFLOW 3 /* an arena (incidences / contacts) was too small even after growth */
#define DEME_ERR_BIN_TOO_FULL 4 /* reference: errOutBinSphNum abort, DEMContactKernels_SphereSphere.cu:121 */
#define DEME_ERR_VELOCITY 5     /* reference: errOutVel, kT.cpp:136-149 */
#define DEME_ERR_COMPILE 6
#define DEME_ERR_PEER 7 /* a collective call gave up because ANOTHER rank of the halo group reported an error (its own message says which) */

/* ---- parameter block: replaces deme::DEMSimParams (DEM/Defines.h:194-265) - */
typedef struct DemeParams {
    uint32_t nvXp2, nvYp2, nvZp2; /* voxel-count bit split, APIPrivate.cpp:373-487 */
    uint32_t nbX, nbY, nbZ;       /* bins per axis, HostSideHelpers.hpp:195-207 */
    double l;                     /* sub-voxel length unit */
    double voxelSize;             /* 2^16 * l */
    double binSize;
    float LBFX, LBFY, LBFZ; /* left-bottom-front corner of the world */
    float Gx, Gy, Gz;
    float h;              /* time step */
    float beta;           /* constant radius inflation (SetExpandFactor) */
    float approxMaxVel;   /* cap used by the margin formula, DEMMiscKernels.cu:50 */
    float expSafetyMulti; /* margin = (min(v,cap)*multi+adder)*h*drift + familyExtra */
    float expSafetyAdder;
    uint32_t integrator;        /* DEME_INTEGRATOR_* */
    uint32_t forceModel;        /* DEME_FORCE_* */
    uint32_t nContactWildcards; /* 4 for Hertzian, 0 frictionless, user value for custom */
    uint32_t cdUpdateFreq;      /* contact detection every K steps; 0 = every step with zero margin
                                   (the reference's SetCDUpdateFreq(0) parity mode) */
    uint32_t errOutBinSphNum;   /* default 32768 */
    float errOutVel;            /* default 1e15 (DEME_HUGE_FLOAT) */
    double timeElapsed;
} DemeParams;

/* ---- model description: replaces the DEMDataKT/DEMDataDT struct-of-pointers
 * (DEM/Defines.h:269-428) plus the JIT constant tables (_clumpTemplateDefs_,
 * _massDefs_, _moiDefs_, _analyticalEntityDefs_, _materialDefs_).
 * All pointers are HOST pointers, copied at upload time.  Owner order: clumps,
 * then analytical objects, then meshes (kT.cpp:804-829).  Sphere order is
 * clump-major. */
typedef struct DemeScene {
    uint32_t nOwners, nOwnerClumps, nSpheres, nAnal, nTri, nMat, nComp, nMassProps;
    /* per owner */
    const uint64_t* voxelID;
    const uint16_t *locX, *locY, *locZ;
    const float *oriQw, *oriQx, *oriQy, *oriQz;
    const float *vX, *vY, *vZ;
    const float *omgBarX, *omgBarY, *omgBarZ;
    const uint8_t* familyID;
    const uint16_t* inertiaPropOffsets;
    /* per sphere */
    const uint32_t* ownerClumpBody;
    const uint16_t* clumpComponentOffset;
    const uint16_t* sphereMaterialOffset;
    /* clump component tables (nComp) and mass-property tables (nMassProps) */
    const float *Radii, *CDRelPosX, *CDRelPosY, *CDRelPosZ;
    const float *MassProperties, *moiX, *moiY, *moiZ;
    /* analytical components (nAnal), AnalyticalCompDefJitify.cu:2-15 */
    const uint8_t* objType;
    const uint32_t* objOwner;
    const float* objNormal; /* +1 inward, -1 outward */
    const uint16_t* objMaterial;
    const float *objRelPosX, *objRelPosY, *objRelPosZ;
    const float *objRotX, *objRotY, *objRotZ;
    const float *objSize1, *objSize2, *objSize3;
    const float* objMass;
    /* materials: E[nMat], nu[nMat]; CoR/mu/Crr are nMat*nMat row-major (APIPrivate.cpp:1877-2026) */
    const float *E, *nu, *CoR, *mu, *Crr;
    /* families */
    const uint8_t* familyMasks;         /* DEME_FAMILY_MASK_ENTRIES, DEMHelperKernels.cuh:58-62 */
    const float* familyExtraMarginSize; /* 256 */
    const uint8_t* familyFlags;         /* 256, DEME_FAMILY_* bits */
    /* triangles (nTri): mesh-major, owner-local node coordinates, APIPrivate.cpp:756-810 */
    const uint32_t* ownerMesh;
    const float *triNode1, *triNode2, *triNode3; /* nTri*3 floats each, xyz interleaved */
    const uint16_t* triMaterialOffset;
    /* per owner, may be NULL: 1 = a ghost copy of a clump that another rank owns and integrates (slab decomposition, SURVEY
     * 8e).  A ghost keeps its TRUE family -- contact masks and family margins apply across a cut as inside a slab -- and is
     * refreshed from its owner rank every step (family included); ghost-ghost pairs are left to the ranks that own them.
     * 2 = a replicated owner that moves under contact forces (a free mesh or analytical body kept on every slab): each slab
     * sums the contributions of the spheres it owns (a ghost sphere's contact with it is left to the sphere's own rank), and
     * deme_halo_group_step adds the per-slab sums up (one ncclAllReduce of 8 floats per such owner) before every replica is
     * integrated with the total.  A context holding such owners can only be stepped through a halo group. */
    const uint8_t* ownerGhost;
} DemeScene;

/* mutable owner state for download/upload round trips */
typedef struct DemeOwnerState {
    uint64_t* voxelID;
    uint16_t *locX, *locY, *locZ;
    float *oriQw, *oriQx, *oriQy, *oriQz;
    float *vX, *vY, *vZ;
    float *omgBarX, *omgBarY, *omgBarZ;
    float *aX, *aY, *aZ;
    float *alphaX, *alphaY, *alphaZ;
    uint8_t* familyID;
} DemeOwnerState;

typedef struct DemeCounts {
    uint64_t nContacts;        /* current contact list length */
    uint64_t nPrevContacts;    /* previous list length (history source) */
    uint64_t nBinSphereTouches;/* (bin, sphere) incidences in the last detection */
    uint64_t nActiveBins;
    uint64_t nSteps;           /* steps taken since creation */
    uint64_t nDetections;
    uint32_t maxSpheresInBin;
    uint32_t lastStatus;
} DemeCounts;

typedef struct deme_ctx deme_ctx;

/* lifecycle.  Replaces DEMKinematicThread/DEMDynamicThread construction
 * (DEM/APIPublic.cpp:22-72) -- one object, one stream, no worker threads. */
int deme_ctx_create(int device, deme_ctx** out);
void deme_ctx_destroy(deme_ctx* ctx);
const char* deme_last_error(const deme_ctx* ctx);
const char* deme_version(void);
int deme_ctx_set_stream(deme_ctx* ctx, void* hip_stream); /* NULL = context-owned stream */
int deme_sync(deme_ctx* ctx);

/* Arithmetic mode of the per-step kernels.  Contact DETECTION (bin assignments, contact lists, history map) is the same
 * decision code in both modes -- bit-exact against the reference's fp64 predicates.
 *  DEME_ARITH_FAST  (default)  the contact-force kernel works from a per-owner derived view (world position in fp64, world
 *      angular velocity, mass: rewritten by the integrator every step), evaluates contacts in the world frame with 1-ulp
 *      hardware reciprocals / square roots and FMA contraction, and accumulates world-frame forces and torques that the
 *      integrator converts once per owner.  Same formulas as kernel/DEMCalcForceKernels.cu + FullHertzianForceModel.cu,
 *      re-associated: results agree with DEME_ARITH_EXACT to fp32 rounding (tolerance stated in tests/test_fast_mode.py).
 *  DEME_ARITH_EXACT  the reference's operation order and correctly rounded division / square root throughout: owner states
 *      bit-identical to the CPU oracle (the parity suite runs in this mode).
 * The process-wide default can be set with the environment variable DEME_ARITH=exact|fast. */
#define DEME_ARITH_FAST 1
#define DEME_ARITH_EXACT 0
int deme_set_arith_mode(deme_ctx* ctx, int mode);
int deme_get_arith_mode(const deme_ctx* ctx);
/* Which kernel evaluates the contact forces of the current list (what a profile of the stepping loop shows as its dominant
 * kernel): "k_tile_forces<M, MESH>" -- the owner-tile pass of the fast mode, csrc/deme_tile.h; MESH = true when the list holds
 * sphere-triangle contacts, which the mesh variant of the general kernel evaluates first -- , "k_forces_fast<M>" (fast mode, a
 * scene the tile pass does not take: replicated free owners, more than 16 materials, tables beyond 4 KB), "k_calc_forces<M, 0>"
 * (exact mode, contact recording) or "deme_custom_forces_ss" (a run-time compiled model); M = 0 Hertzian with history, 1
 * frictionless.  Also the largest tile's foreign-owner count and local list length of the current list.
 * deme_tile_stats: {tiles, tiles that do not fit the LDS area and are evaluated by k_tile_forces_big -- one workgroup each, the
 * rest of the list stays tiled --, largest halo, largest local list} of the current list (zeros when it is not tiled). */
int deme_tile_stats(const deme_ctx* ctx, uint32_t out[4]);
/* A run-time compiled force model is compiled into the tile pass as well ("deme_custom_tile<MESH>": csrc/deme_jit.h splices the
 * user's statements where the Hertzian block of k_tile_forces is, the reference's vocabulary filled from the staged records,
 * DEMCalcForceKernels.cu:233-251).  A list takes it when its tiles hold at least minContactsPerTile contacts on average (default
 * 320: below that -- single spheres with one or two contacts each -- the general kernel is faster, measured on configs[4]);
 * 0 = always. */
int deme_set_tile_policy(deme_ctx* ctx, uint32_t minContactsPerTileCustom);
int deme_force_kernel_name(const deme_ctx* ctx, char* name, size_t cap, uint32_t* tileMaxHalo, uint32_t* tileMaxList);
/* The one-kernel time step (csrc/deme_tile_step.h, "k_tile_step<M>"): every tile is CLOSED -- it also evaluates the contacts that
 * hold its owners as B from other tiles (a contact that straddles two tiles is evaluated by both, bit-identically) -- so that the
 * tile owns the complete sums of its owners and integrates them in the same launch (calculateForces + integrateOwners of
 * dT.cpp:2424-2447 in one kernel; owners and contact history are double-buffered).  OFF by default: on the packed bed of BASELINE
 * configs[1] the closed tiles stage ~220 foreign owners instead of ~95 and evaluate 1.2x the contacts, and the one launch
 * (143 us) is slower than force pass + integrator (92 + 43 us) -- DESIGN.md 3.7.  Built-in models, FAST arithmetic, no mesh,
 * no ghosts, no prescription / family-rule kernels, no recording; anything else keeps the two-kernel form silently.
 * Where it pays (round 5, profiles/r05/fused_small_beds.txt): beds whose step is two short launches -- 10^4 clumps 0.0258 -> 0.0192
 * ms/step (-26 %), 5x10^4 0.0373 -> 0.0306 (-18 %); at 2x10^5 the closed tiles no longer fit and the two-kernel form runs anyway.
 * on = 2: by size -- the one-kernel step for scenes of at most 10^5 owners (what the C++ shell asks for), the two kernels beyond.
 * Environment DEME_FUSED=2 / 1 / 0 overrides the switch for every context of the process. */
int deme_set_fused_step(deme_ctx* ctx, int on);

/* The engine's own numbering (csrc/deme_order.inc).  Owner and sphere ids at this boundary are ALWAYS the caller's -- load order,
 * as in the reference (DEM/dT.cpp:700-800).  Inside, a scene uploaded in the fast arithmetic mode (no ghosts, built-in force
 * model) whose clumps are not already numbered compactly is kept in box-shaped tiles of 128 clumps (recursive bisection), so that
 * the owner tiles of the force pass are clusters of the bed; every entry point translates.  *reordered = 1 if the current scene is
 * kept that way; spread[0] / [1]: mean surface of the bounding box of 128 consecutive clumps (in squares of one bin edge), in the
 * caller's order / in the engine's (0 when the question was not asked).  deme_set_reorder(ctx, 0) before deme_upload_scene keeps the caller's order (env DEME_REORDER=0
 * does it for the process). */
int deme_get_order(const deme_ctx* ctx, int* reordered, double spread[2]);
int deme_set_reorder(deme_ctx* ctx, int enable);
/* A running simulation drifts away from the order it was given (a mixer, a flow).  The engine watches the tiles of every detection
 * (mean foreign owners per tile, tiles that no longer fit) and renews its order at the start of a detection when they have
 * degraded by half since the last ordering, at most every 20 detections; deme_renew_order does it now.  The caller's ids never
 * change.  Host-driven (positions to the host, the order as at upload, records, the current contact list and its history
 * re-keyed): about a second at 1e6 clumps. */
int deme_renew_order(deme_ctx* ctx);
int deme_order_renewals(const deme_ctx* ctx, uint64_t* n); /* how many times the order was renewed (by the engine or by the call above) */
/* The order itself, without a context or a GPU (host code; tests): order[k] = the caller's clump kept in slot k. */
int deme_order_probe(const DemeParams* p, size_t nClumps, const uint64_t* voxelID, const uint16_t* locX, const uint16_t* locY,
                     const uint16_t* locZ, uint32_t* order, double spread[2]);

/* setSimParams / UpdateSimParams (APIPrivate.cpp:1121, dT.cpp:2463-2466) */
int deme_set_params(deme_ctx* ctx, const DemeParams* p);
/* allocateGPUArrays + initGPUArrays + packDataPointers (APIPrivate.cpp:1169-1290) */
int deme_upload_scene(deme_ctx* ctx, const DemeScene* s);
/* DEMTracker setters / SetTriNodeRelPos analogue: overwrite owner state (n = nOwners) */
int deme_upload_owner_state(deme_ctx* ctx, const DemeOwnerState* st);
int deme_download_owner_state(deme_ctx* ctx, DemeOwnerState* st);
/* SetTriNodeRelPos / DEMTracker::UpdateMesh (APIPublic.cpp:709-730): rewrite nTri*3 floats per node array */
int deme_update_tri_nodes(deme_ctx* ctx, const float* n1, const float* n2, const float* n3);

/* kT unpackMyBuffer margin step (kT.cpp:100-191, DEMMiscKernels.cu:37-61):
 * per-owner margin from |v|; drift = number of steps the list must stay valid.
 * drift 0 => margin = familyExtraMargin only. */
int deme_compute_margins(deme_ctx* ctx, uint32_t drift);
/* direct margin override, n = nOwners (tests) */
int deme_set_margins(deme_ctx* ctx, const float* marginSize);

/* contactDetection() (algorithms/DEMCubContactDetection.cu:38-1123): binning,
 * bin-sorted sweep, sphere-analytical and sphere-triangle detection, history map. */
int deme_detect_contacts(deme_ctx* ctx);
/* dT unpack + migrateEnduringContacts (dT.cpp:1955-1987, 2040-2144): permute
 * contact wildcards through the history map of the last detection. */
int deme_migrate_history(deme_ctx* ctx);
/* calculateForces() (dT.cpp:2146-2214): clear a/alpha, per-contact force, accumulate. */
int deme_calc_forces(deme_ctx* ctx);
/* integrateOwnerMotions() (dT.cpp:2216-2224) */
int deme_integrate(deme_ctx* ctx);
/* DoDynamics inner loop (dT.cpp:2401-2467): nsteps of {detect every K, forces, integrate}. */
int deme_step(deme_ctx* ctx, uint32_t nsteps);
/* The reference runs its contact detection (kT) beside the dynamics (dT) with bounded staleness (kT.cpp:100-216, dT.cpp:1955-2038,
 * 2276-2299).  Here, opt-in: leadSteps = D > 0 makes deme_step start a detection D steps before the list is due -- from a copy of
 * the owner records, on a stream of its own, with margins for the K + D steps between the copy and the end of the new list's
 * service -- while the main stream works through those D steps with the current list; the new list is swapped in when they are
 * enqueued.  No contact is missed; the list holds a few more near-pairs (zero contributions) than a lock-step detection's, so an
 * exact-mode trajectory is the lock-step one.  Needs K > D and a deme_step call of at least D steps at the right moment; scenes
 * with a mesh, ghosts, persistent contacts or the adaptive controllers keep the lock-step detection.  0 switches it off. */
int deme_set_async_detection(deme_ctx* ctx, uint32_t leadSteps);

int deme_get_counts(deme_ctx* ctx, DemeCounts* out);

/* outputs for parity checks (capacity in elements; returns DEME_ERR_INVALID if too small) */
/* (bin, sphere) incidence list in bin-sorted order: binIDsEachSphereTouches_sorted /
 * sphereIDsEachBinTouches_sorted (DEMCubContactDetection.cu:169-186) */
int deme_download_bin_incidence(deme_ctx* ctx, uint32_t* binIDs, uint32_t* sphereIDs, size_t cap);
/* contact list: idGeometryA/B, contactType, contactMapping (Defines.h:306-322) */
int deme_download_contacts(deme_ctx* ctx, uint32_t* idA, uint32_t* idB, uint8_t* type, uint32_t* mapping, size_t cap);
/* contact wildcards, array w (0..nContactWildcards-1), length nContacts */
int deme_download_contact_wildcard(deme_ctx* ctx, uint32_t w, float* out, size_t cap);
int deme_upload_contact_wildcard(deme_ctx* ctx, uint32_t w, const float* in, size_t n);
/* Restart: load a saved contact list with its wildcards (DEMClumpBatch::SetExistingContacts /
 * SetExistingContactWildcards, Structs.h:857-880; loaded by dT at initialisation).  The
 * pairs are geometry ids (sphere A; sphere / triangle / analytical component B), `type` the reference's
 * contact_t (1 SS, 2 SM, other: analytical); `wildcards` is float[n][nContactWildcards].  The list only feeds
 * the history map of the next detection, which therefore always runs before the next force evaluation. */
int deme_seed_contacts(deme_ctx* ctx, const uint32_t* idA, const uint32_t* idB, const uint8_t* type, const float* wildcards,
                       size_t n);
/* per-contact records (ContactInfoWriteBack.cu): force, torque-only force, local contact
 * points; each nContacts*3 floats; any pointer may be NULL. Recording must have been enabled. */
int deme_set_record_contacts(deme_ctx* ctx, int enable);
int deme_download_contact_records(deme_ctx* ctx, float* force, float* torqueOnly, float* cpA, float* cpB, size_t cap);
/* per-sphere world position (LBF-shifted frame, as kT sees it) and inflated radius */
int deme_download_sphere_geometry(deme_ctx* ctx, double* X, double* Y, double* Z, float* R, size_t cap);

/* Force-model hook (DEMForceModel::DefineCustomModel, AuxClasses.h:422-485;
 * equipForceModel APIPrivate.cpp:1381-1574): splice a user statement block into
 * the contact-force kernel and compile it for gfx950 at run time (hipRTC).
 * wildcardNames: nWildcards names of per-contact float history variables. */
int deme_compile_force_model(deme_ctx* ctx, const char* src, size_t len, const char* const* wildcardNames,
                             uint32_t nWildcards, const char* prerequisites);

/* Family motion prescriptions (SetFamilyPrescribedLinVel / AngVel / Position / Quaternion, AddFamilyPrescribedAcc /
 * AngAcc; API.h:720-838).  The three strings are the bodies of the `switch (family)` statements the reference generates in
 * equipFamilyPrescribedMotions (APIPrivate.cpp:1600-1708) for applyPrescribedVel / applyPrescribedPos /
 * applyAddedAcceleration (DEMIntegrationKernels.cu:8-98): "case <family>: { ...; break; }" sequences over the names
 * vX..omgBarZ, X, Y, Z, oriQw..oriQz, accX..angAccZ, t, ownerID and the ...Prescribed flags.  They are compiled for gfx950
 * with hipRTC; the families concerned must carry DEME_FAMILY_PRESCRIBED in DemeScene.familyFlags.  NULL / "" clears. */
int deme_compile_prescriptions(deme_ctx* ctx, const char* velCases, const char* posCases, const char* accCases);

/* On-the-fly family changes (DEMSolver::ChangeFamilyWhen, API.h:1024; kernel applyFamilyChanges,
 * DEMModeratorKernels.cu:10-60; run between the force evaluation and the integration of every step, dT.cpp:2437-2443).
 * `rules` is the _familyChangeRules_ text equipFamilyOnFlyChanges generates (APIPrivate.cpp:1576-1598).  If it mentions
 * acc / accX / accY / accZ the owners' contact accelerations are reduced before the rules run.  NULL / "" clears. */
int deme_compile_family_rules(deme_ctx* ctx, const char* rules);
/* DEMSolver::ChangeFamily(ID_from, ID_to) (API.h:1028): immediate, all owners of a family */
int deme_change_family(deme_ctx* ctx, uint32_t from, uint32_t to);

/* Adaptive controllers (reference: DEMKinematicThread::calibrateParams DEM/kT.cpp:43-98 -- bin-size
 * hill climb on kT's time per detection; DEMDynamicThread::calibrateParams DEM/dT.cpp:2276-2299 -- drift tuner;
 * API: UseAdaptiveBinSize, SetAdaptiveBinSizeDelaySteps / MaxRate / Acc / UpperProactivity / LowerProactivity,
 * UseAdaptiveUpdateFreq, SetCDMaxUpdateFreq, DEM/API.h:253-309).  Here both run on device timers (HIP events around the
 * detection / around a window of steps), inside deme_step:
 *  - bin size: every binObserveSteps detections the average device time per detection is compared with the previous
 *    window's; the change rate accelerates in the current direction on an improvement and turns round otherwise, with the
 *    reference's rule, clamps and safety overrides (too many spheres in a bin => shrink; too many bins => grow);
 *  - update frequency: this build has no second thread to drift ahead of, so the quantity tuned is K = cdUpdateFreq itself,
 *    hill-climbing the measured time per step (detection amortised over K + force pass over the K-inflated list) every
 *    freqObserveDetections detections, within [1, maxUpdateFreq].
 * Neither changes the physics: the contact set does not depend on the bin size, and a larger K only adds non-touching
 * pairs to the list (tests/test_adaptive.py: bit-identical state with the controllers on).  Off by default. */
typedef struct DemeAdaptive {
    uint32_t autoBinSize;           /* UseAdaptiveBinSize */
    uint32_t binObserveSteps;       /* SetAdaptiveBinSizeDelaySteps (reference default 25) */
    float binMaxRate;               /* SetAdaptiveBinSizeMaxRate (0.05) */
    float binAcc;                   /* SetAdaptiveBinSizeAcc (0.1) */
    float binUpperSafety;           /* 0.25: shrink when maxSpheresInBin > this * errOutBinSphNum */
    float binLowerSafety;           /* 0.3: grow when the bin count > this * 2^32 */
    uint32_t autoUpdateFreq;        /* UseAdaptiveUpdateFreq */
    uint32_t maxUpdateFreq;         /* SetCDMaxUpdateFreq */
    uint32_t freqObserveDetections; /* detections per adjustment of K */
} DemeAdaptive;
int deme_set_adaptive(deme_ctx* ctx, const DemeAdaptive* a);
/* current bin size / cdUpdateFreq and how many adjustments each controller has made */
int deme_get_adaptive_state(deme_ctx* ctx, double* binSize, uint32_t* cdUpdateFreq, uint32_t* nBinChanges, uint32_t* nFreqChanges);

/* Acceleration added by the script for the NEXT step only (reference: DEMTracker::AddAcc / AddAngAcc, AuxClasses.h:264-274 ->
 * DEMDynamicThread::addOwnerNextStepAcc / addOwnerNextStepAngAcc, dT.cpp:3160-3174: the values are written into a / alpha and a
 * flag keeps prepareAccArrays from clearing them once, DEMPrepForceKernels.cu:14-31 -- co-simulation hands forces over this way).
 * acc / angAcc: n x 3 floats for owners [owner, owner + n), either may be null; like the reference's setVal a second call
 * for the same owner replaces the first.  The integrator adds them to the contact sums of the coming step (after the sums,
 * whose order stays fixed) and the records clear themselves.  angAcc is in the owner's local frame like alpha. */
int deme_add_owner_acc(deme_ctx* ctx, uint32_t owner, uint32_t n, const float* acc, const float* angAcc);

/* Persistent contacts (reference: DEM/API.h:874-905 MarkFamilyPersistentContactEither/Both, MarkFamilyPersistentContact,
 * MarkPersistentContact and their Remove* inverses; DEM/APIPrivate.cpp:33-117; algorithms/DEMCubContactDetection.cu:605-802).
 * Qualifies contacts of the CURRENT list: mode 0 every contact, 1 either owner's family == N1, 2 both == N1, 3 the family
 * pair (N1, N2) in either order.  mark != 0: they stay in the contact list at every later detection whether or not the
 * sweep finds them (the force kernel still treats a separated pair as NOT_A_CONTACT); mark == 0: the qualification is
 * removed.  Refused for a history-less force model, like the reference.  Host-synchronous. */
int deme_mark_persistent_contacts(deme_ctx* ctx, int mode, uint32_t N1, uint32_t N2, int mark);
int deme_num_persistent_contacts(deme_ctx* ctx, size_t* n);
/* The marked set itself, as (sphere A, geometry B, contact type) triples in list order -- for a restart or a re-decomposition
 * (dem-engine_amd/decomp.py carries the marks to the ranks that own the pairs afterwards).  Upload replaces the set. */
int deme_download_persistent_contacts(deme_ctx* ctx, uint32_t* idA, uint32_t* idB, uint8_t* type, size_t cap);
int deme_upload_persistent_contacts(deme_ctx* ctx, const uint32_t* idA, const uint32_t* idB, const uint8_t* type, size_t n);

/* Owner and geometry wildcards of user force models (DEMForceModel::SetPerOwnerWildcards / SetPerGeometryWildcards,
 * AuxClasses.h:422-485; Models.h:319-360).  Owner wildcards are per-owner float arrays the fragment sees as `name`,
 * `name_A`, `name_B` (aliases, indexed by AOwner / BOwner); geometry wildcards are per-sphere / per-triangle /
 * per-analytical-component arrays seen as `name_A[AGeo]`, `name_B[BGeo]`.  Arrays start at zero; at most 8 of each.
 * deme_compile_force_model_ex declares them (after deme_upload_scene), the upload / download calls move one array:
 * kind 0 owners, 1 spheres, 2 triangles, 3 analytical components. */
int deme_compile_force_model_ex(deme_ctx* ctx, const char* src, size_t len, const char* const* contactWildcards, uint32_t nContactWc,
                                const char* const* ownerWildcards, uint32_t nOwnerWc, const char* const* geoWildcards,
                                uint32_t nGeoWc, const char* prerequisites);
int deme_upload_wildcard_array(deme_ctx* ctx, uint32_t kind, uint32_t index, const float* in, size_t n);
int deme_download_wildcard_array(deme_ctx* ctx, uint32_t kind, uint32_t index, float* out, size_t cap);

/* SetFamilyClumpMaterial / SetFamilyMeshMaterial (DEM/API.h:970-974; dT::setFamilyClumpMaterial, dT.cpp): every sphere
 * (kind 0) or triangle (kind 1) whose owner currently belongs to `family` takes material index `material` (load order, < nMat).
 * The per-contact records carry materials, so the next step starts with a contact detection. */
int deme_set_family_material(deme_ctx* ctx, uint32_t family, uint32_t material, int kind);
/* device memory in use / in total, as hipMemGetInfo reports it for the context's device (DEMSolver::ShowMemStats, API.h:584) */
int deme_device_memory(deme_ctx* ctx, size_t* usedBytes, size_t* totalBytes);

/* compile-only check of a fragment (no context, no GPU needed): same generator and hipRTC options,
 * 2 dummy materials; the compiler log is copied into `log`. */
int deme_jit_probe(const char* src, const char* const* wildcardNames, uint32_t nWildcards, const char* prerequisites,
                   char* log, size_t logCap);
/* the same with owner and geometry wildcards declared (DEMForceModel::SetPerOwnerWildcards / SetPerGeometryWildcards) */
int deme_jit_probe_ex(const char* src, const char* const* wildcardNames, uint32_t nWildcards, const char* const* ownerNames,
                      uint32_t nOwnerWc, const char* const* geoNames, uint32_t nGeoWc, const char* prerequisites, char* log,
                      size_t logCap);

/* Inspectors (DEMInspector, AuxClasses.cpp:19-170; kernels DEMSphereQueryKernels.cu:13-54,
 * DEMOwnerQueryKernels.cu:11-63; reduction dT.cpp:2556-2640): a per-sphere or per-owner quantity evaluated on
 * the device and reduced there.  Ghost owners of a slab decomposition are left out.  No region filter. */
#define DEME_INSPECT_CLUMP_MAX_Z 0          /* "clump_max_z": max over spheres of Z + radius */
#define DEME_INSPECT_CLUMP_MIN_Z 1          /* "clump_min_z" */
#define DEME_INSPECT_CLUMP_MAX_ABSV 2       /* "clump_max_absv": sphere-centre speed incl. rotation */
#define DEME_INSPECT_CLUMP_MASS 3           /* "clump_mass": sum over clump owners */
#define DEME_INSPECT_MAX_ABSV 4             /* "max_absv": max owner speed, all owner kinds */
#define DEME_INSPECT_CLUMP_KINETIC_ENERGY 5 /* "clump_kinetic_energy" */
#define DEME_INSPECT_ABSV 6                 /* "absv": per-owner values only (deme_inspect_values) */
#define DEME_INSPECT_CLUMP_VOLUME 7         /* "clump_volume": sum of the declared template volumes (deme_upload_volumes) */
int deme_inspect(deme_ctx* ctx, uint32_t quantity, float* out);
/* unreduced values (DEMInspector::GetValues): one per sphere (quantities 0-2) or per owner (3-6) */
int deme_inspect_values(deme_ctx* ctx, uint32_t quantity, float* out, size_t cap);
/* Region-limited inspection (reference: DEMSolver::CreateInspector(quantity, region) DEM/API.h:675, AuxClasses.cpp:205-223).
 * `code` is the reference's region string: a C++ statement block that returns a bool from float X, Y, Z (sphere centre for
 * the per-sphere quantities, owner CoM otherwise), e.g. "return (X * X + Y * Y <= 0.25 * 0.25) && (Z <= -0.3);".  It is
 * compiled once at run time; elements outside the region do not take part in the reduction (an empty region yields the
 * reduction's identity: -FLT_MAX / FLT_MAX / 0).  deme_inspect(q) == deme_inspect_region(q, -1). */
int deme_compile_region(deme_ctx* ctx, const char* code, int* regionId);
int deme_inspect_region(deme_ctx* ctx, uint32_t quantity, int regionId, float* out);
/* Declared clump volumes, one per mass-property entry (reference: DEMClumpTemplate::volume, APIPrivate.cpp:730, 1849-1858). */
int deme_upload_volumes(deme_ctx* ctx, const float* volumes, size_t n);

/* timing of the kernels this library launched (HIP events on the context stream);
 * names: "calc_forces", "integrate", "detect"; returns avg ms per launch since last reset */
int deme_kernel_time_ms(deme_ctx* ctx, const char* name, double* avg_ms, uint64_t* launches);
int deme_kernel_time_reset(deme_ctx* ctx);
/* enable = 0: off; n > 0: every n-th launch of each timed kernel is bracketed with events (an event pair costs ~3 us of
 * dispatch gap, 4.7 % of a 0.29 ms step when every launch is timed) */
int deme_set_timing(deme_ctx* ctx, int enable);

/* multi-GPU slab decomposition helpers (no reference equivalent).
 * pack/unpack ghost-owner state into a caller-provided DEVICE buffer that the
 * host exchanges with RCCL send/recv. 56 bytes per ghost. */
#define DEME_GHOST_BYTES 56
int deme_halo_pack(deme_ctx* ctx, const uint32_t* d_ownerIDs, uint32_t n, void* d_buf);
int deme_halo_unpack(deme_ctx* ctx, const uint32_t* d_ownerIDs, uint32_t n, const void* d_buf);

/* Halo exchange overlapped with the interior force evaluation (north_star; no reference equivalent).  The context owns a
 * second stream for the ghost traffic.  Per step:
 *   deme_step_overlap_begin   forces of the owner runs that read no ghost owner (compute stream); *detectionDue = 1 when
 *                             this step starts with a contact detection, which needs the ghosts first: nothing is launched
 *   deme_halo_pack_async      pack on the halo stream, ordered after the previous step's integration
 *   (the caller's send / recv of the packed records, enqueued on deme_halo_stream)
 *   deme_halo_unpack_async    unpack on the halo stream
 *   deme_step_overlap_end     compute stream waits for the unpack; remaining forces, integration
 *                             (or, when a detection was due, the whole un-split step)
 * The two force passes together evaluate every contact exactly once, in the same arithmetic as deme_step. */
int deme_halo_stream(deme_ctx* ctx, void** hipStream);
int deme_halo_pack_async(deme_ctx* ctx, const uint32_t* d_ownerIDs, uint32_t n, void* d_buf);
int deme_halo_unpack_async(deme_ctx* ctx, const uint32_t* d_ownerIDs, uint32_t n, const void* d_buf);
int deme_halo_sync(deme_ctx* ctx);
int deme_step_overlap_begin(deme_ctx* ctx, int* detectionDue);
int deme_step_overlap_end(deme_ctx* ctx);


/* ---- the exchange driven from the library (north_star: "RCCL halo exchange of ghost clumps over xGMI ... overlapped with
 * interior force evaluation on a second HIP stream", host orchestration in C++).  A halo group = one RCCL communicator + the
 * slabs this process holds (normally one: one process per GPU; several slabs in one process exchange by sends to self, which is
 * how the path is tested on a single GPU).  RCCL is bound at run time (a process with PyTorch loaded shares PyTorch's copy).
 *   deme_halo_unique_id       ncclGetUniqueId: rank 0 calls it and hands the 128 bytes to the other ranks (any side channel)
 *   deme_halo_group_create    ncclCommInitRank(world, id, rank) on `device`; id == NULL with world == 1: a one-rank communicator
 *   deme_halo_group_attach    a slab: its context, and per side the neighbour's rank (-1: none; the own rank together with the
 *                             neighbour's context when this process holds it), the LOCAL owner ids whose records are sent there
 *                             and the local owner ids (ghost copies) that take the records arriving from there; lists in the
 *                             same clump order on both sides (dem-engine_amd/decomp.py builds them)
 *   deme_halo_group_step      nsteps x { interior force pass | pack -> ncclGroupStart, ncclSend / ncclRecv per face,
 *                             ncclGroupEnd -> unpack | ghost-dependent force pass, [ncclAllReduce of the replicated free
 *                             owners' a / alpha,] integration } for every attached slab; asynchronous like deme_step
 *                             (deme_halo_group_sync waits)
 *   deme_halo_group_exchange  the exchange alone (ghost copies refreshed after an upload) */
typedef struct deme_halo_group deme_halo_group;
int deme_halo_unique_id(unsigned char* id128);
int deme_halo_group_create(const unsigned char* id128, int rank, int world, int device, deme_halo_group** out);
void deme_halo_group_destroy(deme_halo_group* g);
const char* deme_halo_group_last_error(const deme_halo_group* g);
int deme_halo_group_attach(deme_halo_group* g, deme_ctx* ctx, int leftRank, deme_ctx* leftLocal, const uint32_t* sendLeft,
                           uint32_t nSendLeft, const uint32_t* recvLeft, uint32_t nRecvLeft, int rightRank, deme_ctx* rightLocal,
                           const uint32_t* sendRight, uint32_t nSendRight, const uint32_t* recvRight, uint32_t nRecvRight);
int deme_halo_group_step(deme_halo_group* g, uint32_t nsteps);
int deme_halo_group_exchange(deme_halo_group* g);
int deme_halo_group_sync(deme_halo_group* g);
int deme_halo_group_stats(const deme_halo_group* g, uint64_t* exchanges, uint64_t* bytesSentPerStep);
/* host microseconds spent enqueuing since creation / the last reset: [0] interior force passes, [1] packs, [2] the RCCL group,
 * [3] unpacks + boundary passes + integration */
int deme_halo_group_host_time(deme_halo_group* g, double us[4], int reset);
/* Migration inside the library (SURVEY 8e "migration at re-bin time"; the algorithm of decomp.migrate_neighbours, on the device).
 *   deme_halo_group_set_slab     what the group has to know about a slab beyond its exchange lists: global clump / sphere ids of
 *                                its owners and spheres (scene order: own clumps, ghosts from the left, ghosts from the right,
 *                                replicated owners), the slab's x range [xLo, xHi) and the halo thickness; flipMask: the contact
 *                                wildcards that are B->A vectors (bit w; 7 = delta_tan_x/y/z of the Hertzian model)
 *   deme_halo_group_migrate      collective over the group: every own clump whose centre crossed a face moves to that face
 *                                neighbour with its state, template ids and the history rows of its contacts (global sphere
 *                                ids; ncclSend / ncclRecv between ranks, buffer hand-over between slabs of one process), ghost
 *                                sets are renewed from the neighbours' current clumps, every slab is re-assembled and re-seeded on
 *                                the device and its exchange lists rebuilt.  Only counts pass through the host.  The next step of
 *                                every slab starts with a contact detection.
 *   deme_halo_group_slab_counts  own / ghost-left / ghost-right clumps, owners, spheres, seeded contacts of a slab
 *   deme_halo_group_download_ids global ids (and sphere -> owner / component) of a slab's current numbering, for the caller's books */
int deme_halo_group_set_slab(deme_halo_group* g, deme_ctx* ctx, const uint32_t* ownerGlobal, const uint32_t* sphereGlobal, uint32_t nOwn,
                             uint32_t nGhostLeft, uint32_t nGhostRight, double xLo, double xHi, double halo, uint32_t flipMask);
int deme_halo_group_migrate(deme_halo_group* g, uint32_t* clumpsMoved);
int deme_halo_group_slab_counts(const deme_halo_group* g, const deme_ctx* ctx, uint32_t counts[6]);
int deme_halo_group_download_ids(deme_halo_group* g, deme_ctx* ctx, uint32_t* ownerGlobal, uint32_t* sphereGlobal, uint32_t* sphereOwner,
                                 uint16_t* sphereComp);
/* how many ranks the group's RCCL communicator spans, as RCCL itself reports it (ncclCommCount): a scaling run checks this
 * against the number of processes it believes it launched */
int deme_halo_group_comm_count(const deme_halo_group* g, int* ranks);
/* One evaluation -- and one contact history -- per contact that straddles a cut (the reference's rule that a pair
 * belongs to exactly one bin, DEMContactKernels_SphereSphere.cu:212, carried over to ranks).  0 (default): both ranks evaluate a
 * contact between an own clump and a ghost, each keeps its history copy, the force on the ghost is dropped.  1: the LEFT slab of a
 * cut evaluates it; the right slab leaves the pair (and its ghosts' wall / mesh contacts) off its list, and after the
 * ghost-dependent force pass of every step the left slab sends a / alpha of each right ghost's contact sum to the ghost's owner
 * (32 bytes per ghost, ncclSend / ncclRecv in one group, the forward exchange's lists read backwards), which adds them before
 * integrating.  Call after every slab is attached; the contact lists are rebuilt at the next step.  Not combined with replicated
 * free owners (DemeScene.ownerGhost = 2). */
int deme_halo_group_set_cross_contacts(deme_halo_group* g, int evaluateOnce);

/* ---- the decomposition itself, behind the C-ABI (round 5; north_star: "the reference's two-GPU kT/dT split is replaced by a per-GPU
 * spatial-domain decomposition ... host orchestration in C++").  The reference picks its devices in the constructor
 * (DEM/API.h:52-56, DEM/APIPublic.cpp:22-110: DEMSolver(nGPUs) / DEMSolver(device ids)); here that constructor opens a deme_multi.
 *
 * deme_decomp_*: the PLAN -- pure host code, no device needed.  A global scene (no ghosts) is cut into nSlabs slabs along `axis`
 * (0 / 1 / 2, or -1: the longest side of the clumps' bounding box) at equal-count boundaries snapped to bin faces (`edges`: nSlabs + 1
 * boundaries of the caller's own instead; the outer two are taken as -inf / +inf); a slab's scene is [own clumps | ghosts from the
 * lower neighbour | ghosts from the upper neighbour | replicated owners], a ghost being a neighbour's clump whose centre lies within
 * `halo` of the shared face (halo <= 0: four clump reaches + twice the widest family margin).  Every rule is the one
 * dem-engine_amd/decomp.py states in numpy (the CPU tests hold the two against each other). */
#define DEME_DECOMP_SHARED_FREE 1u /* allow replicated owners that move under contact forces (DemeScene.ownerGhost = 2) */
#define DEME_DECOMP_NO_SNAP 2u     /* keep the equal-count boundaries where the quantiles put them (not snapped to bin faces) */
#define DEME_DECOMP_SPATIAL_ORDER 4u /* number every slab's own clumps in the engine's own order (compact owner tiles, csrc/deme_order.inc;
                                      * ghosts follow their owner slab's order) instead of by ascending global id: a scene with ghosts keeps
                                      * the order it is uploaded in, so this is what gives a slab of a sampler-ordered bed its tiles.
                                      * deme_multi_build sets it unless DEME_DECOMP_CALLER_ORDER is given */
#define DEME_DECOMP_CALLER_ORDER 8u
typedef struct deme_decomp deme_decomp;
int deme_decomp_create(const DemeParams* p, const DemeScene* scene, uint32_t nSlabs, int axis, double halo, const double* edges, uint32_t flags,
                       deme_decomp** out, char* err, size_t errCap);
void deme_decomp_destroy(deme_decomp* d);
int deme_decomp_info(const deme_decomp* d, uint32_t* nSlabs, int* axis, double* halo, double* edges /* nSlabs + 1 */, uint32_t* nFreeReplicated);
/* one slab of the plan: its scene (pointers into the plan's storage: valid until deme_decomp_destroy), {own, ghosts from below, ghosts
 * from above} clump counts, the global owner id of every slab owner and the global sphere id of every slab sphere, the LOCAL owner
 * ids sent to the lower / upper neighbour (what arrives lands on the ghost ranges, in order), the slab's range along the axis */
int deme_decomp_slab(const deme_decomp* d, uint32_t slab, DemeScene* scene, uint32_t counts[3], const uint32_t** ownerGlobal,
                     const uint32_t** sphereGlobal, const uint32_t** sendLower, uint32_t* nSendLower, const uint32_t** sendUpper,
                     uint32_t* nSendUpper, double range[2]);

/* The slabs of a plan that THIS rank holds (slab s belongs to rank s * world / nSlabs; nSlabs must be a multiple of the group's world)
 * become contexts on the group's device: deme_ctx_create, deme_set_arith_mode(arith: DEME_ARITH_*, or -1 for the process default),
 * deme_set_params, deme_upload_scene, deme_halo_group_attach, deme_halo_group_set_slab (flipMask as there), then one exchange.
 * The contexts belong to the group (deme_halo_group_destroy destroys them); deme_halo_group_slab_ctx hands them out in slab order
 * for what a script sets per context (force models, prescriptions, recording ...).  deme_halo_group_set_axis: the axis the slabs of
 * an attach-by-hand group were cut along (default 0; _build sets it from the plan). */
int deme_halo_group_build(deme_halo_group* g, const DemeParams* p, const deme_decomp* plan, int arith, uint32_t flipMask);
int deme_halo_group_set_axis(deme_halo_group* g, int axis);
int deme_halo_group_slab_ctx(deme_halo_group* g, uint32_t i, deme_ctx** ctx, uint32_t* globalSlab);
int deme_halo_group_num_slabs(const deme_halo_group* g, uint32_t* n);
/* Owner state by GLOBAL owner id: every slab of this rank writes its OWN clumps' rows (replicated owners: the first slab's) into the
 * caller's arrays of the global scene's length / reads them from there (ghost copies are refreshed by one exchange).  With several
 * ranks each process fills / takes only its part. */
int deme_halo_group_download_state(deme_halo_group* g, DemeOwnerState* global, uint32_t nOwnersGlobal);
int deme_halo_group_upload_state(deme_halo_group* g, const DemeOwnerState* global, uint32_t nOwnersGlobal);

/* One process, several devices: the reference's DEMSolver(nGPUs) / DEMSolver(std::vector<int>) (DEM/API.h:52-56).  deme_multi_create
 * checks every id against the visible devices (an absent one: DEME_ERR_INVALID, the message names it -- GpuManager.cpp:64-68 throws
 * there), opens one halo group per device -- one RCCL rank each, the communicators initialised together in one grouped
 * ncclCommInitRank -- and deme_multi_build cuts the scene into nDevices * slabsPerDevice slabs (deme_decomp_create) and hands every
 * device its block (deme_halo_group_build).  deme_multi_step steps all groups (one host thread per device when there are several:
 * each group's loop is the per-process loop of deme_halo_group_step) and migrates clumps between slabs every `migrateEvery` steps
 * (deme_multi_set_migration; 0 = never).  With one device and one slab per device this is a plain context behind the same calls. */
typedef struct deme_multi deme_multi;
int deme_multi_create(const int* devices, int nDevices, deme_multi** out, char* err, size_t errCap);
void deme_multi_destroy(deme_multi* m);
const char* deme_multi_last_error(const deme_multi* m);
int deme_multi_build(deme_multi* m, const DemeParams* p, const DemeScene* scene, uint32_t slabsPerDevice, int axis, double halo, uint32_t flags,
                     int arith, uint32_t flipMask);
int deme_multi_num_slabs(const deme_multi* m, uint32_t* n);
int deme_multi_slab_ctx(deme_multi* m, uint32_t slab, deme_ctx** ctx);
/* the same handle for questions that change nothing (kernel names, bin size, timers): the merged contact list of the run stays valid */
int deme_multi_slab_ctx_peek(const deme_multi* m, uint32_t slab, const deme_ctx** ctx);
int deme_multi_set_migration(deme_multi* m, uint32_t migrateEvery);
int deme_multi_step(deme_multi* m, uint32_t nsteps);
int deme_multi_sync(deme_multi* m);
int deme_multi_download_state(deme_multi* m, DemeOwnerState* global, uint32_t nOwnersGlobal);
int deme_multi_upload_state(deme_multi* m, const DemeOwnerState* global, uint32_t nOwnersGlobal);
/* contacts of all slabs together (a contact that straddles a cut is on two lists unless deme_halo_group_set_cross_contacts(1):
 * `nContacts` counts list entries), steps taken, detections of the first slab, clumps that changed slabs so far */
int deme_multi_counts(deme_multi* m, DemeCounts* sum, uint64_t* clumpsMigrated);
/* back to the state after deme_multi_create (slabs, contexts and plan dropped; the groups opened anew): a scene re-upload */
int deme_multi_reset(deme_multi* m);
/* The contact list of a decomposed run in GLOBAL sphere ids (analytical-component / triangle ids are global already): a pair that
 * straddles a cut is reported once, sphere-sphere pairs as (smaller, larger) id, rows in the canonical order of an undivided
 * context (A, then class, then B).  Per-contact wildcards and recorded forces / contact points follow in the same order; where a slab
 * held a pair the other way round, its B -> A vector wildcards (the flipMask given at build) and its recorded force change sign and
 * its two contact points swap.  The list is rebuilt after every step / state upload. */
int deme_multi_num_contacts(deme_multi* m, size_t* n);
int deme_multi_download_contacts(deme_multi* m, uint32_t* idA, uint32_t* idB, uint8_t* type, size_t cap);
int deme_multi_download_contact_wildcard(deme_multi* m, uint32_t w, float* out, size_t cap);
int deme_multi_download_contact_records(deme_multi* m, float* force, float* torqueOnly, float* cpA, float* cpB, size_t cap);
/* Restart and marked pairs of a decomposed run, in GLOBAL ids: deme_seed_contacts / deme_*_persistent_contacts for every slab that holds
 * a pair (both geometries present, one of them the slab's own), in the slab's ids and with the slab's sign of the B -> A vector
 * wildcards; the downloads report a pair once (sphere-sphere pairs smaller id first, ascending). */
int deme_multi_seed_contacts(deme_multi* m, const uint32_t* idA, const uint32_t* idB, const uint8_t* type, const float* wildcards, size_t n);
int deme_multi_mark_persistent_contacts(deme_multi* m, int mode, uint32_t N1, uint32_t N2, int mark);
int deme_multi_num_persistent_contacts(deme_multi* m, size_t* n);
int deme_multi_download_persistent_contacts(deme_multi* m, uint32_t* idA, uint32_t* idB, uint8_t* type, size_t cap);
int deme_multi_upload_persistent_contacts(deme_multi* m, const uint32_t* idA, const uint32_t* idB, const uint8_t* type, size_t n);
/* deme_upload_contact_wildcard for the merged list (n = deme_multi_num_contacts): every slab's copy of a pair takes the value */
int deme_multi_upload_contact_wildcard(deme_multi* m, uint32_t w, const float* in, size_t n);
/* Moving the slab boundaries: equal counts again from the clumps' CURRENT coordinates (snapped to bin faces), then the migration of
 * deme_halo_group_migrate moves the clumps that now lie beyond a boundary.  One call moves a boundary by less than the narrower of its
 * two slabs minus the halo (a clump travels to a face neighbour only); newEdges (nSlabs + 1, may be NULL) receives the boundaries in
 * force afterwards.  deme_multi_slab_counts: {own, ghosts below, ghosts above, owners, spheres, seeded contacts} and the range of a slab. */
int deme_multi_rebalance(deme_multi* m, uint32_t* clumpsMoved, double* newEdges);
/* every n-th migration that deme_multi_step finds due (deme_multi_set_migration) recomputes the boundaries first; 0: never (default) */
int deme_multi_set_rebalance(deme_multi* m, uint32_t everyNthMigration);
int deme_multi_slab_counts(deme_multi* m, uint32_t slab, uint32_t counts[6], double range[2]);
/* deme_download / _upload_wildcard_array by GLOBAL id (kind 0: per owner, 1: per sphere; n = the global scene's count): a row is read
 * from the slab that owns the clump (a replicated owner's from the first slab) and written to every copy.  Kinds 2, 3 (triangles,
 * analytical components: replicated geometry) are written to every slab and read back from the first one.  The library's
 * migration carries owner and sphere wildcard arrays with the clumps (deme_halo_group_migrate). */
int deme_multi_download_wildcard_array(deme_multi* m, uint32_t kind, uint32_t index, float* out, size_t cap);
int deme_multi_upload_wildcard_array(deme_multi* m, uint32_t kind, uint32_t index, const float* in, size_t n);
/* deme_inspect_values by GLOBAL id (per sphere for the sphere-wise quantities, per owner otherwise) */
int deme_multi_inspect_values(deme_multi* m, uint32_t quantity, float* out, size_t cap);
/* deme_add_owner_acc by GLOBAL owner id: a clump's entry goes to the slab that owns it, a replicated owner's to every slab */
int deme_multi_add_owner_acc(deme_multi* m, uint32_t owner, uint32_t n, const float* acc, const float* angAcc);
/* the visible HIP devices (0 and DEME_OK where there is none: what the constructors check ids against) */
int deme_device_count(int* n);
/* Measurement aid (quote the attainable rate beside the nominal 8 TB/s): a hand-written 16 B / lane streaming copy of
 * `bytes` (choose well beyond the 256 MiB Infinity Cache), best of `reps` timings, read + written bytes per second in GB/s. */
int deme_copy_rate_probe(int device, size_t bytes, int reps, double* GBs);
/* {owners, clump owners, spheres, contact wildcards} of the scene a context holds (a slab's change when clumps migrate) */
int deme_scene_sizes(const deme_ctx* ctx, uint32_t out[4]);

#ifdef __cplusplus
}
#endif
#endif /* DEME_HIP_H */
