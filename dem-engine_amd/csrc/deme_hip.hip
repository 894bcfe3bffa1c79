// deme_hip.hip -- host orchestration (C++) and the C-ABI of include/deme_hip.h.
//
// One context = one GPU = one HIP stream.  The reference splits this work between a kinematic
// thread (DEM/kT.cpp) and a dynamic thread (DEM/dT.cpp) that exchange device buffers under mutexes;
// here contact detection and stepping are phases on the same stream, sized through two small
// device->host reads per detection (the reference makes >=10, DEMCubContactDetection.cu:121-905).
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <dlfcn.h>
#include <map>
#include <string>
#include <thread>
#include <vector>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
// Radix-sort configurations measured on MI355X for the detection's list sizes (a few million entries: the library's defaults are
// tuned for lists that fill the chip many times over; at these sizes workgroups of 1024 with 8 keys each shorten the chained
// look-back of every pass).  tools/rsbench.hip, profiles/r04/r04j_radix_configs.txt: incidences (8.2e6 pairs, 21 bits) 227 -> 206 us (4.4e6: 192 -> 149),
// crossing records by B owner (1.4e6 pairs, 20 bits, two 10-bit passes) 108 -> 68 us, contact keys (4.3e6 u64, 24 bits) 150 -> 141 us.
template <unsigned RB, unsigned IPT = 8>
using DemeRadixCfg = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                                rocprim::radix_sort_onesweep_config<rocprim::kernel_config<1024, IPT>, rocprim::kernel_config<1024, IPT>, RB,
                                                                                    rocprim::block_radix_rank_algorithm::match>,
                                                1024 * 1024>;
#include <array>
#include <iterator>
#include <limits>

#include "../../include/deme_hip.h"
#include "deme_device.h"
#include "deme_force.h"
#include "deme_force_fast.h"
#include "deme_jit.h"
#include "deme_kernels.h"
#include "deme_tile.h"
#ifndef DEME_TILE_P_PARTS
#define DEME_TILE_P_PARTS 32u  // (deme_tile_p.h)
#endif
namespace deme_dev {
int launch_tile_forces_p(int which, unsigned nCU, unsigned ldsBytes, hipStream_t st, const DevParams& dp, const TileArgs& ta);  // deme_tile_p.hip
}
#include "deme_tile_step.h"
#include "deme_migrate.h"
#include "deme_mesh_kernels.h"

using namespace deme_dev;

namespace {

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    template <typename T>
    T* as() const {
        return reinterpret_cast<T*>(p);
    }
};

struct TimerSlot {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
    double total_ms = 0;
    uint64_t launches = 0;
    uint64_t seen = 0;  // launches since the last reset, timed or not
};

}  // namespace

struct deme_ctx {
    int device = 0;
    int nCU = 256;  // compute units of the device (how many persistent workgroups a launch holds)
    hipStream_t stream = nullptr;
    bool ownStream = true;
    std::string err;
    DemeParams hp{};
    bool haveParams = false, haveScene = false;
    DevParams dp{};
    uint32_t nOwners = 0, nOwnerClumps = 0, nSpheres = 0, nAnal = 0, nMat = 0, nComp = 0, nMassProps = 0;
    // model
    DevBuf owners, spheres, acc, comp, massProps, anal, matPair, E, nu, CoR, mu, Crr, famMasks, famExtra, famFlags;
    std::vector<uint8_t> hObjType;  // host copy for contact-type decoding on download
    // detection scratch
    DevBuf sphFam;  // per sphere: its owner's family word, for the sweeps (written by k_sphere_prep when masks, margins or ghosts are in play)
    DevBuf geo, binLo, binN, counts, offsets, incKeys[2], incVals[2], keysRaw, keysMid, keysSorted[2], mapping, wc[2], ctr, segCtr,
        scanTmp, sortTmp, rec[4], stage;
    // per-contact contributions and the per-owner gather lists (built once per detection)
    DevBuf conA4, conA2, conB4, conB2, aSum, ownerA, ownerB[2], bIdx[2], aStart, bStart, heavy, fixedFlag, heavyList, rangeCtr;
    uint32_t nHeavy = 0, nHeavyFree = 0, nSA = 0, nSM = 0;
    DevBuf info;
    // owner-tile form of the force pass (deme_tile.h), rebuilt per detection
    DevBuf tileCtr;  // k_tile_forces_p: (tiles handed out, workgroups through) per pass; zero between launches
    DevBuf tileBig, bigList, tInfo, hList, hCount, tileMode, tileOrg, rIdx, rStart, remKey[2], remVal, lPos, lOff, lCount, tileRem, tileBase, rankC, rec32;
    // the heavy-owner counts of a tiled list are fetched without stopping the stream: the copy lands in pinned memory, the first
    // reader (launch_reduce_heavy, one force launch later) waits for its event
    RangeCounters* hrPinned = nullptr;
    hipEvent_t hrEvent = nullptr;
    bool hrPending = false, heavyOverflow = false;
    bool ctrFresh = false;  // DetectCounters were zeroed by the margins kernel's launch and nothing has counted in them since
    uint64_t nListed = 0;      // contacts of the list the owner arrays (ownerA, ownerB[0], info ...) were built for
    int listKeys = 0;
    bool legacyLists = false;  // the B-sorted list of the round-2 kernels exists for the current contact list (built on demand when the list has tile structures)
    bool tileActive = false;  // the current list has tile structures (built-in model, fast mode, every halo fits)
    bool conTile = false;     // the contributions in memory were written by the tile kernel
    int tileEnable = 1;       // DEME_TILE=0 keeps the round-2 kernels (A/B measurements)
    // the whole step in one kernel (deme_tile_step.h): closed tiles -- a contact that straddles two tiles is evaluated by both --
    // integrate their own owners; owners and history are double-buffered
#define DEME_FUSED_AUTO_MAX_OWNERS 100000u  // deme_set_fused_step(ctx, 2): the one-kernel step where it was measured faster (header)
    int fusedEnable = 0;          // deme_set_fused_step / DEME_FUSED=1: the one-kernel step (measured slower on the packed bed: DESIGN 3.7)
    bool fusedList = false;       // the current list has the incoming structures (decided per detection)
    bool fusedChecked = false;    // ... and the device has been asked whether every closed tile fits
    bool fusedPrevValid = false;  // the last step was a fused one: the buffers of its start are intact (a / alpha can be replayed from them)
    uint64_t nFusedSteps = 0;
    DevBuf ownersNext, recContact, inCnt, inStart, tInfoIn, inContact, hCountIn;
    // A run-time compiled model takes the tile pass only when a tile holds enough contacts to pay for its staging (measured on
    // configs[4], 1e6 single spheres with 1.6 contacts each = 200 per tile: tile pass 0.058 ms against 0.044 ms for the general
    // kernel; at the 555 per tile of three-sphere clumps the tile pass wins as it does for the built-in models).
    uint32_t tileMinContactsCustom = 320;
    uint64_t lastSegMax = 0;             // longest segment of the last detection (sizes the early launch of k_compact_keys)
    size_t keySegMin = (size_t)1 << 20;  // arenas from this many slots on are cut into DEME_KEY_SEGS segments (DEME_KEY_SEG_MIN; 0: never)
    uint32_t tileMaxHalo = 0, tileMaxList = 0, nBigTiles = 0, tileMaxHaloIn = 0;
    // persistent contacts: the sorted set of marked keys (host copy + device copy appended to every detection's raw keys)
    DevBuf persistKeys, binStat;
    // acceleration the script adds for the next step only (deme_add_owner_acc): device records + host mirror
    DevBuf nextAcc;
    std::vector<AccRec> hNextAcc;
    bool nextAccPending = false;
    std::vector<uint64_t> hPersist;
    // triangles (mesh path)
    uint32_t nTri = 0;
    DevBuf tris, triWorld, triLo, triHi, triCounts, triOffsets, triKeys[2], triVals[2];
    size_t triCap = 0;
    uint64_t nTriInc = 0;
    // run-time compiled user force model
    deme_jit::MaterialTables mt;
    hipModule_t customMod = nullptr;
    hipFunction_t customFn[3] = {nullptr, nullptr, nullptr};
    hipFunction_t customTileFn[4] = {nullptr, nullptr, nullptr, nullptr};  // the tile pass with the user's model (deme_jit::kTileEntry)
    std::map<size_t, std::vector<char>> jitCache;
    bool conValid = false;
    int keysCur = 0, wcCur = 0;
    size_t incCap = 0, cntCap = 0;
    uint64_t nInc = 0, nContacts = 0, nPrev = 0, nWcStored = 0;
    uint64_t nActiveBins = 0;
    uint32_t maxInBin = 0;
    bool haveList = false, mapFresh = false;
    bool seeded = false;
    // family motion prescriptions: run-time compiled kernel + the owners it applies to
    hipModule_t prescMod = nullptr;
    hipFunction_t prescFn = nullptr;
    DevBuf prescList, prescSlot, prescRec, smFlag, smList, cDefer, blockMode;
    DevBuf userWc[4][8];  // [kind: owners, spheres, triangles, analytical][index]
    uint32_t nOwnerWc = 0, nGeoWc = 0;
    bool hasGhosts = false;  // a family carries DEME_FAMILY_GHOST: force passes can be split for the halo overlap
    // asynchronous detection (deme_set_async_detection): part 1 of a detection on its own stream, from a snapshot of the owners,
    // `asyncLead` steps before the list it builds is swapped in
    uint32_t asyncLead = 0;
    // (slab group) one evaluation per cross-cut contact: where an own clump finds the a / alpha the left neighbour evaluated for it
    DevBuf revSlot;
    const void* revAcc = nullptr;
    bool pairsOnce = false;
    PrescArgs laterPa{nullptr, nullptr};  // the prescription records of the step whose integration is split (launch_integrate_later)
    bool crossStale = false;  // a scene was uploaded while the group evaluates cross-cut contacts once: rev_setup_slab runs again first
    hipEvent_t evPass1 = nullptr;
    char* pin = nullptr;       // 16 KB of pinned host memory: where the detection's read-backs land
    int spinSync = 1;          // DEME_SPIN_SYNC=0: blocking waits at the detection's read-backs
    int pass1Beside = 0;       // DEME_PASS1_BESIDE=1: the ghost-dependent force pass on the halo stream, beside the tail of the interior
                               // pass (off: on the one-GPU harness -- two slabs competing for one GPU -- it costs 5 %; not measured with
                               // one slab per GPU, where the interior pass leaves the GPU to a few per cent of the tiles)
    bool listOwnersSnap = false;  // (asynchronous detection) part 2 runs beside the steps too: it reads the snapshot, writes the spare set
    DevBuf spare[40];              // the second set of the list structures (list_set): the steps in flight read one, part 2 builds the other
    bool snapPending = false;  // (slab group) take the owner snapshot of an asynchronous detection in this step, once the ghosts are in place
    DevBuf ownersSnap;
    hipStream_t detStream = nullptr;
    hipEvent_t evSnap = nullptr, evP1 = nullptr;
    uint64_t nAsyncDetections = 0;
    std::vector<uint32_t> hShared;  // replicated free owners (DemeScene.ownerGhost bit 1): their a / alpha are summed across slabs
    DevBuf sharedIds, sharedBuf;
    bool tailFused = true;
    bool inGroupStep = false;
    hipStream_t haloStream = nullptr;
    hipEvent_t evStepDone = nullptr, evHaloDone = nullptr;
    bool overlapDetect = false;  // the step opened by deme_step_overlap_begin needs a detection first
    hipModule_t rulesMod = nullptr;
    // adaptive controllers (deme_set_adaptive)
    DemeAdaptive ad{};
    hipEvent_t evDet0 = nullptr, evDet1 = nullptr, evWin0 = nullptr, evWin1 = nullptr;
    double binAccMs = 0, binPrevMs = -1, freqPrevMs = -1;
    float binRate = 0.f;
    uint32_t binObs = 0, freqObs = 0, nBinChanges = 0, nFreqChanges = 0;
    uint64_t winStartStep = 0;
    int freqDir = 1, binFlip = 1;
    bool winOpen = false;
    // inspector region filters: one module per compiled region (id = index)
    std::vector<hipModule_t> regionMod;
    std::vector<hipFunction_t> regionFn;
    DevBuf volumes;  // per mass-property entry: declared clump volume ("clump_volume" inspector)
    bool haveVolumes = false;
    hipFunction_t rulesFn = nullptr;  // on-the-fly family changes
    bool rulesNeedAcc = false;
    uint32_t nPresc = 0;
    uint8_t hostFamFlags[DEME_NUM_FAMILIES] = {0};
    bool prescDirty = true;  // owner -> family assignment changed: rebuild the list  // the list was loaded by deme_seed_contacts: it only feeds the next history map
    bool record = false;
    uint64_t nSteps = 0, nDetections = 0;
    uint32_t stepsSinceCD = 0;
    // the script moved a body, changed a family or rewrote triangle nodes: the K-step list and its margins no longer cover the
    // scene, so the next step starts with a detection (the reference: pendingCriticalUpdate / stampLastDynamicUpdateProdDate = -1,
    // DEM/dT.cpp:2351)
    bool listStale = false;
    uint32_t lastStatus = 0;
    double timeElapsed = 0;
    // arithmetic mode (deme_set_arith_mode): DEME_ARITH_FAST (default) or DEME_ARITH_EXACT
    int arith = DEME_ARITH_FAST;
    uint32_t xcdGroup = 0;  // XCD-aware block order of the force kernel (ForceArgs::xcdGroup)
    // engine-side spatial order (deme_order.inc): slot -> caller id and back, for owners and spheres (empty: the caller's order)
    bool permuted = false;
    int reorderEnable = 1;
    std::vector<uint32_t> hO2E, hE2O, hS2E, hE2S;
    DevBuf dO2E, dS2E;
    double orderSpread[2] = {0, 0};  // cells per tile bounding box: in the caller's order, along the curve
    // renewing the order at run time (order_renew): the engine watches the tiles of every detection -- mean foreign owners per
    // tile, tiles that no longer fit -- against what they were right after the last (re)ordering
    bool orderEligible = false;     // the scene may be reordered (fast mode, no ghosts: what order_decide checks)
    bool orderStructOK = false;     // ... the part of that which does not depend on the arithmetic mode
    bool geoStale = false;          // the per-sphere geometry of the last detection sits in slots an order renewal has left
    bool orderRenewDue = false;
    uint32_t orderBaseHalo = 0;     // mean halo (x 16) at the first detection after the last ordering; 0 = not taken yet
    uint64_t orderDetAt = 0, nOrderRenewals = 0;
    uint64_t listSerial = 0, viewSerial = ~0ull;  // the caller's view of the current list (ids and order), cached per list
    std::vector<uint64_t> viewKeys;
    std::vector<uint32_t> viewPerm;
    int timing = 0;  // 0 off; n > 0: every n-th launch of each timed kernel is bracketed with HIP events
    std::map<std::string, TimerSlot> timers;
    std::vector<hipEvent_t> eventPool;
};

namespace {

int fail(deme_ctx* c, int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (c) {
        c->err = buf;
        c->lastStatus = (uint32_t)code;
    }
    return code;
}

#define HIPCK(call)                                                                                   \
    do {                                                                                              \
        hipError_t _e = (call);                                                                       \
        if (_e != hipSuccess)                                                                         \
            return fail(c, DEME_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

// The sizing read-backs of a detection: the host has nothing else to do, the GPU idles until it is back with the next launches,
// and a blocking wait wakes up tens of microseconds late -- so the host polls the stream (for at most a few milliseconds; a
// detection beside the steps, whose wait is long by design, blocks instead of burning a core).
template <typename T>
static inline T* pin_at(deme_ctx* c, size_t off) {
    return reinterpret_cast<T*>(c->pin + off);
}
static hipError_t sync_readback(deme_ctx* c, hipStream_t st) {
    if (c->spinSync && st == c->stream) {
        const auto t0 = std::chrono::steady_clock::now();
        for (int it = 0;; it++) {
            const hipError_t q = hipStreamQuery(st);
            if (q != hipErrorNotReady)
                return q;
            if ((it & 255) == 255 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(5))
                break;
        }
    }
    return hipStreamSynchronize(st);
}

int ensure(deme_ctx* c, DevBuf& b, size_t bytes, bool keep = false) {
    if (bytes <= b.bytes && b.p)
        return DEME_OK;
    size_t want = std::max<size_t>(bytes, 256);
    if (b.p)  // a buffer that grows once usually grows again (a settling bed: every detection a few more contacts): leave room
        want = std::max(want, b.bytes + b.bytes / 4);
    void* np = nullptr;
    HIPCK(hipMalloc(&np, want));
    if (keep && b.p && b.bytes)
        HIPCK(hipMemcpyAsync(np, b.p, b.bytes, hipMemcpyDeviceToDevice, c->stream));
    if (b.p) {
        HIPCK(hipStreamSynchronize(c->stream));
        HIPCK(hipFree(b.p));
    }
    b.p = np;
    b.bytes = want;
    return DEME_OK;
}

template <typename T>
int upload(deme_ctx* c, DevBuf& b, const T* src, size_t n) {
    int rc = ensure(c, b, std::max<size_t>(n, 1) * sizeof(T));
    if (rc)
        return rc;
    if (n && src)
        HIPCK(hipMemcpyAsync(b.p, src, n * sizeof(T), hipMemcpyHostToDevice, c->stream));
    else if (n)
        HIPCK(hipMemsetAsync(b.p, 0, n * sizeof(T), c->stream));
    return DEME_OK;
}

inline unsigned grid_for(size_t n, unsigned block = 256) { return (unsigned)((n + block - 1) / block); }

// ---- kernel timing (HIP events on the context stream) ------------------------------------------
hipEvent_t get_event(deme_ctx* c) {
    if (!c->eventPool.empty()) {
        hipEvent_t e = c->eventPool.back();
        c->eventPool.pop_back();
        return e;
    }
    hipEvent_t e;
    hipEventCreate(&e);
    return e;
}
struct ScopedTimer {
    deme_ctx* c;
    TimerSlot* slot = nullptr;
    hipEvent_t a = nullptr, b = nullptr;
    hipStream_t st = nullptr;
    ScopedTimer(deme_ctx* ctx, const char* name, bool always = false, hipStream_t stream = nullptr) : c(ctx), st(stream ? stream : ctx->stream) {
        if (!c->timing)
            return;
        slot = &c->timers[name];
        if (!always && (slot->seen++ % (uint64_t)c->timing) != 0) {  // sampled: an event pair costs ~3 us of dispatch gap per launch
            slot = nullptr;
            return;
        }
        if (slot->pending.size() >= 8192) {  // bounded bookkeeping
            slot = nullptr;
            return;
        }
        a = get_event(c);
        b = get_event(c);
        hipEventRecord(a, st);
    }
    ~ScopedTimer() {
        if (slot) {
            hipEventRecord(b, st);
            slot->pending.emplace_back(a, b);
        }
    }
};
void drain_timers(deme_ctx* c) {
    for (auto& kv : c->timers) {
        for (auto& pr : kv.second.pending) {
            hipEventSynchronize(pr.second);
            float ms = 0;
            hipEventElapsedTime(&ms, pr.first, pr.second);
            kv.second.total_ms += ms;
            kv.second.launches++;
            c->eventPool.push_back(pr.first);
            c->eventPool.push_back(pr.second);
        }
        kv.second.pending.clear();
    }
}

void refresh_dev_params(deme_ctx* c) {
    DevParams& d = c->dp;
    const DemeParams& h = c->hp;
    d.nvXp2 = h.nvXp2, d.nvYp2 = h.nvYp2;
    d.nbX = h.nbX, d.nbY = h.nbY, d.nbZ = h.nbZ;
    d.mNbX = fast_div_magic(h.nbX), d.mNbXY = fast_div_magic(h.nbX * h.nbY);
    d.l = h.l, d.voxelSize = h.voxelSize, d.binSize = h.binSize;
    d.LBFX = h.LBFX, d.LBFY = h.LBFY, d.LBFZ = h.LBFZ;
    d.Gx = h.Gx, d.Gy = h.Gy, d.Gz = h.Gz;
    d.h = h.h;
    d.approxMaxVel = h.approxMaxVel, d.expSafetyMulti = h.expSafetyMulti, d.expSafetyAdder = h.expSafetyAdder;
    d.errOutVel = h.errOutVel;
    d.integrator = h.integrator, d.forceModel = h.forceModel, d.nW = h.nContactWildcards;
    d.nOwners = c->nOwners, d.nSpheres = c->nSpheres, d.nAnal = c->nAnal, d.nMat = c->nMat;
    d.errOutBinSphNum = h.errOutBinSphNum ? h.errOutBinSphNum : 32768u;
    d.comp = c->comp.as<float4>();
    d.massProps = c->massProps.as<float4>();
    d.anal = c->anal.as<AnalObj>();
    d.matPair = c->matPair.as<MatPair>();
    d.E = c->E.as<float>(), d.nu = c->nu.as<float>(), d.CoR = c->CoR.as<float>(), d.mu = c->mu.as<float>(),
    d.Crr = c->Crr.as<float>();
    d.familyMasks = c->famMasks.as<uint8_t>();
    d.familyExtra = c->famExtra.as<float>();
    d.familyFlags = c->famFlags.as<uint8_t>();
    d.tris = c->tris.p;
    d.nTri = c->nTri;
    d.s2e = c->permuted ? c->dS2E.as<uint32_t>() : nullptr;
    d.o2e = c->permuted ? c->dO2E.as<uint32_t>() : nullptr;
}

int check_ready(deme_ctx* c) {
    if (!c)
        return DEME_ERR_INVALID;
    if (!c->haveParams || !c->haveScene)
        return fail(c, DEME_ERR_INVALID, "deme_set_params and deme_upload_scene must be called first");
    // (a process may hold contexts on several devices -- deme_multi: every entry point works on the context's own device)
    if (hipSetDevice(c->device) != hipSuccess)
        return fail(c, DEME_ERR_HIP, "device %d cannot be selected", c->device);
    return DEME_OK;
}

// per-material-pair constants of the Hertzian models, evaluated once on the host with the formulas
// of kernel/DEMHelperKernels.cuh:433-445 (matProxy2ContactParam<float>) and
// FullHertzianForceModel.cu:56-57 (loge, beta).  Host overload resolution (log/sqrt in double) as in
// the pinned reference build.
void build_mat_pairs(const DemeScene* s, std::vector<MatPair>& out) {
    const uint32_t n = s->nMat;
    out.resize((size_t)n * n);
    for (uint32_t a = 0; a < n; a++)
        for (uint32_t b = 0; b < n; b++) {
            MatPair m{};
            const float Y1 = s->E[a], nu1 = s->nu[a], Y2 = s->E[b], nu2 = s->nu[b];
            const float invE = (1.0f - nu1 * nu1) / Y1 + (1.0f - nu2 * nu2) / Y2;
            m.E_cnt = 1.0f / invE;
            const float invG = 2.0f * (2.0f - nu1) * (1.0f + nu1) / Y1 + 2.0f * (2.0f - nu2) * (1.0f + nu2) / Y2;
            m.G_cnt = 1.0f / invG;
            m.CoR = s->CoR ? s->CoR[a * n + b] : 0.f;
            m.mu = s->mu ? s->mu[a * n + b] : 0.f;
            m.Crr = s->Crr ? s->Crr[a * n + b] : 0.f;
            const float loge = (float)((m.CoR < 1e-12) ? log(1e-12) : logf(m.CoR));  // log(float) -> float overload
            m.beta = (float)(loge / sqrt(loge * loge + 9.869604401089358));
            out[(size_t)a * n + b] = m;
        }
}

int grow_contact_arena(deme_ctx* c, size_t cap) {
    const uint32_t nW = std::max<uint32_t>(c->hp.nContactWildcards, 1);
    int rc = 0;
    rc |= ensure(c, c->keysRaw, cap * 8);
    // sorted lists and wildcards carry state between detections: keep their contents
    rc |= ensure(c, c->keysSorted[0], cap * 8, true);
    rc |= ensure(c, c->keysSorted[1], cap * 8, true);
    rc |= ensure(c, c->mapping, cap * 4, true);
    rc |= ensure(c, c->wc[0], cap * 4 * nW, true);
    rc |= ensure(c, c->wc[1], cap * 4 * nW, true);
    if (c->record)
        for (int k = 0; k < 4; k++)
            rc |= ensure(c, c->rec[k], cap * 12);
    rc |= ensure(c, c->conA4, cap * 16);
    rc |= ensure(c, c->conA2, cap * 8);
    rc |= ensure(c, c->conB4, cap * 16);
    rc |= ensure(c, c->conB2, cap * 8);
    rc |= ensure(c, c->ownerA, cap * 4);
    for (int k = 0; k < 2; k++) {
        rc |= ensure(c, c->ownerB[k], cap * 4);
        rc |= ensure(c, c->bIdx[k], cap * 4);
    }
    rc |= ensure(c, c->info, cap * 16);
    rc |= ensure(c, c->tInfo, cap * 8);
    rc |= ensure(c, c->rIdx, cap * 4);
    rc |= ensure(c, c->lPos, cap * 2);
    rc |= ensure(c, c->remVal, (cap + 1) * 4);
    rc |= ensure(c, c->rankC, (cap + 1) * 4);
    rc |= ensure(c, c->recContact, (cap + 1) * 4);
    rc |= ensure(c, c->tInfoIn, cap * 8);
    rc |= ensure(c, c->inContact, cap * 4);
    rc |= ensure(c, c->rec32, cap * 32);
    rc |= ensure(c, c->remKey[0], (cap + 1) * 4);
    rc |= ensure(c, c->remKey[1], (cap + 1) * 4);
    if (c->hasGhosts) {
        rc |= ensure(c, c->cDefer, cap);
        rc |= ensure(c, c->blockMode, (cap / DEME_FORCE_BLOCK + 2) * 4);
    }
    if (c->nTri) {
        rc |= ensure(c, c->smFlag, cap);
        rc |= ensure(c, c->smList, cap * 4);
    }
    if (rc)
        return rc;
    c->cntCap = cap;
    return DEME_OK;
}

int grow_incidence_arena(deme_ctx* c, size_t cap) {
    int rc = 0;
    for (int k = 0; k < 2; k++) {
        rc |= ensure(c, c->incKeys[k], cap * 4);
        rc |= ensure(c, c->incVals[k], cap * 4);
    }
    if (rc)
        return rc;
    c->incCap = cap;
    return DEME_OK;
}

int status_to_error(deme_ctx* c, uint32_t st) {
    if (st & DEME_ST_NONFINITE)
        return fail(c, DEME_ERR_VELOCITY, "a non-finite owner velocity was found at time %.9g", c->timeElapsed);
    if (st & DEME_ST_VELOCITY)
        return fail(c, DEME_ERR_VELOCITY,
                    "system max velocity exceeded the error-out velocity %.7g at time %.9g (SetErrorOutVelocity)",
                    (double)c->hp.errOutVel, c->timeElapsed);
    return DEME_OK;
}

int do_margins(deme_ctx* c, uint32_t drift) {
    HIPCK(hipMemsetAsync(c->ctr.p, 0, sizeof(DetectCounters), c->stream));
    c->ctrFresh = true;
    hipLaunchKernelGGL(k_margins, dim3(grid_for(c->nOwners)), dim3(256), 0, c->stream, c->dp, c->owners.as<OwnerRec>(),
                       drift, c->ctr.as<DetectCounters>());
    return DEME_OK;
}

// contactDetection() equivalent, in two parts.  Part 1 -- everything up to the sorted key list of the new contacts -- reads the
// owners through `ow` and touches detection scratch only, so it can run on its own stream from a snapshot of the owner records
// while the main stream keeps stepping with the current list (deme_set_async_detection; the reference's kT does the same on its
// own GPU thread).  Part 2 -- history map and the per-owner gather lists the force and integration kernels read -- runs on the
// main stream when the list is swapped in.
int do_migrate(deme_ctx* c);
int detect_part1(deme_ctx* c, hipStream_t st, OwnerRec* ow, bool async, uint64_t* nCout) {
    const uint32_t nS = c->nSpheres;
    DetectCounters hc{};
    // The margins kernel zeroes the counters and may leave status bits (velocity limit, non-finite state): when it has just run the
    // block is not zeroed again and its status comes back with the first sizing read-back below (no read-back of its own).
    const bool countersFresh = c->ctrFresh;
    c->ctrFresh = false;
    bool statusSeen = false;

    if (int rc = ensure(c, c->segCtr, DEME_KEY_SEGS * DEME_KEY_SEG_STRIDE * 8))
        return rc;
    for (int attempt = 0; attempt < 4; attempt++) {
        if (attempt > 0 || !countersFresh)
            HIPCK(hipMemsetAsync(c->ctr.p, 0, sizeof(DetectCounters), st));
        // the raw key arena: 64 segments with a counter each once it is large enough for every segment to hold a fair share of any
        // list (see KeyArena); one segment, counted in DetectCounters::nContactsRaw, below that
        const bool segmented = c->keySegMin && c->cntCap >= c->keySegMin && c->cntCap >= DEME_KEY_SEGS * 64u;
        KeyArena ar;
        ar.keys = c->keysRaw.as<uint64_t>();
        ar.segMask = segmented ? DEME_KEY_SEGS - 1u : 0u;
        ar.segCap = segmented ? (uint64_t)c->cntCap / DEME_KEY_SEGS : (uint64_t)c->cntCap;
        ar.ctr = segmented ? c->segCtr.as<unsigned long long>() : &c->ctr.as<DetectCounters>()->nContactsRaw;
        if (segmented)
            HIPCK(hipMemsetAsync(c->segCtr.p, 0, DEME_KEY_SEGS * DEME_KEY_SEG_STRIDE * 8, st));
        if (nS) {
            c->geoStale = false;
            hipLaunchKernelGGL(k_sphere_prep, dim3(grid_for(nS)), dim3(256), 0, st, c->dp,
                               ow, c->spheres.as<SphereRec>(), c->geo.as<GeoRec>(),
                               c->binLo.as<uint4>(), c->binN.as<uint2>(), c->counts.as<uint32_t>(),
                               ar, c->ctr.as<DetectCounters>(), c->sphFam.as<uint16_t>());
            size_t tmp = c->scanTmp.bytes;
            HIPCK(rocprim::exclusive_scan(c->scanTmp.p, tmp, c->counts.as<uint32_t>(), c->offsets.as<uint32_t>(), 0u,
                                          (size_t)nS + 1, rocprim::plus<uint32_t>(), st));
        }
        uint32_t P = 0;
        bool filled = false;
        if (nS && c->incCap) {  // the incidences go out before the host knows how many there are (the kernel guards the arena's end):
                                // the GPU works through the read-back below instead of idling; repeated if the arena has to grow
            hipLaunchKernelGGL(k_fill_incidence, dim3(grid_for(nS)), dim3(256), 0, st, c->dp,
                               c->binLo.as<uint4>(), c->binN.as<uint2>(), c->offsets.as<uint32_t>(),
                               c->incKeys[0].as<uint32_t>(), c->incVals[0].as<uint32_t>(), (uint64_t)c->incCap);
            filled = true;
        }
        if (nS) {
            // (read-backs land in pinned memory: a copy into pageable memory goes through the runtime's staging buffer and costs
            // tens of microseconds of host time before the wait even starts)
            HIPCK(hipMemcpyAsync(pin_at<uint32_t>(c, 0), c->offsets.as<uint32_t>() + nS, 4, hipMemcpyDeviceToHost, st));
            HIPCK(hipMemcpyAsync(pin_at<DetectCounters>(c, 64), c->ctr.p, sizeof(hc), hipMemcpyDeviceToHost, st));
            HIPCK(sync_readback(c, st));
            P = *pin_at<uint32_t>(c, 0), hc = *pin_at<DetectCounters>(c, 64);
            if (!statusSeen) {
                statusSeen = true;
                if (int rc = status_to_error(c, hc.status & ~DEME_ST_INCIDENCE))
                    return rc;
            }
            if (hc.status & DEME_ST_INCIDENCE)
                return fail(c, DEME_ERR_OVERFLOW,
                            "a sphere touches more than %u bins (margin far larger than the bin size): the incidence list cannot be built",
                            DEME_MAX_BINS_PER_SPHERE);
            if ((uint64_t)hc.maxCount * nS > 0xFFFFFFFFull) {  // the 32-bit offsets may have wrapped: exact total in 64 bits
                auto in64 = rocprim::make_transform_iterator(c->counts.as<uint32_t>(),
                                                             [] __host__ __device__(uint32_t v) { return (unsigned long long)v; });
                unsigned long long* tot = &c->ctr.as<DetectCounters>()->nContactsRaw;  // borrowed: saved and restored around the reduction (k_sphere_prep has already counted its sphere-analytical contacts here)
                unsigned long long keep = 0, total = 0;
                HIPCK(hipMemcpyAsync(&keep, tot, 8, hipMemcpyDeviceToHost, st));
                size_t need = 0;
                HIPCK(rocprim::reduce(nullptr, need, in64, tot, 0ull, (size_t)nS, rocprim::plus<unsigned long long>(), st));
                if (int rc = ensure(c, c->sortTmp, need))
                    return rc;
                need = c->sortTmp.bytes;
                HIPCK(rocprim::reduce(c->sortTmp.p, need, in64, tot, 0ull, (size_t)nS, rocprim::plus<unsigned long long>(), st));
                HIPCK(hipMemcpyAsync(&total, tot, 8, hipMemcpyDeviceToHost, st));
                HIPCK(hipStreamSynchronize(st));
                HIPCK(hipMemcpyAsync(tot, &keep, 8, hipMemcpyHostToDevice, st));
                HIPCK(hipStreamSynchronize(st));
                if (total > 0xFFFFFFFFull)
                    return fail(c, DEME_ERR_OVERFLOW, "%llu bin-sphere incidences do not fit the 32-bit list offsets (use larger bins)", total);
            }
        }
        if (P > c->incCap) {
            if (int rc = grow_incidence_arena(c, (size_t)P + P / 4 + 1024))
                return rc;
            filled = false;
        }
        c->nInc = P;
        int sortedIdx = 0;
        if (P) {
            if (!filled)
                hipLaunchKernelGGL(k_fill_incidence, dim3(grid_for(nS)), dim3(256), 0, st, c->dp,
                                   c->binLo.as<uint4>(), c->binN.as<uint2>(), c->offsets.as<uint32_t>(),
                                   c->incKeys[0].as<uint32_t>(), c->incVals[0].as<uint32_t>(), (uint64_t)c->incCap);
            const uint64_t nBins = (uint64_t)c->hp.nbX * c->hp.nbY * c->hp.nbZ;
            unsigned bits = 1;
            while (bits < 32 && (1ull << bits) < nBins)
                bits++;
            size_t need = 0;
            // bin ids of up to 22 bits: two passes of 11 bits (16 keys per thread) take as long as three of 8 and save a pass's launches
            const bool twoPass = bits > 16 && bits <= 22;
            auto sortInc = [&](void* tmp, size_t& bytes) {
                return twoPass ? rocprim::radix_sort_pairs<DemeRadixCfg<11, 16>>(tmp, bytes, c->incKeys[0].as<uint32_t>(), c->incKeys[1].as<uint32_t>(),
                                                                                 c->incVals[0].as<uint32_t>(), c->incVals[1].as<uint32_t>(), (size_t)P, 0, bits, st)
                               : rocprim::radix_sort_pairs<DemeRadixCfg<8>>(tmp, bytes, c->incKeys[0].as<uint32_t>(), c->incKeys[1].as<uint32_t>(),
                                                                            c->incVals[0].as<uint32_t>(), c->incVals[1].as<uint32_t>(), (size_t)P, 0, bits, st);
            };
            HIPCK(sortInc(nullptr, need));
            if (int rc = ensure(c, c->sortTmp, need))
                return rc;
            need = c->sortTmp.bytes;
            HIPCK(sortInc(c->sortTmp.p, need));
            sortedIdx = 1;
            const uint32_t nWin = (uint32_t)grid_for(P, SW_T);
            if (int rc = ensure(c, c->binStat, (size_t)nWin * sizeof(uint2)))
                return rc;
            static_assert(SW_WPB == 1, "k_sweep writes one statistics record per window");
            hipLaunchKernelGGL(k_sweep, dim3(nWin), dim3(SW_T), 0, st, c->dp,
                               c->incKeys[1].as<uint32_t>(), c->incVals[1].as<uint32_t>(), P, c->geo.as<GeoRec>(),
                               c->sphFam.as<uint16_t>(), ar, c->binStat.as<uint2>());
            hipLaunchKernelGGL(k_bin_stats_final, dim3((nWin + 2047u) / 2048u), dim3(256), 0, st, c->binStat.as<uint2>(), nWin,
                               c->ctr.as<DetectCounters>());
        }
        (void)sortedIdx;
        // ---- sphere-triangle contacts (only when a mesh is loaded and some sphere is registered in a bin)
        c->nTriInc = 0;
        if (c->nTri && P) {
            const uint32_t nT = c->nTri;
            hipLaunchKernelGGL(k_tri_prep, dim3(grid_for(nT)), dim3(256), 0, st, c->dp, nT, c->tris.as<TriRec>(),
                               ow, c->triWorld.as<TriWorld>(), c->triLo.as<int4>(),
                               c->triHi.as<int4>(), c->triCounts.as<uint32_t>());
            size_t tmp = c->scanTmp.bytes;
            HIPCK(rocprim::exclusive_scan(c->scanTmp.p, tmp, c->triCounts.as<uint32_t>(), c->triOffsets.as<uint32_t>(), 0u,
                                          (size_t)nT + 1, rocprim::plus<uint32_t>(), st));
            uint32_t TP = 0;
            HIPCK(hipMemcpyAsync(&TP, c->triOffsets.as<uint32_t>() + nT, 4, hipMemcpyDeviceToHost, st));
            HIPCK(hipStreamSynchronize(st));
            if (TP > c->triCap) {
                const size_t cap = (size_t)TP + TP / 4 + 1024;
                for (int k = 0; k < 2; k++)
                    if (ensure(c, c->triKeys[k], cap * 4) || ensure(c, c->triVals[k], cap * 4))
                        return c->lastStatus;
                c->triCap = cap;
            }
            c->nTriInc = TP;
            if (TP) {
                hipLaunchKernelGGL(k_tri_fill, dim3(grid_for(nT)), dim3(256), 0, st, c->dp, nT,
                                   c->triWorld.as<TriWorld>(), c->triLo.as<int4>(), c->triHi.as<int4>(),
                                   c->triOffsets.as<uint32_t>(), c->triKeys[0].as<uint32_t>(), c->triVals[0].as<uint32_t>(),
                                   (uint64_t)c->triCap);
                const uint64_t nBins = (uint64_t)c->hp.nbX * c->hp.nbY * c->hp.nbZ;
                unsigned bits = 1;
                while (bits < 32 && (1ull << bits) < nBins)
                    bits++;
                size_t need = 0;
                HIPCK(rocprim::radix_sort_pairs(nullptr, need, c->triKeys[0].as<uint32_t>(), c->triKeys[1].as<uint32_t>(),
                                                c->triVals[0].as<uint32_t>(), c->triVals[1].as<uint32_t>(), (size_t)TP, 0, bits,
                                                st));
                if (int rc = ensure(c, c->sortTmp, need))
                    return rc;
                need = c->sortTmp.bytes;
                HIPCK(rocprim::radix_sort_pairs(c->sortTmp.p, need, c->triKeys[0].as<uint32_t>(), c->triKeys[1].as<uint32_t>(),
                                                c->triVals[0].as<uint32_t>(), c->triVals[1].as<uint32_t>(), (size_t)TP, 0, bits,
                                                st));
                hipLaunchKernelGGL(k_tri_sweep, dim3(grid_for(TP)), dim3(256), 0, st, c->dp, TP,
                                   c->triKeys[1].as<uint32_t>(), c->triVals[1].as<uint32_t>(), c->triWorld.as<TriWorld>(), P,
                                   c->incKeys[1].as<uint32_t>(), c->incVals[1].as<uint32_t>(), c->geo.as<GeoRec>(),
                                   c->sphFam.as<uint16_t>(), ar);
            }
        }
        // the gaps between the arena's segments are closed while the host waits for the counts: a launch sized from the last
        // detection's longest segment, the rest (if a segment grew past that) after the read-back
        const int nextEarly = c->keysCur ^ 1;
        uint32_t compactBlocks = 0;
        if (segmented) {
            compactBlocks = (uint32_t)std::min<uint64_t>(grid_for(std::max<uint64_t>(c->lastSegMax + c->lastSegMax / 8 + 1024, 1)), grid_for(ar.segCap));
            hipLaunchKernelGGL(k_compact_keys, dim3(compactBlocks, DEME_KEY_SEGS), dim3(256), 0, st, c->keysRaw.as<uint64_t>(), ar.segCap,
                               c->segCtr.as<unsigned long long>(), 0u, c->keysSorted[nextEarly].as<uint64_t>());
        }
        HIPCK(hipMemcpyAsync(pin_at<DetectCounters>(c, 64), c->ctr.p, sizeof(hc), hipMemcpyDeviceToHost, st));
        const unsigned long long* hseg = pin_at<unsigned long long>(c, 1024);
        if (segmented)
            HIPCK(hipMemcpyAsync(pin_at<unsigned long long>(c, 1024), c->segCtr.p, DEME_KEY_SEGS * DEME_KEY_SEG_STRIDE * 8, hipMemcpyDeviceToHost, st));
        HIPCK(sync_readback(c, st));
        hc = *pin_at<DetectCounters>(c, 64);
        if (!statusSeen) {
            statusSeen = true;
            if (int rc = status_to_error(c, hc.status & ~DEME_ST_INCIDENCE))
                return rc;
        }
        unsigned long long segMax = 0;
        if (segmented) {
            hc.nContactsRaw = 0;
            for (uint32_t g = 0; g < DEME_KEY_SEGS; g++) {
                hc.nContactsRaw += hseg[g * DEME_KEY_SEG_STRIDE];
                segMax = std::max(segMax, hseg[g * DEME_KEY_SEG_STRIDE]);
            }
        }
        if (hc.nContactsRaw > c->cntCap || segMax > ar.segCap) {  // arena (or one segment of it) too small: grow and redo the emitting kernels
            if (async)
                HIPCK(hipStreamSynchronize(c->stream));
            const size_t want = std::max<size_t>(hc.nContactsRaw, (size_t)segMax * DEME_KEY_SEGS);
            if (int rc = grow_contact_arena(c, want + want / 4 + 1024))
                return rc;
            continue;
        }
        c->nActiveBins = hc.nActiveBins;
        c->maxInBin = hc.maxInBin ? hc.maxInBin : (P ? 1u : 0u);
        if (c->maxInBin > c->dp.errOutBinSphNum)
            return fail(c, DEME_ERR_BIN_TOO_FULL,
                        "a bin contains %u sphere components, exceeding the allowance %u (SetMaxSphereInBin)", c->maxInBin,
                        c->dp.errOutBinSphNum);
        uint64_t nC = hc.nContactsRaw;
        const size_t nPersist = c->hPersist.size();
        if (nPersist) {  // marked contacts join the list whether or not the sweep found them (DEMCubContactDetection.cu:605-802)
            if (nC + nPersist > c->cntCap) {
                if (async)
                HIPCK(hipStreamSynchronize(c->stream));
            if (int rc = grow_contact_arena(c, (size_t)nC + nPersist + nC / 4 + 1024))
                    return rc;
                continue;
            }
            nC += nPersist;
        }
        const int next = c->keysCur ^ 1;
        uint64_t* rawKeys = c->keysRaw.as<uint64_t>();
        if (segmented) {  // (the gaps were closed above; the next list's buffer is free until the rank sort fills it)
            c->lastSegMax = segMax;
            if (grid_for(segMax) > compactBlocks)
                hipLaunchKernelGGL(k_compact_keys, dim3((unsigned)grid_for(segMax) - compactBlocks, DEME_KEY_SEGS), dim3(256), 0, st,
                                   c->keysRaw.as<uint64_t>(), ar.segCap, c->segCtr.as<unsigned long long>(), compactBlocks,
                                   c->keysSorted[next].as<uint64_t>());
            if (hc.nContactsRaw)
                rawKeys = c->keysSorted[next].as<uint64_t>();
        }
        if (nPersist)
            HIPCK(hipMemcpyAsync(rawKeys + hc.nContactsRaw, c->persistKeys.p, nPersist * 8, hipMemcpyDeviceToDevice, st));
        if (nC) {
            // key = A << 33 | class << 31 | B.  One radix sort over the occupied upper bits [31, 33 + bits(A)) groups the keys by
            // (sphere A, class) -- 3 passes at 3e6 spheres, where a full 64-bit order took 6 to 8 -- and k_segment_rank_sort puts each
            // group's few partners in order (DEMCubContactDetection.cu:811-1110 runs five full sorts for the same list order)
            auto bits_of = [](uint64_t n) {
                unsigned b = 1;
                while (b < 31 && (1ull << b) < n)
                    b++;
                return b;
            };
            const unsigned bitsA = bits_of(c->dp.nSpheres);
            if (int rc = ensure(c, c->keysMid, (size_t)c->cntCap * 8))  // (its own scratch: an asynchronous detection runs beside force passes)
                return rc;
            uint64_t* mid = c->keysMid.as<uint64_t>();
            size_t needHi = 0;
            HIPCK(rocprim::radix_sort_keys<DemeRadixCfg<8>>(nullptr, needHi, rawKeys, mid, (size_t)nC, 31, 33 + bitsA, st));
            if (int rc = ensure(c, c->sortTmp, needHi))
                return rc;
            needHi = c->sortTmp.bytes;
            HIPCK(rocprim::radix_sort_keys<DemeRadixCfg<8>>(c->sortTmp.p, needHi, rawKeys, mid, (size_t)nC, 31, 33 + bitsA, st));
            hipLaunchKernelGGL(k_segment_rank_sort, dim3(grid_for(nC)), dim3(256), 0, st, (uint32_t)nC, mid,
                               c->keysSorted[next].as<uint64_t>());
            if (nPersist) {  // a marked contact the sweep found as well appears once (markDuplicateContacts)
                unsigned long long* cnt = &c->ctr.as<DetectCounters>()->nContactsRaw;
                size_t need2 = 0;
                HIPCK(rocprim::unique(nullptr, need2, c->keysSorted[next].as<uint64_t>(), c->keysRaw.as<uint64_t>(), cnt, (size_t)nC,
                                      rocprim::equal_to<uint64_t>(), st));
                if (int rc = ensure(c, c->scanTmp, need2))
                    return rc;
                need2 = c->scanTmp.bytes;
                HIPCK(rocprim::unique(c->scanTmp.p, need2, c->keysSorted[next].as<uint64_t>(), c->keysRaw.as<uint64_t>(), cnt,
                                      (size_t)nC, rocprim::equal_to<uint64_t>(), st));
                unsigned long long nU = 0;
                HIPCK(hipMemcpyAsync(&nU, cnt, 8, hipMemcpyDeviceToHost, st));
                HIPCK(hipStreamSynchronize(st));
                nC = nU;
                HIPCK(hipMemcpyAsync(c->keysSorted[next].p, c->keysRaw.p, (size_t)nC * 8, hipMemcpyDeviceToDevice, st));
            }
        }
        *nCout = nC;
        return DEME_OK;
    }
    return fail(c, DEME_ERR_OVERFLOW, "contact arena kept overflowing");
}

// The owner records the list builders read (tile origins, ghost and family flags): the live ones -- or, while an asynchronous
// detection builds its list beside steps that are integrating them, the snapshot part 1 was made from
static inline const OwnerRec* list_owners(deme_ctx* c) {
    return c->listOwnersSnap ? c->ownersSnap.as<OwnerRec>() : c->owners.as<OwnerRec>();
}

// The B-sorted contact list of the round-2 kernels (k_forces_fast / k_calc_forces + the integrator's gather): owner-B sort, run
// starts, heavy / fixed flags, the deferral flags of the halo overlap.  Enqueued on the main stream; the caller reads rangeCtr.
int build_legacy_lists(deme_ctx* c) {
    const uint64_t nC = c->nListed;
    if (nC) {
        unsigned obits = 1;
        while (obits < 32 && (1ull << obits) < (uint64_t)c->nOwners)
            obits++;
        size_t need = 0;
        HIPCK(rocprim::radix_sort_pairs(nullptr, need, c->ownerB[0].as<uint32_t>(), c->ownerB[1].as<uint32_t>(),
                                        c->bIdx[0].as<uint32_t>(), c->bIdx[1].as<uint32_t>(), (size_t)nC, 0, obits, c->stream));
        if (int rc = ensure(c, c->sortTmp, need))
            return rc;
        need = c->sortTmp.bytes;
        HIPCK(rocprim::radix_sort_pairs(c->sortTmp.p, need, c->ownerB[0].as<uint32_t>(), c->ownerB[1].as<uint32_t>(),
                                        c->bIdx[0].as<uint32_t>(), c->bIdx[1].as<uint32_t>(), (size_t)nC, 0, obits, c->stream));
    }
    if (c->hasGhosts) {  // (the arena may have been sized before the scene's family flags were known)
        if (ensure(c, c->cDefer, c->cntCap) || ensure(c, c->blockMode, (c->cntCap / DEME_FORCE_BLOCK + 2) * 4))
            return c->lastStatus;
        HIPCK(hipMemsetAsync(c->blockMode.p, 0, c->blockMode.bytes, c->stream));
    }
    if (nC)
        hipLaunchKernelGGL(k_run_starts, dim3(grid_for(nC)), dim3(256), 0, c->stream, (uint32_t)nC, c->ownerB[1].as<uint32_t>(),
                           c->nOwners, c->bStart.as<uint32_t>());
    else
        HIPCK(hipMemsetAsync(c->bStart.p, 0, ((size_t)c->nOwners + 1) * 4, c->stream));
    // (the counters may hold the tile builders' count of the same owners)
    HIPCK(hipMemsetAsync(&c->rangeCtr.as<RangeCounters>()->nHeavy, 0, 2 * sizeof(unsigned int), c->stream));
    hipLaunchKernelGGL(k_owner_ranges, dim3(grid_for((size_t)c->nOwners + 1)), dim3(256), 0, c->stream, c->dp, (uint32_t)nC,
                       c->ownerA.as<uint32_t>(), c->ownerB[1].as<uint32_t>(), list_owners(c), c->aStart.as<uint32_t>(),
                       c->bStart.as<uint32_t>(), c->heavy.as<uint8_t>(), c->fixedFlag.as<uint8_t>(), c->heavyList.as<uint32_t>(),
                       (uint32_t)(c->heavyList.bytes / 4), c->rangeCtr.as<RangeCounters>(), c->info.as<uint4>(),
                       c->hasGhosts ? c->cDefer.as<uint8_t>() : (uint8_t*)nullptr, c->blockMode.as<uint32_t>());
    c->legacyLists = true;
    return DEME_OK;
}

void resolve_heavy_counts(deme_ctx* c);
// a list built with tile structures that the round-2 kernels evaluate after all (contact recording switched on, the arithmetic
// mode changed, DEME_TILE=0 set between detections ...): its B-sorted form is built now
int ensure_legacy_lists(deme_ctx* c) {
    if (c->legacyLists || !c->haveList)
        return DEME_OK;
    resolve_heavy_counts(c);
    if (int rc = build_legacy_lists(c))
        return rc;
    RangeCounters hr{};
    HIPCK(hipMemcpyAsync(&hr, c->rangeCtr.p, sizeof(hr), hipMemcpyDeviceToHost, c->stream));
    HIPCK(hipStreamSynchronize(c->stream));
    if (hr.nHeavy > c->heavyList.bytes / 4)
        return fail(c, DEME_ERR_OVERFLOW, "%u owners exceed the heavy-owner list", hr.nHeavy);
    c->nHeavy = hr.nHeavy, c->nHeavyFree = hr.nHeavyFree;
    return DEME_OK;
}

int detect_part2(deme_ctx* c, uint64_t nC) {
    const int next = c->keysCur ^ 1;
    {
        if (nC) {
            const uint64_t nPrev = c->haveList ? c->nContacts : 0;
            hipLaunchKernelGGL(k_history, dim3(grid_for(nC)), dim3(256), 0, c->stream, (uint32_t)nC,
                               c->keysSorted[next].as<uint64_t>(), (uint32_t)nPrev,
                               c->keysSorted[c->keysCur].as<uint64_t>(), c->mapping.as<uint32_t>());
        }
        // per-owner gather lists for the atomics-free accumulation
        HIPCK(hipMemsetAsync(c->rangeCtr.p, 0, sizeof(RangeCounters), c->stream));
        const bool tileEligible = c->tileEnable && nC && c->arith == DEME_ARITH_FAST &&
                                  (c->hp.forceModel != DEME_FORCE_CUSTOM ||
                                   (c->customTileFn[0] && nC >= (uint64_t)c->tileMinContactsCustom * ((c->nOwners + DEME_TILE_NB - 1) / DEME_TILE_NB))) &&
                                  c->hShared.empty() && c->nMat <= 16 && c->nAnal <= 65535 && c->nComp <= 65535 &&
                                  tile_table_bytes(c->nComp, c->nMat, c->nAnal, c->nMassProps, c->dp.familyTrivial) <= DEME_TILE_TABLE_MAX &&
                                  c->nComp + c->nMat * c->nMat * 2u + c->nAnal * 4u <= DEME_TILE_T && c->nMassProps <= DEME_TILE_T;
        const uint32_t nTiles = (c->nOwners + DEME_TILE_NB - 1) / DEME_TILE_NB;
        if (tileEligible)
            HIPCK(hipMemsetAsync(c->tileRem.p, 0, ((size_t)nTiles + 1) * 4, c->stream));
        if (nC) {
            hipLaunchKernelGGL(k_contact_owners, dim3(grid_for(nC)), dim3(256), 0, c->stream, c->dp, (uint32_t)nC,
                               c->keysSorted[next].as<uint64_t>(), c->spheres.as<SphereRec>(), c->ownerA.as<uint32_t>(),
                               c->ownerB[0].as<uint32_t>(), c->bIdx[0].as<uint32_t>(), c->info.as<uint4>(),
                               c->nTri ? c->smFlag.as<uint8_t>() : (uint8_t*)nullptr,
                               tileEligible ? c->tileRem.as<uint32_t>() : (uint32_t*)nullptr, (uint32_t)DEME_TILE_NB);
            if (c->nTri) {  // work list of the mesh-variant force kernel; its length lands in RangeCounters::nSM
                unsigned int* cnt = &c->rangeCtr.as<RangeCounters>()->nSM;
                size_t need2 = 0;
                HIPCK(rocprim::select(nullptr, need2, rocprim::counting_iterator<uint32_t>(0), c->smFlag.as<uint8_t>(),
                                      c->smList.as<uint32_t>(), cnt, (size_t)nC, c->stream));
                if (int rc = ensure(c, c->scanTmp, need2))
                    return rc;
                need2 = c->scanTmp.bytes;
                HIPCK(rocprim::select(c->scanTmp.p, need2, rocprim::counting_iterator<uint32_t>(0), c->smFlag.as<uint8_t>(),
                                      c->smList.as<uint32_t>(), cnt, (size_t)nC, c->stream));
            }
            hipLaunchKernelGGL(k_run_starts, dim3(grid_for(nC)), dim3(256), 0, c->stream, (uint32_t)nC, c->ownerA.as<uint32_t>(),
                               c->nOwners, c->aStart.as<uint32_t>());
        } else {
            HIPCK(hipMemsetAsync(c->aStart.p, 0, ((size_t)c->nOwners + 1) * 4, c->stream));
        }
        c->legacyLists = false;
        c->nListed = nC, c->listKeys = next;
        bool tiled = false;
        RangeCounters hr{};
        if (tileEligible) {  // owner tiles (deme_tile.h): the builders of the list structures k_tile_forces and the integrator read
            size_t need = 0;
            HIPCK(rocprim::exclusive_scan(nullptr, need, c->tileRem.as<uint32_t>(), c->tileBase.as<uint32_t>(), 0u, (size_t)nTiles + 1,
                                          rocprim::plus<uint32_t>(), c->stream));
            if (int rc = ensure(c, c->scanTmp, need))
                return rc;
            need = c->scanTmp.bytes;
            HIPCK(rocprim::exclusive_scan(c->scanTmp.p, need, c->tileRem.as<uint32_t>(), c->tileBase.as<uint32_t>(), 0u, (size_t)nTiles + 1,
                                          rocprim::plus<uint32_t>(), c->stream));
            hipLaunchKernelGGL(k_tile_build, dim3(nTiles), dim3(256), 0, c->stream, c->dp, c->nOwners, c->info.as<uint4>(),
                               c->ownerB[0].as<uint32_t>(), c->aStart.as<uint32_t>(), list_owners(c), c->tileBase.as<uint32_t>(), c->tInfo.as<uint2>(),
                               c->hList.as<uint32_t>(), c->hCount.as<uint32_t>(),
                               c->hasGhosts ? c->tileMode.as<uint32_t>() : (uint32_t*)nullptr, c->lOff.as<uint16_t>(),
                               c->lPos.as<uint16_t>(), c->lCount.as<uint32_t>(), c->rankC.as<uint32_t>(), c->remKey[0].as<uint32_t>(),
                               c->remVal.as<uint32_t>(), c->tileOrg.as<int64_t>(), c->rangeCtr.as<RangeCounters>(), nTiles,
                               c->tileBig.as<uint32_t>(), c->bigList.as<uint32_t>(), c->recContact.as<uint32_t>());
            hipLaunchKernelGGL(k_tile_stats, dim3((nTiles + 1023u) / 1024u), dim3(256), 0, c->stream, nTiles, c->hCount.as<uint32_t>(),
                               c->lCount.as<uint32_t>(), c->rangeCtr.as<RangeCounters>());
            uint32_t nR = 0;  // crossing contacts = records = entries of the sort below: the one size the host has to know
            HIPCK(hipMemcpyAsync(pin_at<uint32_t>(c, 8), c->tileBase.as<uint32_t>() + nTiles, 4, hipMemcpyDeviceToHost, c->stream));
            HIPCK(hipMemcpyAsync(pin_at<RangeCounters>(c, 128), c->rangeCtr.p, sizeof(hr), hipMemcpyDeviceToHost, c->stream));
            HIPCK(sync_readback(c, c->stream));
            nR = *pin_at<uint32_t>(c, 8), hr = *pin_at<RangeCounters>(c, 128);
            nR += hr.nExtra;  // the records of the tiles that do not fit (k_tile_forces_big): every contact of such a tile has one
            c->nBigTiles = hr.nBig;
            {
                unsigned obits = 1;
                while (obits < 32 && (1ull << obits) < (uint64_t)c->nOwners)
                    obits++;
                if (nR) {
                    size_t needS = 0;
                    HIPCK(rocprim::radix_sort_pairs<DemeRadixCfg<10>>(nullptr, needS, c->remKey[0].as<uint32_t>(), c->remKey[1].as<uint32_t>(),
                                                    c->remVal.as<uint32_t>(), c->rIdx.as<uint32_t>(), (size_t)nR, 0, obits, c->stream));
                    if (int rc = ensure(c, c->sortTmp, needS))
                        return rc;
                    needS = c->sortTmp.bytes;
                    HIPCK(rocprim::radix_sort_pairs<DemeRadixCfg<10>>(c->sortTmp.p, needS, c->remKey[0].as<uint32_t>(), c->remKey[1].as<uint32_t>(),
                                                    c->remVal.as<uint32_t>(), c->rIdx.as<uint32_t>(), (size_t)nR, 0, obits, c->stream));
                    hipLaunchKernelGGL(k_run_starts, dim3(grid_for(nR)), dim3(256), 0, c->stream, nR, c->remKey[1].as<uint32_t>(),
                                       c->nOwners, c->rStart.as<uint32_t>());
                } else {
                    HIPCK(hipMemsetAsync(c->rStart.p, 0, ((size_t)c->nOwners + 1) * 4, c->stream));
                }
                hipLaunchKernelGGL(k_owner_ranges_tile, dim3(grid_for(c->nOwners)), dim3(256), 0, c->stream, c->dp, list_owners(c),
                                   c->aStart.as<uint32_t>(), c->lOff.as<uint16_t>(), c->rStart.as<uint32_t>(), c->heavy.as<uint8_t>(),
                                   c->fixedFlag.as<uint8_t>(), c->heavyList.as<uint32_t>(), (uint32_t)(c->heavyList.bytes / 4),
                                   c->rangeCtr.as<RangeCounters>());
                tiled = true;
                // closed tiles (deme_tile_step.h): the contacts that hold a tile's owners as B from other tiles, in the tile's frame
                static const int fusedEnv = getenv("DEME_FUSED") ? atoi(getenv("DEME_FUSED")) : -1;  // (1 / 0 override the context's switch)
                c->fusedList = false;
                const int fusedMode = fusedEnv < 0 ? c->fusedEnable : fusedEnv;  // 0 off, 1 on, 2 by the size of the bed
                if ((fusedMode == 2 ? c->nOwners <= DEME_FUSED_AUTO_MAX_OWNERS : fusedMode != 0) && hr.nBig == 0 && c->hp.forceModel != DEME_FORCE_CUSTOM && !c->record && c->nTri == 0 &&
                    !c->hasGhosts && !c->prescFn && !c->rulesFn && c->hShared.empty() && c->asyncLead == 0 && !c->listOwnersSnap &&
                    !c->ad.autoBinSize && !c->ad.autoUpdateFreq) {
                    hipLaunchKernelGGL(k_in_count, dim3(grid_for((size_t)c->nOwners + 1)), dim3(256), 0, c->stream, c->dp, list_owners(c),
                                       c->rStart.as<uint32_t>(), c->inCnt.as<uint32_t>());
                    size_t needI = 0;
                    HIPCK(rocprim::exclusive_scan(nullptr, needI, c->inCnt.as<uint32_t>(), c->inStart.as<uint32_t>(), 0u, (size_t)c->nOwners + 1,
                                                  rocprim::plus<uint32_t>(), c->stream));
                    if (int rc = ensure(c, c->scanTmp, needI))
                        return rc;
                    needI = c->scanTmp.bytes;
                    HIPCK(rocprim::exclusive_scan(c->scanTmp.p, needI, c->inCnt.as<uint32_t>(), c->inStart.as<uint32_t>(), 0u, (size_t)c->nOwners + 1,
                                                  rocprim::plus<uint32_t>(), c->stream));
                    hipLaunchKernelGGL(k_tile_incoming, dim3(nTiles), dim3(256), 0, c->stream, c->dp, c->nOwners, c->info.as<uint4>(),
                                       c->rStart.as<uint32_t>(), c->rIdx.as<uint32_t>(), c->recContact.as<uint32_t>(), c->inStart.as<uint32_t>(),
                                       c->hList.as<uint32_t>(), c->hCount.as<uint32_t>(), c->hCountIn.as<uint32_t>(), c->tInfoIn.as<uint2>(),
                                       c->inContact.as<uint32_t>(), c->rangeCtr.as<RangeCounters>());
                    c->fusedList = true;  // (whether every closed tile fits comes back with the counters below: fused_ready)
                }
                c->fusedChecked = false;
            }
        }
        c->hrPending = c->heavyOverflow = false;
        if (tiled) {  // (hr holds the tile extremes already; nSA / nSM are not used by the tile path)
            if (!c->hrPinned) {
                HIPCK(hipHostMalloc((void**)&c->hrPinned, sizeof(RangeCounters), hipHostMallocDefault));
                HIPCK(hipEventCreateWithFlags(&c->hrEvent, hipEventDisableTiming));
            }
            HIPCK(hipMemcpyAsync(c->hrPinned, c->rangeCtr.p, sizeof(RangeCounters), hipMemcpyDeviceToHost, c->stream));
            HIPCK(hipEventRecord(c->hrEvent, c->stream));
            c->hrPending = true;
            c->nHeavy = c->nHeavyFree = 0;
        } else {
            if (int rc = build_legacy_lists(c))
                return rc;
            HIPCK(hipMemcpyAsync(&hr, c->rangeCtr.p, sizeof(hr), hipMemcpyDeviceToHost, c->stream));
            HIPCK(hipStreamSynchronize(c->stream));
            if (hr.nHeavy > c->heavyList.bytes / 4)
                return fail(c, DEME_ERR_OVERFLOW, "%u owners exceed the heavy-owner list", hr.nHeavy);
            c->nHeavy = hr.nHeavy;
            c->nHeavyFree = hr.nHeavyFree;
        }
        c->tileActive = tiled;
        c->fusedPrevValid = false;  // (a replay of the last fused step would read the list it was taken with)
        c->tileMaxHalo = hr.tileMaxHalo, c->tileMaxList = hr.tileMaxList;
        if (tiled && c->orderEligible) {  // has the bed drifted away from the order it was given?  (order_renew at the next detection)
            const uint32_t fit = std::max(1u, nTiles - std::min(nTiles, hr.nBig));
            const uint32_t mean16 = (uint32_t)(16ull * hr.tileHaloSum / fit);
            if (!c->orderBaseHalo) {
                c->orderBaseHalo = std::max(mean16, 16u);
                c->orderDetAt = c->nDetections;
            } else if (c->nDetections - c->orderDetAt >= 20 && nTiles > 4 &&
                       (hr.nBig > std::max(4u, nTiles / 64u) || mean16 > c->orderBaseHalo + c->orderBaseHalo / 2)) {
                c->orderRenewDue = true;
            }
        }
        c->nSA = hr.nSA;
        c->nSM = hr.nSM;
        c->conValid = false;
        c->nPrev = c->haveList ? c->nContacts : 0;
        c->nContacts = nC;
        c->keysCur = next;
        c->listSerial++;
        c->haveList = true;
        c->seeded = false;
        c->mapFresh = true;
        c->nDetections++;
        return DEME_OK;
    }
}


int do_detect(deme_ctx* c) {
    // a history map nobody has applied yet (two detections in a row): apply it now, or the next map would be taken from a
    // list whose wildcards are still stored against the list before it
    if (c->mapFresh)
        if (int rc = do_migrate(c))
            return rc;
    ScopedTimer tm(c, "detect", true);  // once per K steps: always timed
    uint64_t nC = 0;
    if (int rc = detect_part1(c, c->stream, c->owners.as<OwnerRec>(), false, &nC))
        return rc;
    return detect_part2(c, nC);
}

int do_migrate(deme_ctx* c) {
    const uint32_t nW = c->hp.nContactWildcards;
    if (!c->mapFresh)
        return DEME_OK;
    c->mapFresh = false;
    if (nW == 0) {
        c->nWcStored = c->nContacts;
        return DEME_OK;
    }
    const int next = c->wcCur ^ 1;
    if (c->nContacts)
        hipLaunchKernelGGL(k_migrate, dim3(grid_for(c->nContacts)), dim3(256), 0, c->stream, (uint32_t)c->nContacts, nW,
                           c->mapping.as<uint32_t>(), (uint32_t)c->nWcStored, c->wc[c->wcCur].as<float>(),
                           c->wc[next].as<float>());
    c->wcCur = next;
    c->nWcStored = c->nContacts;
    return DEME_OK;
}

GatherArgs gather_args(deme_ctx* c) {
    GatherArgs g{};
    g.aStart = c->aStart.as<uint32_t>(), g.bStart = c->bStart.as<uint32_t>(), g.bIdx = c->bIdx[1].as<uint32_t>();
    g.heavy = c->heavy.as<uint8_t>();
    g.conA4 = c->conA4.as<float4>(), g.conA2 = c->conA2.as<float2>();
    g.conB4 = c->conB4.as<float4>(), g.conB2 = c->conB2.as<float2>();
    g.aSum = c->aSum.as<float4>();
    g.nextAcc = c->nextAccPending ? c->nextAcc.as<AccRec>() : nullptr;
    if (c->pairsOnce && c->revAcc)
        g.revSlot = c->revSlot.as<uint32_t>(), g.revAcc = (const float4*)c->revAcc;
    g.world = c->arith == DEME_ARITH_FAST ? 1u : 0u;
    if (c->conTile) {  // the tile kernel's sums and its records of tile-crossing contacts
        g.bStart = c->rStart.as<uint32_t>(), g.bIdx = c->rIdx.as<uint32_t>();
        g.rec32 = c->rec32.as<float4>();
        g.tile = 1u;
    }
    return g;
}

// heavy owners: skipFixed=true in the stepping loop (a fixed owner's a/alpha are only needed by queries)
void resolve_heavy_counts(deme_ctx* c) {
    if (!c->hrPending)
        return;
    c->hrPending = false;
    if (const hipError_t e = hipEventSynchronize(c->hrEvent); e != hipSuccess) {
        fail(c, DEME_ERR_HIP, "waiting for the heavy-owner counts: %s", hipGetErrorString(e));
        c->heavyOverflow = true;  // (the stepping loop returns the error: launch_integrate)
        return;
    }
    const size_t cap = c->heavyList.bytes / 4;
    c->nHeavy = c->hrPinned->nHeavy, c->nHeavyFree = c->hrPinned->nHeavyFree;
    if (c->fusedList) {  // the halos were extended by the incoming contacts' owners (k_tile_incoming): the kernels' LDS follows
        c->tileMaxHaloIn = c->hrPinned->tileMaxHaloIn;
        if (c->hrPinned->nBigIn)  // a closed tile does not fit: this list keeps force pass + integrator
            c->fusedList = false;
        c->fusedChecked = true;
    }
    if (c->nHeavy > cap) {
        fail(c, DEME_ERR_OVERFLOW, "%u owners exceed the heavy-owner list", c->nHeavy);
        c->nHeavy = (uint32_t)cap;
        c->heavyOverflow = true;  // (the stepping loop returns the error)
    }
}

void launch_reduce_heavy(deme_ctx* c, bool skipFixed) {
    resolve_heavy_counts(c);
    if (c->nHeavy == 0 || (skipFixed && c->nHeavyFree == 0))
        return;
    hipLaunchKernelGGL(k_reduce_heavy, dim3(std::min<uint32_t>(c->nHeavy, 1024)), dim3(256), 0, c->stream, c->dp, gather_args(c),
                       c->owners.as<OwnerRec>(), c->heavyList.as<uint32_t>(), &c->rangeCtr.as<RangeCounters>()->nHeavy,
                       skipFixed ? c->fixedFlag.as<uint8_t>() : (const uint8_t*)nullptr, c->acc.as<AccRec>());
}

// the stride of the staged owner records of a tile launch (TileArgs::rs16): padded by 16 bytes when that costs no workgroup per CU
// (160 KB of LDS per CU, allocated in 512-byte blocks; the kernels' registers allow four workgroups)
static uint32_t tile_record_stride(uint32_t hCap, uint32_t lCap, uint32_t tabBytes, int model) {
    const uint32_t base = tile_rec16(model);
    static const int padEnv = getenv("DEME_TILE_PAD_RECORDS") ? atoi(getenv("DEME_TILE_PAD_RECORDS")) : 1;
    if (model == 2 || !padEnv)  // (80-byte records start on 16 banks already)
        return base;
    auto groups = [&](uint32_t rs) { return std::min<uint32_t>(4u, 163840u / ((tile_lds_bytes(hCap, lCap, tabBytes, rs) + 511u) & ~511u)); };
    return groups(base + 1u) == groups(base) ? base + 1u : base;
}
// pass: -1 everything in one launch; 0 / 1 the two halves of a split step (contacts that read no ghost owner / the rest)
// `fs`: the stream of a tile-form launch when it is not the context's (the ghost-dependent pass of a split step runs on the halo
// stream, beside the tail of the interior pass)
int launch_forces(deme_ctx* c, int pass = -1, hipStream_t fs = nullptr) {
    c->fusedPrevValid = false;
    if (c->nContacts == 0) {
        c->conValid = true;
        c->conTile = false;
        return DEME_OK;
    }
    if (c->hp.forceModel == DEME_FORCE_CUSTOM && !c->customFn[0])
        return fail(c, DEME_ERR_INVALID, "custom force model selected but none compiled (deme_compile_force_model)");
    ForceArgs a{};
    a.owners = c->owners.as<OwnerRec>();
    a.spheres = c->spheres.as<SphereRec>();
    a.keys = c->keysSorted[c->keysCur].as<uint64_t>();
    a.info = c->info.as<uint4>();
    a.wc = c->wc[c->wcCur].as<float>();
    a.conA4 = c->conA4.as<float4>(), a.conA2 = c->conA2.as<float2>();
    a.conB4 = c->conB4.as<float4>(), a.conB2 = c->conB2.as<float2>();
    a.aSum = c->aSum.as<float4>();
    a.aStart = c->aStart.as<uint32_t>();
    a.smList = c->smList.as<uint32_t>();
    a.nSM = c->nSM;
    for (int k = 0; k < 8; k++) {
        a.ownerWc[k] = c->userWc[0][k].as<float>();
        a.geoWcSph[k] = c->userWc[1][k].as<float>();
        a.geoWcTri[k] = c->userWc[2][k].as<float>();
        a.geoWcAnal[k] = c->userWc[3][k].as<float>();
    }
    if (pass >= 0 && c->hasGhosts && c->cDefer.p) {
        a.cDefer = c->cDefer.as<uint8_t>();
        a.blockMode = c->blockMode.as<uint32_t>();
        a.pass = (uint32_t)pass;
    }
    a.nContacts = (uint32_t)c->nContacts;
    a.timeElapsed = (float)c->timeElapsed;
    a.xcdGroup = c->xcdGroup;
    const bool fastMode = c->arith == DEME_ARITH_FAST;
    a.world = fastMode ? 1u : 0u;
    // the fast kernel covers the built-in models' hot classes; contact recording (body-frame contact points) and user
    // fragments (the reference's body-frame vocabulary) run the general kernel, with world-frame contributions in fast mode
    // contact recording: the tile pass writes the records itself (REC instances) for the built-in models; user models record
    // through the general kernel
    const bool fastKernel = fastMode && c->hp.forceModel != DEME_FORCE_CUSTOM;
    const bool customTile = fastMode && !c->record && c->hp.forceModel == DEME_FORCE_CUSTOM && c->customTileFn[0];
    if ((fastKernel || customTile) && c->tileActive && c->tileEnable) {  // owner tiles: deme_tile.h
        if (c->fusedList && !c->fusedChecked)
            resolve_heavy_counts(c);
        TileArgs ta{};
        ta.owners = a.owners;
        ta.tInfo = c->tInfo.as<uint2>();
        ta.aStart = a.aStart;
        ta.hList = c->hList.as<uint32_t>(), ta.hCount = c->hCount.as<uint32_t>(), ta.org = c->tileOrg.as<int64_t>();
        ta.lOff = c->lOff.as<uint16_t>(), ta.lPos = c->lPos.as<uint16_t>(), ta.lCount = c->lCount.as<uint32_t>();
        ta.wc = a.wc;
        ta.tSum = a.aSum;
        ta.rec32 = c->rec32.as<float4>(), ta.rankC = c->rankC.as<uint32_t>();
        ta.nOwners = c->nOwners;
        ta.nTiles = (c->nOwners + DEME_TILE_NB - 1) / DEME_TILE_NB;
        ta.xcdGroup = c->xcdGroup;
        ta.tileBig = c->tileBig.as<uint32_t>(), ta.bigList = c->bigList.as<uint32_t>(), ta.info = a.info;
        ta.keys = a.keys, ta.timeElapsed = a.timeElapsed;
        if (c->record) {
            for (int k = 0; k < 4; k++)
                ta.rec[k] = c->rec[k].as<float>();
            a.recForce = c->rec[0].as<float>(), a.recTorque = c->rec[1].as<float>(), a.recCPA = c->rec[2].as<float>(),
            a.recCPB = c->rec[3].as<float>();  // (the mesh variant of the general kernel records its own contacts)
        }
        for (int k = 0; k < 8; k++)
            ta.ownerWc[k] = a.ownerWc[k], ta.geoWcSph[k] = a.geoWcSph[k], ta.geoWcAnal[k] = a.geoWcAnal[k];
        if (pass >= 0 && c->hasGhosts) {
            ta.tileMode = c->tileMode.as<uint32_t>();
            ta.pass = (uint32_t)pass;
        }
        unsigned nBlk = ta.nTiles;
        if (ta.xcdGroup)
            nBlk = (nBlk + 8u * ta.xcdGroup - 1u) / (8u * ta.xcdGroup) * (8u * ta.xcdGroup);
        // LDS sized from this list's largest tile (rounded up so that a launch configuration serves many detections)
        ta.hCap = std::min<uint32_t>(DEME_TILE_HMAX, (c->tileMaxHalo + 15u) & ~15u);
        ta.lCap = std::min<uint32_t>(DEME_TILE_LMAX, (c->tileMaxList + 15u) & ~15u);
        ta.nComp = c->nComp, ta.nAnal = c->nAnal, ta.nMass = c->nMassProps;
        static const uint32_t ldsPad = getenv("DEME_TILE_LDS_PAD") ? (uint32_t)atoi(getenv("DEME_TILE_LDS_PAD")) : 0u;  // occupancy experiments
        const uint32_t tabBytes = tile_table_bytes(c->nComp, c->nMat, c->nAnal, c->nMassProps, c->dp.familyTrivial);
        ta.rs16 = tile_record_stride(ta.hCap, ta.lCap, tabBytes, customTile ? 2 : 0);
        static const int swzEnv = getenv("DEME_TILE_SWIZZLE") ? atoi(getenv("DEME_TILE_SWIZZLE")) : 1;  // (2: rotate instead of padding everywhere)
        if (swzEnv == 2 && !customTile)
            ta.rs16 = tile_rec16(0);
        ta.swz = (swzEnv && !customTile && ta.rs16 == tile_rec16(0)) ? 1u : 0u;
        const uint32_t ldsBytes = tile_lds_bytes(ta.hCap, ta.lCap, tabBytes, ta.rs16) + ldsPad;
        hipStream_t st = fs ? fs : c->stream;
        ScopedTimer tm(c, "calc_forces", false, st);
        const bool mesh = c->nTri > 0 && c->nSM > 0;
        if (mesh) {
            ta.conA4 = a.conA4, ta.conA2 = a.conA2, ta.conB4 = a.conB4, ta.conB2 = a.conB2;
            if (pass != 1) {  // sphere-triangle contacts: the mesh variant of the general kernel, before the tiles that read its records
                              // (they read no ghost owner -- meshes are replicated, not ghosted --: all of them go with pass 0)
                const dim3 gm(grid_for(std::max<uint32_t>(c->nSM, 1u), DEME_FORCE_BLOCK)), bm(DEME_FORCE_BLOCK);
                if (customTile) {
                    void* args0[] = {&c->dp, &a};
                    HIPCK(hipModuleLaunchKernel(c->customFn[1], gm.x, 1, 1, DEME_FORCE_BLOCK, 1, 1, 0, c->stream, args0, nullptr));
                } else if (c->hp.forceModel == DEME_FORCE_HERTZIAN)
                    hipLaunchKernelGGL((k_calc_forces<0, 1>), gm, bm, 0, c->stream, c->dp, a);
                else
                    hipLaunchKernelGGL((k_calc_forces<1, 1>), gm, bm, 0, c->stream, c->dp, a);
            }
        }
        const unsigned nBig = c->nBigTiles;  // tiles that do not fit LDS: one workgroup each, the same outputs (deme_tile.h)
#if DEME_TILE_STAMPS
        // measurement builds: DEME_TILE_STAMPS_FILE=path[:launch] -- the phase stamps of every tile of that launch (default the 100th), 16 words per tile
        static const char* stampEnv = getenv("DEME_TILE_STAMPS_FILE");
        static unsigned long long* stampBuf = nullptr;
        static int stampLaunch = 0, stampAt = 100;
        static std::string stampPath;
        if (stampEnv && !stampBuf && !customTile) {
            stampPath = stampEnv;
            const size_t colon = stampPath.rfind(':');
            if (colon != std::string::npos)
                stampAt = atoi(stampPath.c_str() + colon + 1), stampPath.resize(colon);
            HIPCK(hipMalloc(&stampBuf, (size_t)ta.nTiles * 16 * 8));
            HIPCK(hipMemset(stampBuf, 0, (size_t)ta.nTiles * 16 * 8));
        }
        const bool stampNow = stampBuf && ++stampLaunch == stampAt;
        ta.stamps = stampNow ? stampBuf : nullptr;
        struct StampDump {
            bool on; hipStream_t st; unsigned long long* buf; size_t n; const std::string& path;
            ~StampDump() {
                if (!on) return;
                hipStreamSynchronize(st);
                std::vector<unsigned long long> h(n);
                hipMemcpy(h.data(), buf, n * 8, hipMemcpyDeviceToHost);
                if (FILE* f = fopen(path.c_str(), "wb")) { fwrite(h.data(), 8, n, f); fclose(f); }
            }
        } stampDump{stampNow, st, stampBuf, (size_t)ta.nTiles * 16, stampPath};
#endif
        if (customTile) {  // the same kernels compiled at run time around the user's statements (deme_jit.h)
            void* argsT[] = {&c->dp, &ta};
            HIPCK(hipModuleLaunchKernel(c->customTileFn[mesh ? 1 : 0], nBlk, 1, 1, DEME_TILE_T, 1, 1, ldsBytes, st, argsT, nullptr));
            if (nBig)
                HIPCK(hipModuleLaunchKernel(c->customTileFn[mesh ? 3 : 2], nBig, 1, 1, DEME_TILE_T, 1, 1, 0, st, argsT, nullptr));
        } else {
            // (model, mesh records, recording) -> the instance of the two kernels
            const int model = c->hp.forceModel == DEME_FORCE_HERTZIAN ? 0 : 1;
            const int which = model * 4 + (mesh ? 2 : 0) + (c->record ? 1 : 0);
            // persistent workgroups (deme_tile_p.h): as many as the chip holds at once, each taking tile after tile from a counter
            static const int persistEnv = getenv("DEME_TILE_PERSIST") ? atoi(getenv("DEME_TILE_PERSIST")) : 0;
            if (persistEnv) {
                const size_t ctrWords = ((size_t)DEME_TILE_P_PARTS + 1u) * 32u;  // (deme_tile_p.h: one set per pass)
                if (!c->tileCtr.p) {
                    if (int rc = ensure(c, c->tileCtr, 3 * ctrWords * 4))
                        return rc;
                    HIPCK(hipMemset(c->tileCtr.p, 0, 3 * ctrWords * 4));
                }
                ta.tileCtr = c->tileCtr.as<uint32_t>() + ctrWords * (size_t)(pass + 1);
            }
            hipError_t perr = hipSuccess;
            auto go = [&](auto tileK, auto bigK) {
                if (persistEnv)
                    perr = (hipError_t)launch_tile_forces_p(which, (unsigned)c->nCU, ldsBytes, st, c->dp, ta);
                else
                    hipLaunchKernelGGL(tileK, dim3(nBlk), dim3(DEME_TILE_T), ldsBytes, st, c->dp, ta);
                if (nBig)
                    hipLaunchKernelGGL(bigK, dim3(nBig), dim3(DEME_TILE_T), 0, st, c->dp, ta);
            };
            switch (which) {
                case 0: go(k_tile_forces<0, false, false>, k_tile_forces_big<0, false, false>); break;
                case 1: go(k_tile_forces<0, false, true>, k_tile_forces_big<0, false, true>); break;
                case 2: go(k_tile_forces<0, true, false>, k_tile_forces_big<0, true, false>); break;
                case 3: go(k_tile_forces<0, true, true>, k_tile_forces_big<0, true, true>); break;
                case 4: go(k_tile_forces<1, false, false>, k_tile_forces_big<1, false, false>); break;
                case 5: go(k_tile_forces<1, false, true>, k_tile_forces_big<1, false, true>); break;
                case 6: go(k_tile_forces<1, true, false>, k_tile_forces_big<1, true, false>); break;
                default: go(k_tile_forces<1, true, true>, k_tile_forces_big<1, true, true>); break;
            }
            if (perr != hipSuccess)
                return fail(c, DEME_ERR_HIP, "k_tile_forces_p: %s", hipGetErrorString(perr));
        }
        c->conValid = true;
        c->conTile = true;
        return DEME_OK;
    }
    c->conTile = false;
    if (int rc = ensure_legacy_lists(c))  // (a list with tile structures that is evaluated by the other kernels after all: recording switched on, ...)
        return rc;

    if (c->record) {
        a.recForce = c->rec[0].as<float>(), a.recTorque = c->rec[1].as<float>(), a.recCPA = c->rec[2].as<float>(),
        a.recCPB = c->rec[3].as<float>();
    }
    {
        ScopedTimer tm(c, "calc_forces");
        unsigned nBlk = grid_for(a.nContacts, DEME_FORCE_BLOCK);
        if (a.xcdGroup)  // the XCD-aware block map is a bijection on multiples of 8 G blocks
            nBlk = (nBlk + 8u * a.xcdGroup - 1u) / (8u * a.xcdGroup) * (8u * a.xcdGroup);
        const dim3 g(nBlk), b(DEME_FORCE_BLOCK);
        // sphere-mesh contacts read no ghost owner (meshes are replicated, not ghosted): all of them go with pass 0
        const bool hasSM = c->nTri > 0 && c->nSM > 0 && pass != 1;
        const dim3 gm(grid_for(std::max<uint32_t>(c->nSM, 1u), DEME_FORCE_BLOCK));
        if (c->hp.forceModel == DEME_FORCE_HERTZIAN) {
            if (hasSM)  // mesh variant first: the hot variant folds its A-side records into the in-block sums
                hipLaunchKernelGGL((k_calc_forces<0, 1>), gm, b, 0, c->stream, c->dp, a);
            if (fastKernel && !c->record)
                hipLaunchKernelGGL((k_forces_fast<0>), g, b, 0, c->stream, c->dp, a);
            else
                hipLaunchKernelGGL((k_calc_forces<0, 0>), g, b, 0, c->stream, c->dp, a);
        } else if (c->hp.forceModel == DEME_FORCE_HERTZIAN_FRICTIONLESS) {
            if (hasSM)
                hipLaunchKernelGGL((k_calc_forces<1, 1>), gm, b, 0, c->stream, c->dp, a);
            if (fastKernel && !c->record)
                hipLaunchKernelGGL((k_forces_fast<1>), g, b, 0, c->stream, c->dp, a);
            else
                hipLaunchKernelGGL((k_calc_forces<1, 0>), g, b, 0, c->stream, c->dp, a);
        }
        else {  // user model: two entry points of the same code object (hot variant, mesh variant)
            void* args0[] = {&c->dp, &a};
            if (hasSM)
                HIPCK(hipModuleLaunchKernel(c->customFn[1], gm.x, 1, 1, DEME_FORCE_BLOCK, 1, 1, 0,
                                            c->stream, args0, nullptr));
            HIPCK(hipModuleLaunchKernel(c->customFn[0], g.x, 1, 1, DEME_FORCE_BLOCK, 1, 1, 0, c->stream, args0, nullptr));
        }
    }
    c->conValid = true;
    return DEME_OK;
}

// The whole step in one launch (deme_tile_step.h).  dry: replay the force evaluation of the step just taken on the buffers of its
// start and leave a / alpha (state downloads).
bool fused_ready(deme_ctx* c) {
    if (!c->fusedList || !c->tileActive || !c->tileEnable || c->arith != DEME_ARITH_FAST || c->record || c->hp.forceModel == DEME_FORCE_CUSTOM ||
        c->prescFn || c->rulesFn || c->nContacts == 0)
        return false;
    if (!c->fusedChecked)
        resolve_heavy_counts(c);
    return c->fusedList && c->nHeavyFree == 0;
}
int launch_fused_step(deme_ctx* c, bool dry) {
    StepArgs sa{};
    TileArgs& ta = sa.t;
    const int cur = dry ? (c->wcCur ^ 1) : c->wcCur;  // (dry: the step is over, the names are swapped already)
    ta.owners = dry ? c->ownersNext.as<OwnerRec>() : c->owners.as<OwnerRec>();
    ta.tInfo = c->tInfo.as<uint2>();
    ta.aStart = c->aStart.as<uint32_t>();
    ta.hList = c->hList.as<uint32_t>(), ta.hCount = c->hCountIn.as<uint32_t>(), ta.org = c->tileOrg.as<int64_t>();
    ta.lOff = c->lOff.as<uint16_t>(), ta.lPos = c->lPos.as<uint16_t>(), ta.lCount = c->lCount.as<uint32_t>();
    ta.wc = c->wc[cur].as<float>();
    ta.nOwners = c->nOwners;
    ta.nTiles = (c->nOwners + DEME_TILE_NB - 1) / DEME_TILE_NB;
    ta.xcdGroup = c->xcdGroup;
    ta.tileBig = c->tileBig.as<uint32_t>(), ta.bigList = c->bigList.as<uint32_t>(), ta.info = c->info.as<uint4>();
    ta.hCap = std::min<uint32_t>(DEME_TILE_HMAX, (c->tileMaxHaloIn + 15u) & ~15u);
    ta.lCap = std::min<uint32_t>(DEME_TILE_LMAX, (c->tileMaxList + 15u) & ~15u);
    ta.nComp = c->nComp, ta.nAnal = c->nAnal, ta.nMass = c->nMassProps;
    sa.ownersNext = dry ? c->owners.as<OwnerRec>() : c->ownersNext.as<OwnerRec>();
    sa.wcNext = c->wc[cur ^ 1].as<float>();
    sa.tInfoIn = c->tInfoIn.as<uint2>(), sa.inContact = c->inContact.as<uint32_t>(), sa.inStart = c->inStart.as<uint32_t>();
    sa.acc = c->acc.as<AccRec>();
    sa.nextAcc = (!dry && c->nextAccPending) ? c->nextAcc.as<AccRec>() : nullptr;
    sa.dry = dry ? 1u : 0u;
    unsigned nBlk = ta.nTiles;
    if (ta.xcdGroup)
        nBlk = (nBlk + 8u * ta.xcdGroup - 1u) / (8u * ta.xcdGroup) * (8u * ta.xcdGroup);
    const uint32_t tabBytes = tile_table_bytes(c->nComp, c->nMat, c->nAnal, c->nMassProps, c->dp.familyTrivial);
    ta.rs16 = tile_record_stride(ta.hCap, ta.lCap, tabBytes, 0);
    ta.swz = ta.rs16 == tile_rec16(0) ? 1u : 0u;
    const uint32_t ldsBytes = tile_lds_bytes(ta.hCap, ta.lCap, tabBytes, ta.rs16);
    {
        ScopedTimer tm(c, dry ? "fused_replay" : "calc_forces");
        if (c->hp.forceModel == DEME_FORCE_HERTZIAN)
            hipLaunchKernelGGL((k_tile_step<0>), dim3(nBlk), dim3(DEME_TILE_T), ldsBytes, c->stream, c->dp, sa);
        else
            hipLaunchKernelGGL((k_tile_step<1>), dim3(nBlk), dim3(DEME_TILE_T), ldsBytes, c->stream, c->dp, sa);
    }
    if (dry)
        return DEME_OK;
    std::swap(c->owners, c->ownersNext);
    c->wcCur ^= 1;
    if (c->nextAccPending) {
        c->nextAccPending = false;
        std::fill(c->hNextAcc.begin(), c->hNextAcc.end(), AccRec{});
        HIPCK(hipMemsetAsync(c->nextAcc.p, 0, (size_t)c->nOwners * sizeof(AccRec), c->stream));
    }
    c->conValid = false, c->conTile = false;
    c->fusedPrevValid = true;
    c->nFusedSteps++;
    c->stepsSinceCD++;
    c->nSteps++;
    c->timeElapsed += (double)c->hp.h;
    if (c->evStepDone)
        HIPCK(hipEventRecord(c->evStepDone, c->stream));
    return DEME_OK;
}

// owners of prescribed families (rebuilt when families were uploaded); host-side: this is a set-up path
int rebuild_presc_list(deme_ctx* c) {
    c->prescDirty = false;
    c->nPresc = 0;
    if (!c->prescFn)
        return DEME_OK;
    std::vector<uint32_t> list, slot(c->nOwners, 0u);
    if (c->rulesFn) {  // families change on the device: every owner may become prescribed, so every owner gets a record
        list.resize(c->nOwners);
        for (uint32_t o = 0; o < c->nOwners; o++)
            list[o] = slot[o] = o;
    } else {
        std::vector<OwnerRec> h(c->nOwners);
        HIPCK(hipMemcpyAsync(h.data(), c->owners.p, (size_t)c->nOwners * sizeof(OwnerRec), hipMemcpyDeviceToHost, c->stream));
        HIPCK(hipStreamSynchronize(c->stream));
        for (uint32_t o = 0; o < c->nOwners; o++)
            if (c->hostFamFlags[h[o].family & 255u] & DEME_FAMILY_PRESCRIBED) {
                slot[o] = (uint32_t)list.size();
                list.push_back(o);
            }
    }
    c->nPresc = (uint32_t)list.size();
    if (int rc = ensure(c, c->prescList, std::max<size_t>(list.size(), 1) * 4))
        return rc;
    if (int rc = ensure(c, c->prescSlot, std::max<size_t>(c->nOwners, 1) * 4))
        return rc;
    if (int rc = ensure(c, c->prescRec, std::max<size_t>(list.size(), 1) * sizeof(PrescRec)))
        return rc;
    if (!list.empty())
        HIPCK(hipMemcpyAsync(c->prescList.p, list.data(), list.size() * 4, hipMemcpyHostToDevice, c->stream));
    HIPCK(hipMemcpyAsync(c->prescSlot.p, slot.data(), slot.size() * 4, hipMemcpyHostToDevice, c->stream));
    HIPCK(hipStreamSynchronize(c->stream));
    return DEME_OK;
}

// `laterIds` (slab group, one evaluation per cross-cut contact): the owners that wait for a reverse share are left out of this launch
// (GatherArgs::revPhase) and integrated by launch_integrate_later once the share has arrived
int launch_integrate(deme_ctx* c, bool fused, bool heavyDone = false, bool splitLater = false) {
    ScopedTimer tm(c, "integrate");
    PrescArgs pa{nullptr, nullptr};
    if (c->prescFn) {
        if (c->prescDirty)
            if (int rc = rebuild_presc_list(c))
                return rc;
        if (c->nPresc) {
            const OwnerRec* ow = c->owners.as<OwnerRec>();
            const uint32_t* list = c->prescList.as<uint32_t>();
            PrescRec* rec = c->prescRec.as<PrescRec>();
            uint32_t n = c->nPresc;
            float t = (float)c->timeElapsed;
            void* args[] = {&c->dp, &ow, &list, &n, &rec, &t};
            HIPCK(hipModuleLaunchKernel(c->prescFn, grid_for(n), 1, 1, 256, 1, 1, 0, c->stream, args, nullptr));
            pa.rec = rec;
            pa.slot = c->prescSlot.as<uint32_t>();
        }
    }
    if (fused) {
        if (!heavyDone)
            launch_reduce_heavy(c, true);
        if (c->heavyOverflow)
            return c->lastStatus;
        GatherArgs ga = gather_args(c);
        ga.revPhase = splitLater ? 1u : 0u;
        hipLaunchKernelGGL(k_integrate<true>, dim3(grid_for(c->nOwners)), dim3(256), 0, c->stream, c->dp,
                           c->owners.as<OwnerRec>(), c->acc.as<AccRec>(), ga, pa);
    } else {
        hipLaunchKernelGGL(k_integrate<false>, dim3(grid_for(c->nOwners)), dim3(256), 0, c->stream, c->dp,
                           c->owners.as<OwnerRec>(), c->acc.as<AccRec>(), gather_args(c), pa);
    }
    c->laterPa = pa;
    if (splitLater)
        return DEME_OK;  // (launch_integrate_later finishes the step)
    if (c->nextAccPending) {  // one step only (cleanUpAcc clears the flag when it honours it, DEMPrepForceKernels.cu:14-31)
        c->nextAccPending = false;
        std::fill(c->hNextAcc.begin(), c->hNextAcc.end(), AccRec{});
        HIPCK(hipMemsetAsync(c->nextAcc.p, 0, (size_t)c->nOwners * sizeof(AccRec), c->stream));
    }
    return DEME_OK;
}
int launch_integrate_later(deme_ctx* c, const uint32_t* ids, uint32_t n) {
    if (n) {
        ScopedTimer tm(c, "integrate_later");
        hipLaunchKernelGGL(k_integrate_list, dim3(grid_for(n)), dim3(256), 0, c->stream, c->dp, c->owners.as<OwnerRec>(),
                           c->acc.as<AccRec>(), gather_args(c), c->laterPa, ids, n);
    }
    if (c->nextAccPending) {
        c->nextAccPending = false;
        std::fill(c->hNextAcc.begin(), c->hNextAcc.end(), AccRec{});
        HIPCK(hipMemsetAsync(c->nextAcc.p, 0, (size_t)c->nOwners * sizeof(AccRec), c->stream));
    }
    return DEME_OK;
}

// a/alpha of every owner from the current contributions (stand-alone force pass and state downloads)
void launch_full_reduction(deme_ctx* c) {
    hipLaunchKernelGGL(k_gather_acc, dim3(grid_for(c->nOwners)), dim3(256), 0, c->stream, c->dp, gather_args(c),
                       c->owners.as<OwnerRec>(), c->acc.as<AccRec>());
    launch_reduce_heavy(c, false);
}

}  // namespace

#include "deme_order.inc"

// ================================================================================================
extern "C" {

const char* deme_version(void) { return "deme_hip 0.1 (gfx950)"; }

int deme_ctx_create(int device, deme_ctx** out) {
    if (!out)
        return DEME_ERR_INVALID;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device >= n)
        return DEME_ERR_HIP;  // reference: DEMSolver construction throws when no device (GpuManager.cpp:64-68)
    if (hipSetDevice(device) != hipSuccess)
        return DEME_ERR_HIP;
    deme_ctx* c = new deme_ctx();
    c->device = device;
    if (hipDeviceGetAttribute(&c->nCU, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || c->nCU <= 0)
        c->nCU = 256;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        return DEME_ERR_HIP;
    }
    if (ensure(c, c->ctr, sizeof(DetectCounters)) || ensure(c, c->scanTmp, 1 << 20)) {
        delete c;
        return DEME_ERR_HIP;
    }
    hipMemsetAsync(c->ctr.p, 0, sizeof(DetectCounters), c->stream);
    if (hipHostMalloc((void**)&c->pin, 16384, hipHostMallocDefault) != hipSuccess) {
        deme_ctx_destroy(c);
        return DEME_ERR_HIP;
    }
    if (const char* e = getenv("DEME_ARITH"))  // process-wide default of deme_set_arith_mode ("exact" / "fast")
        c->arith = (strcmp(e, "exact") == 0) ? DEME_ARITH_EXACT : DEME_ARITH_FAST;
    if (const char* e = getenv("DEME_XCD_GROUP"))  // tuning knob (profiles/): see force_block_id
        c->xcdGroup = (uint32_t)std::max(0, atoi(e));
    if (const char* e = getenv("DEME_TILE"))  // 0: keep the per-contact-block force kernel (A/B measurements)
        c->tileEnable = atoi(e);
    if (const char* e = getenv("DEME_SPIN_SYNC"))
        c->spinSync = atoi(e);
    if (const char* e = getenv("DEME_PASS1_BESIDE"))
        c->pass1Beside = atoi(e);
    if (const char* e = getenv("DEME_KEY_SEG_MIN"))  // tests lower it to put small scenes through the segmented arena; 0 = one segment always
        c->keySegMin = (size_t)std::max(0ll, atoll(e));
    *out = c;
    return DEME_OK;
}

void deme_ctx_destroy(deme_ctx* c) {
    if (!c)
        return;
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    drain_timers(c);
    for (auto e : c->eventPool)
        hipEventDestroy(e);
    for (hipModule_t m : {c->customMod, c->prescMod, c->rulesMod})
        if (m)
            (void)hipModuleUnload(m);
    for (hipModule_t m : c->regionMod)
        if (m)
            (void)hipModuleUnload(m);
    for (hipEvent_t e : {c->evDet0, c->evDet1, c->evWin0, c->evWin1})
        if (e)
            hipEventDestroy(e);
    if (c->haloStream) {
        hipStreamSynchronize(c->haloStream);
        hipEventDestroy(c->evStepDone);
        hipEventDestroy(c->evHaloDone);
        hipStreamDestroy(c->haloStream);
    }
    if (c->pin)
        hipHostFree(c->pin);
    if (c->evPass1)
        hipEventDestroy(c->evPass1);
    for (DevBuf& b : c->spare)
        if (b.p)
            hipFree(b.p);
    if (c->hrPinned) {
        hipHostFree(c->hrPinned);
        hipEventDestroy(c->hrEvent);
    }
    if (c->detStream) {
        hipStreamSynchronize(c->detStream);
        hipEventDestroy(c->evSnap);
        hipEventDestroy(c->evP1);
        hipStreamDestroy(c->detStream);
    }
    DevBuf* all[] = {&c->tileCtr, &c->hCountIn, &c->ownersNext, &c->recContact, &c->inCnt, &c->inStart, &c->tInfoIn, &c->inContact, &c->sphFam, &c->tileBig, &c->bigList, &c->dO2E, &c->dS2E, &c->segCtr, &c->tInfo, &c->hList, &c->hCount, &c->tileMode, &c->tileOrg, &c->rIdx, &c->rStart, &c->remKey[0], &c->remKey[1], &c->lPos, &c->lOff, &c->lCount, &c->tileRem, &c->tileBase, &c->remVal, &c->rankC, &c->rec32, &c->revSlot, &c->nextAcc, &c->binStat, &c->volumes, &c->persistKeys, &c->owners, &c->spheres, &c->acc, &c->conA4, &c->conA2, &c->conB4, &c->conB2, &c->aSum, &c->prescList, &c->prescSlot, &c->prescRec, &c->smFlag, &c->smList, &c->cDefer, &c->blockMode, &c->ownerA, &c->ownerB[0], &c->ownerB[1], &c->bIdx[0], &c->bIdx[1], &c->aStart, &c->bStart, &c->heavy, &c->fixedFlag, &c->heavyList, &c->rangeCtr, &c->info, &c->tris, &c->triWorld, &c->triLo, &c->triHi, &c->triCounts, &c->triOffsets, &c->triKeys[0], &c->triKeys[1], &c->triVals[0], &c->triVals[1], &c->comp, &c->massProps, &c->anal, &c->matPair,
                     &c->E, &c->nu, &c->CoR, &c->mu, &c->Crr, &c->famMasks, &c->famExtra, &c->famFlags, &c->geo,
                     &c->binLo, &c->binN, &c->counts, &c->offsets, &c->incKeys[0], &c->incKeys[1], &c->incVals[0],
                     &c->incVals[1], &c->keysRaw, &c->keysSorted[0], &c->keysSorted[1], &c->mapping, &c->wc[0],
                     &c->wc[1], &c->ctr, &c->scanTmp, &c->sortTmp, &c->rec[0], &c->rec[1], &c->rec[2], &c->rec[3],
                     &c->stage, &c->sharedIds, &c->sharedBuf, &c->keysMid, &c->ownersSnap};
    for (DevBuf* b : all)
        if (b->p)
            hipFree(b->p);
    for (auto& kind : c->userWc)
        for (auto& b : kind)
            if (b.p)
                hipFree(b.p);
    if (c->ownStream && c->stream)
        hipStreamDestroy(c->stream);
    delete c;
}

const char* deme_last_error(const deme_ctx* c) { return c ? c->err.c_str() : "null context"; }

int deme_ctx_set_stream(deme_ctx* c, void* s) {
    if (!c)
        return DEME_ERR_INVALID;
    hipSetDevice(c->device);  // (a process may hold contexts on several devices: deme_multi)
    HIPCK(hipStreamSynchronize(c->stream));
    if (s) {
        if (c->ownStream)
            hipStreamDestroy(c->stream);
        c->stream = (hipStream_t)s;
        c->ownStream = false;
    } else if (!c->ownStream) {
        HIPCK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        c->ownStream = true;
    }
    return DEME_OK;
}

int deme_set_arith_mode(deme_ctx* c, int mode) {
    if (!c || (mode != DEME_ARITH_FAST && mode != DEME_ARITH_EXACT))
        return DEME_ERR_INVALID;
    hipSetDevice(c->device);  // (a process may hold contexts on several devices: deme_multi)
    if (mode != c->arith) {
        HIPCK(hipStreamSynchronize(c->stream));  // (an asynchronous detection cycle ends inside deme_step: nothing is in flight beside the stream)
        c->arith = mode;
        c->conValid = false;  // stored contributions are in the other mode's units
        if (mode == DEME_ARITH_EXACT) {
            // The exact mode's contract is bit-identity with the oracle, which sums an owner's contributions in the CALLER's order:
            // a context the engine had reordered goes back to that order (deme_order.inc: order_restore) and stays there.
            c->orderEligible = false, c->orderRenewDue = false;
            if (c->permuted && c->haveScene)
                if (int rc = order_restore(c))
                    return rc;
        } else if (c->haveScene && c->orderStructOK) {  // back in the fast mode: the next lock-step detection may reorder again
            c->orderEligible = true;
            c->orderRenewDue = true;
        }
    }
    return DEME_OK;
}
int deme_get_arith_mode(const deme_ctx* c) { return c ? c->arith : -1; }

int deme_force_kernel_name(const deme_ctx* c, char* name, size_t cap, uint32_t* tileMaxHalo, uint32_t* tileMaxList) {
    if (!c || !name || !cap)
        return DEME_ERR_INVALID;
    const int m = c->hp.forceModel == DEME_FORCE_HERTZIAN ? 0 : 1;
    const bool fastKernel = c->arith == DEME_ARITH_FAST && c->hp.forceModel != DEME_FORCE_CUSTOM;
    if (c->fusedPrevValid && c->fusedList && c->tileActive)  // the last step went through the one-kernel step (deme_tile_step.h)
        snprintf(name, cap, "k_tile_step<%d>", m);
    else if (c->hp.forceModel == DEME_FORCE_CUSTOM && c->arith == DEME_ARITH_FAST && !c->record && c->customTileFn[0] && c->tileActive && c->tileEnable)
        snprintf(name, cap, "deme_custom_tile<%s>", (c->nTri > 0 && c->nSM > 0) ? "true" : "false");
    else if (c->hp.forceModel == DEME_FORCE_CUSTOM)
        snprintf(name, cap, "deme_custom_forces_ss");
    else if (fastKernel && c->tileActive && c->tileEnable)
        snprintf(name, cap, "k_tile_forces<%d, %s>", m, (c->nTri > 0 && c->nSM > 0) ? "true" : "false");
    else if (fastKernel && !c->record)
        snprintf(name, cap, "k_forces_fast<%d>", m);
    else
        snprintf(name, cap, "k_calc_forces<%d, 0>", m);
    if (tileMaxHalo)
        *tileMaxHalo = c->tileMaxHalo;
    if (tileMaxList)
        *tileMaxList = c->tileMaxList;
    return DEME_OK;
}

int deme_tile_stats(const deme_ctx* c, uint32_t out[4]) {
    if (!c || !out)
        return DEME_ERR_INVALID;
    out[0] = c->tileActive ? (c->nOwners + DEME_TILE_NB - 1) / DEME_TILE_NB : 0u;
    out[1] = c->tileActive ? c->nBigTiles : 0u;
    out[2] = c->tileMaxHalo, out[3] = c->tileMaxList;
    return DEME_OK;
}
int deme_set_tile_policy(deme_ctx* c, uint32_t minContactsPerTileCustom) {
    if (!c)
        return DEME_ERR_INVALID;
    c->tileMinContactsCustom = minContactsPerTileCustom;
    c->listStale = true;
    return DEME_OK;
}
int deme_set_fused_step(deme_ctx* c, int on) {
    if (!c)
        return DEME_ERR_INVALID;
    c->fusedEnable = on == 2 ? 2 : (on ? 1 : 0);
    c->listStale = true;
    return DEME_OK;
}
int deme_get_order(const deme_ctx* c, int* reordered, double spread[2]) {
    if (!c)
        return DEME_ERR_INVALID;
    if (reordered)
        *reordered = c->permuted ? 1 : 0;
    if (spread)
        spread[0] = c->orderSpread[0], spread[1] = c->orderSpread[1];
    return DEME_OK;
}
int deme_order_renewals(const deme_ctx* c, uint64_t* n) {
    if (!c || !n)
        return DEME_ERR_INVALID;
    *n = c->nOrderRenewals;
    return DEME_OK;
}
int deme_renew_order(deme_ctx* c) {
    if (int rc = check_ready(c))
        return rc;
    if (!c->orderEligible || c->arith != DEME_ARITH_FAST)
        return fail(c, DEME_ERR_INVALID, "this scene keeps the caller's order (exact arithmetic mode, ghosts, or reordering switched off)");
    // (the re-keyed list is left as the SEED of the next detection -- order_apply: deme_calc_forces refuses it, a step detects first,
    // the history map and contact records are not offered from it, the sphere geometry is recomputed by that detection)
    return order_renew(c);
}
int deme_order_probe(const DemeParams* p, size_t nClumps, const uint64_t* voxelID, const uint16_t* locX, const uint16_t* locY,
                     const uint16_t* locZ, uint32_t* order, double spread[2]) {
    if (!p || !voxelID || !locX || !locY || !locZ || !spread || !(p->binSize > 0) || !(p->l > 0))
        return DEME_ERR_INVALID;
    std::vector<uint32_t> ord;
    order_compute(*p, nClumps, voxelID, locX, locY, locZ, ord, spread);
    if (order)
        memcpy(order, ord.data(), nClumps * 4);
    return DEME_OK;
}
int deme_set_reorder(deme_ctx* c, int enable) {
    if (!c)
        return DEME_ERR_INVALID;
    c->reorderEnable = enable ? 1 : 0;
    return DEME_OK;
}

int deme_sync(deme_ctx* c) {
    if (!c)
        return DEME_ERR_INVALID;
    hipSetDevice(c->device);  // (a process may hold contexts on several devices: deme_multi)
    HIPCK(hipStreamSynchronize(c->stream));
    return DEME_OK;
}

int deme_set_params(deme_ctx* c, const DemeParams* p) {
    if (!c || !p)
        return DEME_ERR_INVALID;
    if (p->nContactWildcards > DEME_MAX_WILDCARD_NUM)
        return fail(c, DEME_ERR_INVALID, "at most %d contact wildcards", DEME_MAX_WILDCARD_NUM);
    if (p->forceModel == DEME_FORCE_HERTZIAN && p->nContactWildcards != 4)
        return fail(c, DEME_ERR_INVALID, "the Hertzian model carries 4 contact wildcards");
    if (c->haveParams && c->hp.nContactWildcards != p->nContactWildcards && c->haveList)
        return fail(c, DEME_ERR_INVALID, "the wildcard count cannot change once a contact list exists");
    if (p->nbX == 0 || p->nbY == 0 || p->nbZ == 0 || !(p->binSize > 0) || !(p->voxelSize > 0) || !(p->l > 0))
        return fail(c, DEME_ERR_INVALID, "bin / voxel sizing is incomplete");
    c->hp = *p;
    c->timeElapsed = p->timeElapsed;
    c->haveParams = true;
    refresh_dev_params(c);
    return DEME_OK;
}

// ---- the scratch a scene of nO owners / nS spheres needs: ONE statement of it, used by deme_upload_scene and by the slab migration
// (deme_migrate_host.inc), whose slabs change their counts.  (A second, shorter list in the migration once missed sphFam: a slab that
// had grown wrote its ghosts' family words past the end of the buffer.)
static int owner_scratch_for_counts(deme_ctx* c, size_t nO) {
    if (int rc = ensure(c, c->acc, std::max<size_t>(nO, 1) * sizeof(AccRec)))
        return rc;
    if (ensure(c, c->ownersNext, std::max<size_t>(nO, 1) * sizeof(OwnerRec)) || ensure(c, c->inCnt, (nO + 2) * 4) || ensure(c, c->inStart, (nO + 2) * 4))
        return c->lastStatus;
    c->fusedList = c->fusedPrevValid = false;
    HIPCK(hipMemsetAsync(c->acc.p, 0, c->acc.bytes, c->stream));
    if (ensure(c, c->aStart, (nO + 1) * 4) || ensure(c, c->aSum, (nO + 1) * 32) || ensure(c, c->bStart, (nO + 1) * 4) || ensure(c, c->heavy, nO + 1) ||
        ensure(c, c->fixedFlag, nO + 1) || ensure(c, c->heavyList, 4096 * 4) || ensure(c, c->rangeCtr, sizeof(RangeCounters)))
        return c->lastStatus;
    const size_t nTiles = (nO + DEME_TILE_NB - 1) / DEME_TILE_NB + 1;
    if (ensure(c, c->hList, nTiles * DEME_TILE_HMAX * 4) || ensure(c, c->hCount, nTiles * 4) || ensure(c, c->hCountIn, nTiles * 4) ||
        ensure(c, c->tileMode, nTiles * 4) || ensure(c, c->tileOrg, nTiles * 24) || ensure(c, c->rStart, (nO + 1) * 4) ||
        ensure(c, c->lOff, nTiles * (DEME_TILE_NB + 1) * 2) || ensure(c, c->lCount, nTiles * 4) || ensure(c, c->tileRem, (nTiles + 1) * 4) ||
        ensure(c, c->tileBase, (nTiles + 1) * 4) || ensure(c, c->tileBig, nTiles * 4) || ensure(c, c->bigList, nTiles * 4))
        return c->lastStatus;
    HIPCK(hipMemsetAsync(c->hCount.p, 0, c->hCount.bytes, c->stream));
    c->tileActive = c->conTile = false;
    HIPCK(hipMemsetAsync(c->aStart.p, 0, c->aStart.bytes, c->stream));
    HIPCK(hipMemsetAsync(c->bStart.p, 0, c->bStart.bytes, c->stream));
    HIPCK(hipMemsetAsync(c->heavy.p, 0, c->heavy.bytes, c->stream));
    HIPCK(hipMemsetAsync(c->fixedFlag.p, 0, c->fixedFlag.bytes, c->stream));
    HIPCK(hipMemsetAsync(c->rangeCtr.p, 0, sizeof(RangeCounters), c->stream));
    return DEME_OK;
}
static int sphere_scratch_for_counts(deme_ctx* c, size_t nS) {
    if (ensure(c, c->sphFam, std::max<size_t>(nS, 1) * 2) || ensure(c, c->geo, std::max<size_t>(nS, 1) * sizeof(GeoRec)) ||
        ensure(c, c->binLo, std::max<size_t>(nS, 1) * 16) || ensure(c, c->binN, std::max<size_t>(nS, 1) * 8) || ensure(c, c->counts, (nS + 1) * 4) ||
        ensure(c, c->offsets, (nS + 1) * 4))
        return c->lastStatus;
    size_t need = 0;
    HIPCK(rocprim::exclusive_scan(nullptr, need, c->counts.as<uint32_t>(), c->offsets.as<uint32_t>(), 0u, nS + 1, rocprim::plus<uint32_t>(), c->stream));
    return ensure(c, c->scanTmp, need);
}

int deme_upload_scene(deme_ctx* c, const DemeScene* s) {
    if (!c || !s)
        return DEME_ERR_INVALID;
    if (!c->haveParams)
        return fail(c, DEME_ERR_INVALID, "call deme_set_params before deme_upload_scene");
    if (s->nSpheres >= (1u << 31))
        return fail(c, DEME_ERR_INVALID, "sphere ids must fit 31 bits");
    if (s->nOwners >= (1u << 30))
        return fail(c, DEME_ERR_INVALID, "owner ids must fit 30 bits");
    // Spheres are clump-major with ascending owners (kT.cpp:766-797): the A / B roles of a pair follow the owner
    // numbers (= the reference's "smaller sphere id first" exactly under this contract), and seeded history and persistent marks
    // are canonicalised by sphere id.
    for (size_t i = 1; i < s->nSpheres; i++)
        if (s->ownerClumpBody[i] < s->ownerClumpBody[i - 1])
            return fail(c, DEME_ERR_INVALID, "deme_upload_scene: spheres must be clump-major with ascending owners (sphere %zu belongs to owner %u, "
                        "sphere %zu to owner %u)", i - 1, s->ownerClumpBody[i - 1], i, s->ownerClumpBody[i]);
    hipSetDevice(c->device);
    c->nOwners = s->nOwners, c->nOwnerClumps = s->nOwnerClumps, c->nSpheres = s->nSpheres, c->nAnal = s->nAnal;
    c->nextAccPending = false;  // per-owner records of the previous scene do not carry over
    c->hNextAcc.clear();
    c->nMat = s->nMat, c->nComp = s->nComp, c->nMassProps = s->nMassProps;
    const size_t nO = s->nOwners, nS = s->nSpheres;
    // (persistent marks are kept in the engine's ids: they cross a re-upload in the caller's)
    for (auto& k : c->hPersist)
        k = order_key_out(c, k);
    // the engine's own order of clumps and spheres (deme_order.inc): slot k holds the caller's owner o2e(k)
    c->permuted = order_decide(c, s);
    c->viewSerial = ~0ull;
    c->orderRenewDue = false, c->orderBaseHalo = 0;
    auto o2e = [&](size_t k) { return c->permuted ? (size_t)c->hO2E[k] : k; };
    // owners
    std::vector<OwnerRec> ho(nO);
    for (size_t k = 0; k < nO; k++) {
        const size_t i = o2e(k);
        OwnerRec& r = ho[k];
        r.voxelID = s->voxelID[i];
        r.locX = s->locX[i], r.locY = s->locY[i], r.locZ = s->locZ[i];
        r.inertiaOff = s->inertiaPropOffsets ? s->inertiaPropOffsets[i] : 0;
        r.qw = s->oriQw ? s->oriQw[i] : 1.f, r.qx = s->oriQx ? s->oriQx[i] : 0.f, r.qy = s->oriQy ? s->oriQy[i] : 0.f,
        r.qz = s->oriQz ? s->oriQz[i] : 0.f;
        r.vx = s->vX ? s->vX[i] : 0.f, r.vy = s->vY ? s->vY[i] : 0.f, r.vz = s->vZ ? s->vZ[i] : 0.f;
        r.wx = s->omgBarX ? s->omgBarX[i] : 0.f, r.wy = s->omgBarY ? s->omgBarY[i] : 0.f,
        r.wz = s->omgBarZ ? s->omgBarZ[i] : 0.f;
        r.family = s->familyID ? s->familyID[i] : 0;
        r.margin = 0.f;
    }
    if (int rc = upload(c, c->owners, ho.data(), nO))
        return rc;
    if (int rc = owner_scratch_for_counts(c, nO))
        return rc;
    c->nHeavy = c->nHeavyFree = 0;
    c->hrPending = false;
    c->conValid = false;
    // spheres
    std::vector<SphereRec> hs(nS);
    for (size_t k = 0; k < nS; k++) {
        const size_t i = c->permuted ? (size_t)c->hS2E[k] : k;
        hs[k].owner = order_owner_in(c, s->ownerClumpBody[i]);
        hs[k].comp = s->clumpComponentOffset[i];
        hs[k].mat = s->sphereMaterialOffset ? s->sphereMaterialOffset[i] : 0;
    }
    if (int rc = upload(c, c->spheres, hs.data(), nS))
        return rc;
    // tables
    std::vector<float4> hc(s->nComp), hm(s->nMassProps);
    for (uint32_t i = 0; i < s->nComp; i++)
        hc[i] = make_float4(s->CDRelPosX[i], s->CDRelPosY[i], s->CDRelPosZ[i], s->Radii[i]);
    for (uint32_t i = 0; i < s->nMassProps; i++)
        hm[i] = make_float4(s->MassProperties[i], s->moiX[i], s->moiY[i], s->moiZ[i]);
    if (int rc = upload(c, c->comp, hc.data(), hc.size()))
        return rc;
    if (int rc = upload(c, c->massProps, hm.data(), hm.size()))
        return rc;
    std::vector<AnalObj> ha(s->nAnal);
    c->hObjType.assign(s->nAnal, 0);
    for (uint32_t i = 0; i < s->nAnal; i++) {
        AnalObj& a = ha[i];
        memset(&a, 0, sizeof(a));
        a.relx = s->objRelPosX[i], a.rely = s->objRelPosY[i], a.relz = s->objRelPosZ[i];
        a.rotx = s->objRotX[i], a.roty = s->objRotY[i], a.rotz = s->objRotZ[i];
        a.size1 = s->objSize1 ? s->objSize1[i] : 0.f, a.size2 = s->objSize2 ? s->objSize2[i] : 0.f,
        a.size3 = s->objSize3 ? s->objSize3[i] : 0.f;
        a.normal = s->objNormal ? s->objNormal[i] : 1.f;
        a.mass = s->objMass ? s->objMass[i] : 1e6f;
        a.owner = s->objOwner[i];
        a.type = s->objType[i];
        a.mat = s->objMaterial ? s->objMaterial[i] : 0;
        c->hObjType[i] = s->objType[i];
    }
    if (int rc = upload(c, c->anal, ha.data(), ha.size()))
        return rc;
    c->mt.nMat = s->nMat;
    c->mt.E.assign(s->E, s->E + s->nMat);
    c->mt.nu.assign(s->nu, s->nu + s->nMat);
    if (s->CoR) c->mt.CoR.assign(s->CoR, s->CoR + (size_t)s->nMat * s->nMat);
    if (s->mu) c->mt.mu.assign(s->mu, s->mu + (size_t)s->nMat * s->nMat);
    if (s->Crr) c->mt.Crr.assign(s->Crr, s->Crr + (size_t)s->nMat * s->nMat);
    std::vector<MatPair> hp;
    build_mat_pairs(s, hp);
    if (int rc = upload(c, c->matPair, hp.data(), hp.size()))
        return rc;
    const size_t nM = s->nMat;
    if (upload(c, c->E, s->E, nM) || upload(c, c->nu, s->nu, nM) || upload(c, c->CoR, s->CoR, nM * nM) ||
        upload(c, c->mu, s->mu, nM * nM) || upload(c, c->Crr, s->Crr, nM * nM))
        return c->lastStatus;
    if (upload(c, c->famMasks, s->familyMasks, (size_t)DEME_FAMILY_MASK_ENTRIES) ||
        upload(c, c->famExtra, s->familyExtraMarginSize, (size_t)DEME_NUM_FAMILIES) ||
        upload(c, c->famFlags, s->familyFlags, (size_t)DEME_NUM_FAMILIES))
        return c->lastStatus;
    memset(c->hostFamFlags, 0, sizeof(c->hostFamFlags));
    if (s->familyFlags)
        memcpy(c->hostFamFlags, s->familyFlags, DEME_NUM_FAMILIES);
    c->hasGhosts = false;
    c->hShared.clear();
    {   // per-owner flags: bit 0 ghost copy, bit 1 replicated free owner.  Always written: a context may be given a new scene
        std::vector<uint8_t> fl(std::max<size_t>(nO, 16), 0);
        if (s->ownerGhost)
            for (size_t i = 0; i < nO; i++) {
                fl[i] = s->ownerGhost[i] & 3u;
                c->hasGhosts = c->hasGhosts || (fl[i] & 1u);
                if (fl[i] & 2u)
                    c->hShared.push_back((uint32_t)i);
            }
        if (int rc = ensure(c, c->stage, fl.size()))
            return rc;
        HIPCK(hipMemcpyAsync(c->stage.p, fl.data(), nO, hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(k_set_ghost_bits, dim3(grid_for(nO)), dim3(256), 0, c->stream, (uint32_t)nO, c->owners.as<OwnerRec>(),
                           c->stage.as<uint8_t>());
        HIPCK(hipStreamSynchronize(c->stream));  // fl is a local
        if (!c->hShared.empty())
            if (int rc = upload(c, c->sharedIds, c->hShared.data(), c->hShared.size()))
                return rc;
    }
    c->prescDirty = true;
    bool trivial = true;
    if (s->familyMasks)
        for (size_t i = 0; i < DEME_FAMILY_MASK_ENTRIES && trivial; i++)
            trivial = s->familyMasks[i] == 0;
    if (s->familyExtraMarginSize)
        for (size_t i = 0; i < DEME_NUM_FAMILIES && trivial; i++)
            trivial = s->familyExtraMarginSize[i] == 0.f;
    c->dp.familyTrivial = trivial ? 1u : 0u;
    if (c->pairsOnce) {  // the fresh owners carry no passive marks and the reaction slots were sized for the old scene: the group
                         // sets the mode up again before its next step (rev_setup_slab), until then this context evaluates both ways
        c->pairsOnce = false;
        c->revAcc = nullptr;
        c->crossStale = true;
    }
    c->dp.hasGhosts = (c->hasGhosts ? 1u : 0u) | (c->pairsOnce ? 2u : 0u);
    // detection scratch
    if (int rc = sphere_scratch_for_counts(c, nS))
        return rc;
    if (int rc = grow_incidence_arena(c, std::max<size_t>(8 * nS, 4096)))
        return rc;
    c->nTri = s->nTri;  // before the contact arena: its mesh work-list buffers exist only when triangles do
    if (int rc = grow_contact_arena(c, std::max<size_t>(6 * nS, 4096)))
        return rc;
    // triangles
    if (s->nTri) {
        if (!s->ownerMesh || !s->triNode1 || !s->triNode2 || !s->triNode3)
            return fail(c, DEME_ERR_INVALID, "nTri > 0 but triangle arrays are missing");
        std::vector<TriRec> ht(s->nTri);
        for (uint32_t t = 0; t < s->nTri; t++) {
            for (int k = 0; k < 3; k++) {
                ht[t].n1[k] = s->triNode1[3 * t + k];
                ht[t].n2[k] = s->triNode2[3 * t + k];
                ht[t].n3[k] = s->triNode3[3 * t + k];
            }
            ht[t].owner = s->ownerMesh[t];
            ht[t].mat = s->triMaterialOffset ? s->triMaterialOffset[t] : 0;
            ht[t].pad = 0;
            if (ht[t].owner >= s->nOwners)
                return fail(c, DEME_ERR_INVALID, "triangle %u refers to owner %u of %u", t, ht[t].owner, s->nOwners);
        }
        if (int rc = upload(c, c->tris, ht.data(), ht.size()))
            return rc;
        const size_t nT = s->nTri;
        if (ensure(c, c->triWorld, nT * sizeof(TriWorld)) || ensure(c, c->triLo, nT * 16) || ensure(c, c->triHi, nT * 16) ||
            ensure(c, c->triCounts, (nT + 1) * 4) || ensure(c, c->triOffsets, (nT + 1) * 4))
            return c->lastStatus;
        size_t need = 0;
        HIPCK(rocprim::exclusive_scan(nullptr, need, c->triCounts.as<uint32_t>(), c->triOffsets.as<uint32_t>(), 0u, nT + 1,
                                      rocprim::plus<uint32_t>(), c->stream));
        if (int rc = ensure(c, c->scanTmp, need))
            return rc;
        HIPCK(hipStreamSynchronize(c->stream));
    }
    if (!c->hPersist.empty()) {  // persistent marks survive a re-upload (UpdateClumps appends) as long as their ids still exist
        c->hPersist.erase(std::remove_if(c->hPersist.begin(), c->hPersist.end(),
                                         [&](uint64_t k) {
                                             const uint32_t cls = key_class(k);
                                             const uint32_t nB = cls == DEME_KEY_CLASS_SS ? s->nSpheres : cls == DEME_KEY_CLASS_SM ? s->nTri : s->nAnal;
                                             return key_a(k) >= s->nSpheres || key_b(k) >= nB;
                                         }),
                          c->hPersist.end());
        for (auto& k : c->hPersist)
            k = order_key_in(c, k);
        std::sort(c->hPersist.begin(), c->hPersist.end());
        if (!c->hPersist.empty()) {
            if (int rc = ensure(c, c->persistKeys, c->hPersist.size() * 8))
                return rc;
            HIPCK(hipMemcpyAsync(c->persistKeys.p, c->hPersist.data(), c->hPersist.size() * 8, hipMemcpyHostToDevice, c->stream));
        }
    }
    if (int rc = order_upload_maps(c))
        return rc;
    c->haveScene = true;
    c->haveList = false;
    c->mapFresh = false;
    c->nContacts = c->nPrev = c->nWcStored = 0;
    c->listSerial++;
    c->stepsSinceCD = 0;
    refresh_dev_params(c);
    HIPCK(hipStreamSynchronize(c->stream));  // host staging vectors die here
    return DEME_OK;
}

static int owner_state_io(deme_ctx* c, const DemeOwnerState* st, int dir) {
    if (int rc = check_ready(c))
        return rc;
    if (!st)
        return DEME_ERR_INVALID;
    const size_t n = c->nOwners;
    // stage: one device buffer holding every SoA column back to back
    struct Col {
        void* host;
        size_t elem;
        void** dev;
    };
    OwnerSoA soa{};
    Col cols[] = {{st->voxelID, 8, (void**)&soa.voxelID},   {st->locX, 2, (void**)&soa.locX},
                  {st->locY, 2, (void**)&soa.locY},         {st->locZ, 2, (void**)&soa.locZ},
                  {st->oriQw, 4, (void**)&soa.oriQw},       {st->oriQx, 4, (void**)&soa.oriQx},
                  {st->oriQy, 4, (void**)&soa.oriQy},       {st->oriQz, 4, (void**)&soa.oriQz},
                  {st->vX, 4, (void**)&soa.vX},             {st->vY, 4, (void**)&soa.vY},
                  {st->vZ, 4, (void**)&soa.vZ},             {st->omgBarX, 4, (void**)&soa.omgBarX},
                  {st->omgBarY, 4, (void**)&soa.omgBarY},   {st->omgBarZ, 4, (void**)&soa.omgBarZ},
                  {st->aX, 4, (void**)&soa.aX},             {st->aY, 4, (void**)&soa.aY},
                  {st->aZ, 4, (void**)&soa.aZ},             {st->alphaX, 4, (void**)&soa.alphaX},
                  {st->alphaY, 4, (void**)&soa.alphaY},     {st->alphaZ, 4, (void**)&soa.alphaZ},
                  {st->familyID, 1, (void**)&soa.familyID}};
    size_t total = 0;
    for (auto& cl : cols)
        if (cl.host)
            total += ((n * cl.elem + 15) / 16) * 16;
    if (int rc = ensure(c, c->stage, std::max<size_t>(total, 16)))
        return rc;
    size_t off = 0;
    for (auto& cl : cols) {
        if (!cl.host)
            continue;
        *cl.dev = (char*)c->stage.p + off;
        if (dir == 0)
            HIPCK(hipMemcpyAsync(*cl.dev, cl.host, n * cl.elem, hipMemcpyHostToDevice, c->stream));
        off += ((n * cl.elem + 15) / 16) * 16;
    }
    // downloads see a/alpha of EVERY owner (fixed and heavy ones are only reduced on demand)
    if (dir == 1 && c->conValid && c->haveList)
        launch_full_reduction(c);
    else if (dir == 1 && c->fusedPrevValid && c->haveList && (st->aX || st->aY || st->aZ || st->alphaX || st->alphaY || st->alphaZ))
        if (int rc = launch_fused_step(c, true))  // the last step was a fused one: replay its force evaluation on the buffers of its start
            return rc;
    if (dir == 0 && (st->aX || st->aY || st->aZ || st->alphaX || st->alphaY || st->alphaZ))
        c->conValid = false;  // the caller now owns a/alpha
    AccRec* accView = c->acc.as<AccRec>();
    if (n)
        hipLaunchKernelGGL(k_pack_owners, dim3(grid_for(n)), dim3(256), 0, c->stream, (uint32_t)n,
                           c->owners.as<OwnerRec>(), accView, soa, dir, c->dp.o2e);
    if (dir == 1) {
        off = 0;
        for (auto& cl : cols) {
            if (!cl.host)
                continue;
            HIPCK(hipMemcpyAsync(cl.host, (char*)c->stage.p + off, n * cl.elem, hipMemcpyDeviceToHost, c->stream));
            off += ((n * cl.elem + 15) / 16) * 16;
        }
    }
    HIPCK(hipStreamSynchronize(c->stream));
    return DEME_OK;
}

int deme_upload_owner_state(deme_ctx* c, const DemeOwnerState* st) {
    if (c && st && st->familyID)
        c->prescDirty = true;  // owners may have changed family
    if (c)
        c->fusedPrevValid = false;
    if (c && st && (st->voxelID || st->locX || st->locY || st->locZ || st->oriQw || st->oriQx || st->oriQy || st->oriQz || st->vX ||
                    st->vY || st->vZ || st->omgBarX || st->omgBarY || st->omgBarZ || st->familyID))
        c->listStale = true;  // pose, velocity (it sizes the margins) or family (masks) changed under the K-step list
    return owner_state_io(c, st, 0);
}
int deme_download_owner_state(deme_ctx* c, DemeOwnerState* st) { return owner_state_io(c, st, 1); }

int deme_update_tri_nodes(deme_ctx* c, const float* n1, const float* n2, const float* n3) {
    if (int rc = check_ready(c))
        return rc;
    if (!c->nTri || !n1 || !n2 || !n3)
        return fail(c, DEME_ERR_INVALID, "no triangles loaded or null node arrays");
    const size_t bytes = (size_t)c->nTri * 12;
    if (int rc = ensure(c, c->stage, bytes * 3))
        return rc;
    float* d = c->stage.as<float>();
    HIPCK(hipMemcpyAsync(d, n1, bytes, hipMemcpyHostToDevice, c->stream));
    HIPCK(hipMemcpyAsync(d + 3 * (size_t)c->nTri, n2, bytes, hipMemcpyHostToDevice, c->stream));
    HIPCK(hipMemcpyAsync(d + 6 * (size_t)c->nTri, n3, bytes, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_pack_tris, dim3(grid_for(c->nTri)), dim3(256), 0, c->stream, c->nTri, c->tris.as<TriRec>(), d,
                       d + 3 * (size_t)c->nTri, d + 6 * (size_t)c->nTri);
    HIPCK(hipStreamSynchronize(c->stream));
    c->listStale = true;  // the facets moved under the K-step list
    return DEME_OK;
}

int deme_compute_margins(deme_ctx* c, uint32_t drift) {
    if (int rc = check_ready(c))
        return rc;
    return do_margins(c, drift);
}

int deme_set_margins(deme_ctx* c, const float* m) {
    if (int rc = check_ready(c))
        return rc;
    if (!m)
        return DEME_ERR_INVALID;
    if (int rc = ensure(c, c->stage, std::max<size_t>(c->nOwners, 4) * 4))
        return rc;
    HIPCK(hipMemcpyAsync(c->stage.p, m, (size_t)c->nOwners * 4, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_set_margins, dim3(grid_for(c->nOwners)), dim3(256), 0, c->stream, c->nOwners,
                       c->owners.as<OwnerRec>(), c->stage.as<float>(), c->dp.o2e);
    HIPCK(hipStreamSynchronize(c->stream));
    return DEME_OK;
}

int deme_detect_contacts(deme_ctx* c) {
    if (int rc = check_ready(c))
        return rc;
    return do_detect(c);
}

int deme_migrate_history(deme_ctx* c) {
    if (int rc = check_ready(c))
        return rc;
    return do_migrate(c);
}

int deme_calc_forces(deme_ctx* c) {
    if (int rc = check_ready(c))
        return rc;
    if (c->mapFresh)
        if (int rc = do_migrate(c))
            return rc;
    if (!c->haveList || c->seeded)
        return fail(c, DEME_ERR_INVALID, "no contact list yet: call deme_detect_contacts first");
    if (int rc = launch_forces(c))
        return rc;
    launch_full_reduction(c);  // stand-alone call: a/alpha of every owner (prepareAccArrays + forceToAcc)
    return DEME_OK;
}

static int launch_family_rules(deme_ctx* c, const AccRec* accp);

int deme_integrate(deme_ctx* c) {
    if (int rc = check_ready(c))
        return rc;
    if (c->rulesFn)  // routineChecks(): family changes sit between the force evaluation and the integration (dT.cpp:2437-2443)
        if (int rc = launch_family_rules(c, c->acc.as<AccRec>()))  // the staged path always has a/alpha stored
            return rc;
    if (int rc = launch_integrate(c, false))  // from the stored a/alpha
        return rc;
    c->nSteps++;
    c->timeElapsed += (double)c->hp.h;
    return DEME_OK;
}

static int step_tail(deme_ctx* c);
static bool detection_due(deme_ctx* c);

// ---- adaptive controllers (include/deme_hip.h: DemeAdaptive) ----------------------------------------------------------
static void apply_bin_size(deme_ctx* c, double s) {
    DemeParams& h = c->hp;
    h.binSize = s;
    // hostCalcBinNum (DEM/HostSideHelpers.hpp) as SceneBuilder._calc_bin_num / DEMSolver::Initialize use it
    h.nbX = (uint32_t)(h.voxelSize * (double)(1ull << h.nvXp2) / s) + 1;
    h.nbY = (uint32_t)(h.voxelSize * (double)(1ull << h.nvYp2) / s) + 1;
    h.nbZ = (uint32_t)(h.voxelSize * (double)(1ull << h.nvZp2) / s) + 1;
    refresh_dev_params(c);
}

// the detection just timed took `ms` on the device: DEMKinematicThread::calibrateParams (DEM/kT.cpp:43-98)
static void adapt_bin_size(deme_ctx* c, double ms) {
    const DemeAdaptive& a = c->ad;
    c->binAccMs += ms;
    if (++c->binObs < std::max(1u, a.binObserveSteps))
        return;
    const double cur = c->binAccMs / (double)c->binObs, prev = c->binPrevMs;
    c->binAccMs = 0, c->binObs = 0, c->binPrevMs = cur;
    if (prev < 0)
        return;  // first window: nothing to compare with yet
    int dir = (c->binRate > 0.f) - (c->binRate < 0.f);
    if (dir == 0) {  // the reference draws a random direction here; alternate instead (reproducible runs)
        dir = c->binFlip;
        c->binFlip = -c->binFlip;
    }
    float upd = ((cur < prev) ? dir : -dir) * a.binAcc * a.binMaxRate;
    if ((double)c->maxInBin > (double)a.binUpperSafety * (double)c->dp.errOutBinSphNum)
        upd = -1.f * a.binAcc * a.binMaxRate;  // bins too full: the size must start to decrease
    const double nBins = (double)c->hp.nbX * (double)c->hp.nbY * (double)c->hp.nbZ;
    if (nBins > (double)a.binLowerSafety * 4294967295.0)
        upd = 1.f * a.binAcc * a.binMaxRate;  // too many bins for the 32-bit bin id: the size must start to increase
    c->binRate = std::min(a.binMaxRate, std::max(-a.binMaxRate, c->binRate + upd));
    double s = c->hp.binSize;
    if (c->binRate > 0.f)
        s *= 1.0 + (double)c->binRate;
    else
        s /= 1.0 - (double)c->binRate;
    const double nb = (c->hp.voxelSize * (double)(1ull << c->hp.nvXp2) / s + 1) * (c->hp.voxelSize * (double)(1ull << c->hp.nvYp2) / s + 1) *
                      (c->hp.voxelSize * (double)(1ull << c->hp.nvZp2) / s + 1);
    if (nb >= 4294967295.0)
        return;  // would overflow the bin id: keep the size
    apply_bin_size(c, s);
    c->nBinChanges++;
}

// a window of steps ending at a detection took `ms` per step: hill climb on K = cdUpdateFreq
static void adapt_update_freq(deme_ctx* c, double msPerStep) {
    const DemeAdaptive& a = c->ad;
    const double prev = c->freqPrevMs;
    c->freqPrevMs = msPerStep;
    if (prev >= 0 && msPerStep >= prev)
        c->freqDir = -c->freqDir;  // no improvement: turn round
    const uint32_t K = std::max(1u, c->hp.cdUpdateFreq), hi = std::max(1u, a.maxUpdateFreq);
    const uint32_t stepK = std::max(1u, K / 8);
    const uint32_t nK = c->freqDir > 0 ? std::min(hi, K + stepK) : (K > stepK ? K - stepK : 1u);
    if (nK != K) {
        c->hp.cdUpdateFreq = nK;
        refresh_dev_params(c);
        c->nFreqChanges++;
    }
}

static int ensure_adaptive_events(deme_ctx* c) {
    if (c->evDet0)
        return DEME_OK;
    HIPCK(hipEventCreate(&c->evDet0));
    HIPCK(hipEventCreate(&c->evDet1));
    HIPCK(hipEventCreate(&c->evWin0));
    HIPCK(hipEventCreate(&c->evWin1));
    return DEME_OK;
}

// margins + detection + history migration of a step whose list is due, with the controllers' timing around it
static int order_renew(deme_ctx* c);
static int detection_phase(deme_ctx* c) {
    if (c->orderRenewDue && !c->inGroupStep)  // (asynchronous cycles begin and end inside deme_step: none is in flight here)
        if (int rc = order_renew(c))
            return rc;
    const bool adaptive = c->ad.autoBinSize || c->ad.autoUpdateFreq;
    if (adaptive) {
        if (int rc = ensure_adaptive_events(c))
            return rc;
        if (c->ad.autoUpdateFreq && c->hp.cdUpdateFreq > 0) {
            if (c->winOpen && ++c->freqObs >= std::max(1u, c->ad.freqObserveDetections)) {
                HIPCK(hipEventRecord(c->evWin1, c->stream));
                HIPCK(hipEventSynchronize(c->evWin1));
                float ms = 0.f;
                HIPCK(hipEventElapsedTime(&ms, c->evWin0, c->evWin1));
                const uint64_t n = c->nSteps - c->winStartStep;
                if (n)
                    adapt_update_freq(c, (double)ms / (double)n);
                c->winOpen = false;
            }
            if (!c->winOpen) {  // a window starts at a detection and includes it
                HIPCK(hipEventRecord(c->evWin0, c->stream));
                c->winStartStep = c->nSteps, c->freqObs = 0, c->winOpen = true;
            }
        }
        if (c->ad.autoBinSize)
            HIPCK(hipEventRecord(c->evDet0, c->stream));
    }
    if (int rc = do_margins(c, c->hp.cdUpdateFreq))
        return rc;
    if (int rc = do_detect(c))
        return rc;
    if (adaptive && c->ad.autoBinSize) {
        HIPCK(hipEventRecord(c->evDet1, c->stream));
        HIPCK(hipEventSynchronize(c->evDet1));  // (do_detect ended on a synchronisation: nothing is queued behind)
        float ms = 0.f;
        HIPCK(hipEventElapsedTime(&ms, c->evDet0, c->evDet1));
        adapt_bin_size(c, (double)ms);
    }
    if (int rc = do_migrate(c))
        return rc;
    c->stepsSinceCD = 0;
    c->listStale = false;
    return DEME_OK;
}

int deme_set_adaptive(deme_ctx* c, const DemeAdaptive* a) {
    if (!c || !a)
        return DEME_ERR_INVALID;
    if (a->autoBinSize && (!(a->binMaxRate >= 0.f) || !(a->binAcc > 0.f)))
        return fail(c, DEME_ERR_INVALID, "deme_set_adaptive: binMaxRate must be >= 0 and binAcc > 0");
    c->ad = *a;
    c->binAccMs = 0, c->binPrevMs = -1, c->freqPrevMs = -1, c->binRate = 0.f, c->binObs = 0, c->freqObs = 0, c->winOpen = false;
    return DEME_OK;
}

int deme_get_adaptive_state(deme_ctx* c, double* binSize, uint32_t* cdUpdateFreq, uint32_t* nBinChanges, uint32_t* nFreqChanges) {
    if (!c)
        return DEME_ERR_INVALID;
    if (binSize)
        *binSize = c->hp.binSize;
    if (cdUpdateFreq)
        *cdUpdateFreq = c->hp.cdUpdateFreq;
    if (nBinChanges)
        *nBinChanges = c->nBinChanges;
    if (nFreqChanges)
        *nFreqChanges = c->nFreqChanges;
    return DEME_OK;
}

// ---- asynchronous detection ---------------------------------------------------------------------------------------------------
// K - D steps after a list was swapped in (D = asyncLead), the owners are copied, the next D steps are enqueued on the main stream
// with the CURRENT list, and part 1 of the detection runs beside them on its own stream from the copy, with margins for the K + D
// steps that pass between the copy and the end of the new list's service.  When the D steps are enqueued the main stream waits for
// part 1, builds the gather lists (part 2), migrates the history, and carries on with the new list.  The host blocks in part 1's
// two sizing read-backs while the GPU works through the D steps: no second host thread is needed.  Contacts are never missed (the
// margins cover the whole span); the new list holds a few more near-pairs than a lock-step detection would, whose contributions are
// zero, so a trajectory in the exact arithmetic mode is the lock-step one.
static bool async_detection_can_start(deme_ctx* c, uint32_t stepsLeftInCall) {
    const uint32_t K = c->hp.cdUpdateFreq, D = c->asyncLead;
    if (!D || K <= D || stepsLeftInCall < D)
        return false;
    // (a mesh, ghosts and marked contacts are fine: part 1 reads the owner snapshot, static scene data and the marked set, and
    // writes detection scratch only; whatever changes them between detections marks the list stale, and a stale list is rebuilt
    // in lock-step.  Replicated free owners and the adaptive controllers -- which time a detection on the main stream -- stay
    // with the lock-step detection.)
    if (!c->haveList || c->seeded || c->listStale || c->mapFresh || !c->hShared.empty())
        return false;
    if (c->ad.autoBinSize || c->ad.autoUpdateFreq)
        return false;
    return c->stepsSinceCD == K - D;
}
// the three phases of one asynchronous detection: the owner snapshot (on the main stream, between two integrations), part 1 beside
// the next D steps, the swap
static int async_snapshot(deme_ctx* c) {
    if (!c->detStream) {
        HIPCK(hipStreamCreateWithFlags(&c->detStream, hipStreamNonBlocking));
        HIPCK(hipEventCreateWithFlags(&c->evSnap, hipEventDisableTiming));
        HIPCK(hipEventCreateWithFlags(&c->evP1, hipEventDisableTiming));
    }
    if (int rc = ensure(c, c->ownersSnap, (size_t)c->nOwners * sizeof(OwnerRec)))
        return rc;
    HIPCK(hipMemcpyAsync(c->ownersSnap.p, c->owners.p, (size_t)c->nOwners * sizeof(OwnerRec), hipMemcpyDeviceToDevice, c->stream));
    HIPCK(hipEventRecord(c->evSnap, c->stream));
    c->snapPending = false;
    return DEME_OK;
}
static int async_part1(deme_ctx* c, uint64_t* nC) {
    const uint32_t K = c->hp.cdUpdateFreq, D = c->asyncLead;
    HIPCK(hipStreamWaitEvent(c->detStream, c->evSnap, 0));
    ScopedTimer tm(c, "detect_async_part1", true, c->detStream);
    HIPCK(hipMemsetAsync(c->ctr.p, 0, sizeof(DetectCounters), c->detStream));
    hipLaunchKernelGGL(k_margins, dim3(grid_for(c->nOwners)), dim3(256), 0, c->detStream, c->dp, c->ownersSnap.as<OwnerRec>(),
                       K + D, c->ctr.as<DetectCounters>());
    c->ctrFresh = true;
    if (int rc = detect_part1(c, c->detStream, c->ownersSnap.as<OwnerRec>(), true, nC))
        return rc;
    HIPCK(hipEventRecord(c->evP1, c->detStream));
    return DEME_OK;
}
// Everything detect_part2 / build_legacy_lists write and the stepping kernels read.  While the D steps of an asynchronous cycle are
// in flight (they were enqueued with the current buffers' addresses) the names are swapped to a second set, which part 2 fills on
// the detection stream; when the main stream has waited for it, the names already point at the new list.
static std::vector<DevBuf*> list_set(deme_ctx* c) {
    return {&c->mapping, &c->rangeCtr, &c->tileRem, &c->ownerA, &c->ownerB[0], &c->ownerB[1], &c->bIdx[0], &c->bIdx[1], &c->info,
            &c->smFlag, &c->smList, &c->aStart, &c->bStart, &c->tileBase, &c->tInfo, &c->hList, &c->hCount, &c->tileMode, &c->lOff,
            &c->lPos, &c->lCount, &c->rankC, &c->remKey[0], &c->remKey[1], &c->remVal, &c->tileOrg, &c->rIdx, &c->rStart, &c->heavy,
            &c->fixedFlag, &c->heavyList, &c->cDefer, &c->blockMode, &c->tileBig, &c->bigList};
}
// the second set at the sizes of the one in use (sized by the arenas / the scene); nothing in flight reads the spare set
static int async_size_spare(deme_ctx* c) {
    const std::vector<DevBuf*> set = list_set(c);
    static_assert(sizeof(c->spare) / sizeof(c->spare[0]) >= 33, "spare set too small");
    for (size_t k = 0; k < set.size(); k++)
        if (c->spare[k].bytes < set[k]->bytes) {
            if (c->spare[k].p)
                HIPCK(hipFree(c->spare[k].p));
            c->spare[k].p = nullptr, c->spare[k].bytes = 0;
            HIPCK(hipMalloc(&c->spare[k].p, set[k]->bytes));
            c->spare[k].bytes = set[k]->bytes;
        }
    return DEME_OK;
}
// streams, events and buffers of the asynchronous detection, made when it is switched on (a first cycle that allocates a
// gigabyte of list structures inside a short timed run costs more than the detection it hides)
static int async_prepare(deme_ctx* c) {
    if (!c->detStream) {
        HIPCK(hipStreamCreateWithFlags(&c->detStream, hipStreamNonBlocking));
        HIPCK(hipEventCreateWithFlags(&c->evSnap, hipEventDisableTiming));
        HIPCK(hipEventCreateWithFlags(&c->evP1, hipEventDisableTiming));
    }
    if (!c->haveParams || !c->haveScene)
        return DEME_OK;
    if (int rc = ensure(c, c->ownersSnap, (size_t)c->nOwners * sizeof(OwnerRec)))
        return rc;
    if (int rc = ensure(c, c->keysMid, (size_t)c->cntCap * 8))
        return rc;
    return async_size_spare(c);
}
static int async_part2(deme_ctx* c, uint64_t nC) {
    // (every step that reads the current list has been enqueued: from here on the names belong to the list being built)
    const std::vector<DevBuf*> set = list_set(c);
    hipStream_t mainStream = c->stream;
    if (int rc = async_size_spare(c))
        return rc;
    for (size_t k = 0; k < set.size(); k++)
        std::swap(*set[k], c->spare[k]);
    c->stream = c->detStream;
    c->listOwnersSnap = true;
    int rc = DEME_OK;
    {
        ScopedTimer tm(c, "detect_async_part2", true);
        // (no early return between the swap above and the restore below: a failed call must not leave the context launching on
        // the detection stream)
        hipError_t e = hipMemsetAsync(c->heavy.p, 0, c->heavy.bytes, c->stream);
        if (e == hipSuccess)
            e = hipMemsetAsync(c->fixedFlag.p, 0, c->fixedFlag.bytes, c->stream);
        rc = e == hipSuccess ? detect_part2(c, nC) : fail(c, DEME_ERR_HIP, "asynchronous detection, part 2: %s", hipGetErrorString(e));
    }
    c->listOwnersSnap = false;
    c->stream = mainStream;
    if (rc) {  // the half-built set goes back to the spare slots: the names keep the list the steps are using
        for (size_t k = 0; k < set.size(); k++)
            std::swap(*set[k], c->spare[k]);
        return rc;
    }
    HIPCK(hipEventRecord(c->evP1, c->detStream));
    HIPCK(hipStreamWaitEvent(c->stream, c->evP1, 0));
    if (int rc2 = do_migrate(c))  // the history follows its contacts into the new order: with the wildcards the last step left
        return rc2;
    c->stepsSinceCD = 0;
    c->listStale = false;
    c->nAsyncDetections++;
    return DEME_OK;
}
static int async_detection_cycle(deme_ctx* c) {
    const uint32_t D = c->asyncLead;
    // the copy is ordered on the main stream: after the step just integrated, before the next one
    if (int rc = async_snapshot(c))
        return rc;
    for (uint32_t d = 0; d < D; d++) {  // D steps with the current list (its margins were sized for them)
        if (int rc = launch_forces(c))
            return rc;
        if (int rc = step_tail(c))
            return rc;
    }
    uint64_t nC = 0;
    if (int rc = async_part1(c, &nC))
        return rc;
    return async_part2(c, nC);
}

int deme_set_async_detection(deme_ctx* c, uint32_t leadSteps) {
    if (!c)
        return DEME_ERR_INVALID;
    c->asyncLead = leadSteps;
    if (leadSteps) {
        hipSetDevice(c->device);
        return async_prepare(c);
    }
    return DEME_OK;
}

int deme_step(deme_ctx* c, uint32_t nsteps) {
    if (int rc = check_ready(c))
        return rc;
    if (!c->hShared.empty())
        return fail(c, DEME_ERR_INVALID, "this slab holds replicated free owners: step it through deme_halo_group_step, which adds their accelerations up across the slabs");
    for (uint32_t i = 0; i < nsteps; i++) {
        if (c->orderRenewDue && !c->inGroupStep && async_detection_can_start(c, nsteps - i)) {
            // the order is to be renewed: this update is a lock-step one (the renewal is host-driven and needs the context idle)
            if (int rc = detection_phase(c))
                return rc;
        } else if (async_detection_can_start(c, nsteps - i)) {
            if (int rc = async_detection_cycle(c))
                return rc;
            i += c->asyncLead - 1;
            continue;
        }
        if (detection_due(c))
            if (int rc = detection_phase(c))
                return rc;
        if (fused_ready(c)) {  // the whole step in one launch: closed tiles integrate their own owners (deme_tile_step.h)
            if (int rc = launch_fused_step(c, false))
                return rc;
            continue;
        }
        c->fusedPrevValid = false;
        if (int rc = launch_forces(c))
            return rc;
        if (int rc = step_tail(c))
            return rc;
    }
    return DEME_OK;
}

static int launch_family_rules(deme_ctx* c, const AccRec* accp) {
    OwnerRec* ow = c->owners.as<OwnerRec>();
    uint32_t n = c->nOwners;
    float t = (float)c->timeElapsed;
    void* args[] = {&c->dp, &ow, &accp, &n, &t};
    HIPCK(hipModuleLaunchKernel(c->rulesFn, grid_for(n), 1, 1, 256, 1, 1, 0, c->stream, args, nullptr));
    return DEME_OK;
}

// rules + integration + bookkeeping of one step (after the force evaluation), in two halves: everything that produces the
// separately reduced a / alpha (heavy and replicated owners), then the integration.  A slab group adds the replicated owners'
// sums up across slabs between the two (deme_halo_group_step).
static int step_tail_pre(deme_ctx* c) {
    bool fused = true;
    if (c->rulesFn) {  // routineChecks(): applyFamilyChanges between forces and integration (dT.cpp:2437-2443)
        const AccRec* accp = nullptr;
        if (c->rulesNeedAcc) {
            launch_full_reduction(c);
            accp = c->acc.as<AccRec>();
            fused = false;
        }
        if (int rc = launch_family_rules(c, accp))
            return rc;
    }
    c->tailFused = fused;
    if (fused)
        launch_reduce_heavy(c, true);
    if (c->heavyOverflow)
        return c->lastStatus;
    return DEME_OK;
}
// `later` / `nLater` (slab group, one evaluation per cross-cut contact): the owners that wait for a reverse share; this half of the
// step leaves them out and returns before the step's bookkeeping -- step_tail_later finishes it when the share has arrived
static int step_tail_post(deme_ctx* c, bool splitLater = false) {
    if (int rc = launch_integrate(c, c->tailFused, true, splitLater))
        return rc;
    if (splitLater)
        return DEME_OK;
    c->stepsSinceCD++;
    c->nSteps++;
    c->timeElapsed += (double)c->hp.h;
    if (c->evStepDone)
        HIPCK(hipEventRecord(c->evStepDone, c->stream));
    return DEME_OK;
}
static int step_tail(deme_ctx* c) {
    if (int rc = step_tail_pre(c))
        return rc;
    return step_tail_post(c);
}
static int step_tail_later(deme_ctx* c, const uint32_t* ids, uint32_t n) {
    if (int rc = launch_integrate_later(c, ids, n))
        return rc;
    c->stepsSinceCD++;
    c->nSteps++;
    c->timeElapsed += (double)c->hp.h;
    if (c->evStepDone)
        HIPCK(hipEventRecord(c->evStepDone, c->stream));
    return DEME_OK;
}

static bool detection_due(deme_ctx* c) {
    const uint32_t K = c->hp.cdUpdateFreq;
    return !c->haveList || c->seeded || c->listStale || K == 0 || c->stepsSinceCD >= K;
}

static int ensure_halo_stream(deme_ctx* c) {
    if (c->haloStream)
        return DEME_OK;
    HIPCK(hipStreamCreateWithFlags(&c->haloStream, hipStreamNonBlocking));
    HIPCK(hipEventCreateWithFlags(&c->evStepDone, hipEventDisableTiming));
    HIPCK(hipEventCreateWithFlags(&c->evHaloDone, hipEventDisableTiming));
    HIPCK(hipEventRecord(c->evStepDone, c->stream));
    return DEME_OK;
}

int deme_halo_stream(deme_ctx* c, void** stream) {
    if (!c || !stream)
        return DEME_ERR_INVALID;
    if (int rc = ensure_halo_stream(c))
        return rc;
    *stream = (void*)c->haloStream;
    return DEME_OK;
}

int deme_halo_pack_async(deme_ctx* c, const uint32_t* d_ids, uint32_t n, void* d_buf) {
    if (int rc = check_ready(c))
        return rc;
    if (int rc = ensure_halo_stream(c))
        return rc;
    HIPCK(hipStreamWaitEvent(c->haloStream, c->evStepDone, 0));  // the owners' state of the step just integrated
    if (n)
        hipLaunchKernelGGL(k_halo_pack, dim3(grid_for(n)), dim3(256), 0, c->haloStream, n, d_ids, c->owners.as<OwnerRec>(),
                           (GhostRec*)d_buf);
    return DEME_OK;
}

int deme_halo_unpack_async(deme_ctx* c, const uint32_t* d_ids, uint32_t n, const void* d_buf) {
    if (int rc = check_ready(c))
        return rc;
    if (int rc = ensure_halo_stream(c))
        return rc;
    if (n)
        hipLaunchKernelGGL(k_halo_unpack, dim3(grid_for(n)), dim3(256), 0, c->haloStream, n, d_ids, c->owners.as<OwnerRec>(),
                           (const GhostRec*)d_buf);
    HIPCK(hipEventRecord(c->evHaloDone, c->haloStream));
    return DEME_OK;
}

int deme_halo_sync(deme_ctx* c) {
    if (!c)
        return DEME_ERR_INVALID;
    hipSetDevice(c->device);  // (a process may hold contexts on several devices: deme_multi)
    if (c->haloStream)
        HIPCK(hipStreamSynchronize(c->haloStream));
    return DEME_OK;
}

int deme_step_overlap_begin(deme_ctx* c, int* detectionDue) {
    if (int rc = check_ready(c))
        return rc;
    if (int rc = ensure_halo_stream(c))
        return rc;
    c->overlapDetect = detection_due(c) || !c->hasGhosts;
    if (detectionDue)
        *detectionDue = c->overlapDetect ? 1 : 0;
    if (c->overlapDetect)
        return DEME_OK;  // a detection reads the ghosts' new positions: nothing can start before they are in place
    return launch_forces(c, 0);
}

// second half of an overlapped step up to (not including) the tail: the detection if one is due, the remaining force pass
static int overlap_forces(deme_ctx* c) {
    if (int rc = check_ready(c))
        return rc;
    if (int rc = ensure_halo_stream(c))
        return rc;
    // The ghost-dependent pass of a split step touches a few per cent of the tiles: launched behind the interior pass it would run
    // alone on a mostly empty GPU.  In the tile form it goes to the halo stream instead -- which holds the ghosts as soon as its
    // unpack is done -- and its workgroups take the slots the interior pass frees; the integration waits for both.
    const bool pass1Beside = !c->overlapDetect && !c->snapPending && c->tileActive && c->tileEnable && c->arith == DEME_ARITH_FAST &&
                             !c->record && c->hp.forceModel != DEME_FORCE_CUSTOM && c->nContacts != 0 && c->pass1Beside;
    if (pass1Beside) {
        if (!c->evPass1)
            HIPCK(hipEventCreateWithFlags(&c->evPass1, hipEventDisableTiming));
        if (int rc = launch_forces(c, 1, c->haloStream))
            return rc;
        HIPCK(hipEventRecord(c->evPass1, c->haloStream));
        HIPCK(hipStreamWaitEvent(c->stream, c->evPass1, 0));
        return DEME_OK;
    }
    HIPCK(hipStreamWaitEvent(c->stream, c->evHaloDone, 0));
    if (c->snapPending)  // an asynchronous detection starts from this moment: own clumps and ghosts both hold the state of the step just integrated
        if (int rc = async_snapshot(c))
            return rc;
    if (c->overlapDetect) {
        c->overlapDetect = false;
        if (detection_due(c))
            if (int rc = detection_phase(c))
                return rc;
        return launch_forces(c);
    }
    return launch_forces(c, 1);
}

int deme_step_overlap_end(deme_ctx* c) {
    if (c && !c->hShared.empty() && !c->inGroupStep)
        return fail(c, DEME_ERR_INVALID, "this slab holds replicated free owners: step it through deme_halo_group_step, which adds their accelerations up across the slabs");
    if (int rc = overlap_forces(c))
        return rc;
    return step_tail(c);
}

// ================================================================================================
// Slab halo exchange driven from the library: RCCL send / recv of the packed ghost records between face neighbours, on its own
// stream, overlapped with the interior force evaluation (north_star; SURVEY 8e "ncclGroupStart; ncclSend/ncclRecv ...").
// RCCL is bound at run time (dlopen): a process that has PyTorch loaded shares PyTorch's copy, and libdeme_hip.so has no
// link-time dependency on it.
// ================================================================================================
namespace {
typedef struct ncclComm* ncclComm_t;
typedef struct {
    char internal[128];
} ncclUniqueId_t;
struct RcclApi {
    void* lib = nullptr;
    int (*GetUniqueId)(ncclUniqueId_t*) = nullptr;
    int (*CommInitRank)(ncclComm_t*, int, ncclUniqueId_t, int) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    int (*CommCount)(ncclComm_t, int*) = nullptr;
    std::string err;
};
RcclApi* rccl_api() {
    static RcclApi api;
    if (api.lib || !api.err.empty())
        return &api;
    // PyTorch's copy first if the process already holds it, then the ROCm installation's
    void* h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        if (h)
            break;
        h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    }
    if (!h) {
        api.err = std::string("RCCL could not be loaded: ") + (dlerror() ? dlerror() : "librccl.so not found");
        return &api;
    }
    bool ok = true;
    auto sym = [&](const char* n) {
        void* p = dlsym(h, n);
        ok = ok && p;
        return p;
    };
    api.GetUniqueId = (decltype(api.GetUniqueId))sym("ncclGetUniqueId");
    api.CommInitRank = (decltype(api.CommInitRank))sym("ncclCommInitRank");
    api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
    api.GroupStart = (decltype(api.GroupStart))sym("ncclGroupStart");
    api.GroupEnd = (decltype(api.GroupEnd))sym("ncclGroupEnd");
    api.Send = (decltype(api.Send))sym("ncclSend");
    api.Recv = (decltype(api.Recv))sym("ncclRecv");
    api.AllReduce = (decltype(api.AllReduce))sym("ncclAllReduce");
    api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
    api.CommCount = (decltype(api.CommCount))sym("ncclCommCount");
    if (!ok) {
        api.err = "librccl.so lacks one of the entry points (send / recv / all-reduce)";
        return &api;
    }
    api.lib = h;
    return &api;
}
constexpr int kNcclUint8 = 1;  // ncclUint8 (rccl.h: ncclInt8 = 0, ncclUint8 = 1)
constexpr int kNcclFloat32 = 7, kNcclSum = 0;  // rccl.h: ncclFloat32 = 7; ncclSum = 0

struct HaloSide {
    int peerRank = -1;          // rank that holds the neighbouring slab, -1: none (end of the chain)
    deme_ctx* peerLocal = nullptr;  // the neighbour's context when this process holds it too
    void *sendIds = nullptr, *recvIds = nullptr, *sendBuf = nullptr, *recvBuf = nullptr;
    uint32_t nSend = 0, nRecv = 0;
    // one evaluation per cross-cut contact: a / alpha of the sums this slab evaluated for its RIGHT ghosts travel to their owner
    // (right side: nRecv x 32 bytes out), those the left neighbour evaluated for this slab's clumps come back (left side: nSend x
    // 32 bytes in) -- the lists of the forward exchange, read the other way round
    void* revBuf = nullptr;
    uint32_t revCap = 0;
};
struct MigBuf {  // one direction of a migration exchange: clumps, their spheres, history rows (device buffers, counts on the host)
    void *clumps = nullptr, *spheres = nullptr, *rowH = nullptr, *rowW = nullptr, *counts = nullptr;
    void *owc = nullptr, *swc = nullptr;  // user wildcard arrays of the clumps / spheres: [clump][k], [sphere][k] floats
    uint32_t nC = 0, nS = 0, nR = 0;
    bool borrowed = false;  // the buffers belong to a neighbour slab of this process
};
struct MigPool {  // the scratch of one slab's migrations: one device block, kept between calls (three dozen hipMalloc / hipFree
                  // pairs per call cost more than the kernels of a migration)
    void* base = nullptr;
    size_t cap = 0, lastNeed = 0;
};
struct SlabGeom {  // what deme_halo_group_migrate needs to know about a slab (deme_halo_group_set_slab)
    bool set = false;
    void *ownerGid = nullptr, *sphereGid = nullptr;
    size_t gidCapO = 0, gidCapS = 0;
    double xLo = 0, xHi = 0, halo = 0;
    uint32_t nOwn = 0, nGL = 0, nGR = 0, flipMask = 7u;
};
struct HaloSlab {
    deme_ctx* ctx = nullptr;
    SlabGeom geo;
    HaloSide side[2];  // 0 left, 1 right
    hipEvent_t evPacked = nullptr;
    hipEvent_t evAcc = nullptr;  // this slab's share of the replicated owners' a / alpha is in its buffer
    hipEvent_t evRev = nullptr;  // the sums of this slab's right ghosts are packed
    uint32_t nOwnersAtAttach = 0;
    MigPool pool;
};
}  // namespace

struct deme_halo_group {
    RcclApi* api = nullptr;
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
    hipStream_t xstream = nullptr;  // the exchange runs here: after every slab's pack, before every slab's unpack
    hipEvent_t evExchanged = nullptr;
    hipEvent_t evRevDone = nullptr;
    bool pairsOnce = false;  // deme_halo_group_set_cross_contacts: one evaluation per cross-cut contact, reactions sent back
    uint64_t nRevExchanges = 0;
    std::vector<HaloSlab> slabs;
    int axis = 0;                    // the axis the slabs were cut along (deme_halo_group_set_axis / _build): what "left" and "right" mean
    uint32_t firstSlab = 0;          // number of this rank's first slab in the whole chain (deme_halo_group_build)
    std::vector<deme_ctx*> ownedCtx; // contexts deme_halo_group_build created: destroyed with the group
    uint32_t nOwnersGlobal = 0, nClumpsGlobal = 0;
    uint32_t nShared = 0;            // replicated free owners (the same on every slab): a / alpha summed across slabs every step
    void* sharedSum = nullptr;       // nShared x AccRec
    void* agreeBuf = nullptr;        // 16 bytes: the error flag the ranks add up before a collective phase (mig_agree)
    bool anyPersist = false;         // (during a migration) some slab of some rank holds persistent marks: the rows carry them
    hipEvent_t evReduced = nullptr;
    std::string err;
    uint64_t nExchanges = 0, nReductions = 0, bytesPerStep = 0;
    double hostUs[4] = {0, 0, 0, 0};  // host time spent enqueuing: interior pass, pack, RCCL group, unpack + boundary pass + integration
};

namespace {
int gfail(deme_halo_group* g, int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (g)
        g->err = buf;
    return code;
}
#define GHIP(call)                                                                                             \
    do {                                                                                                       \
        hipError_t _e = (call);                                                                                \
        if (_e != hipSuccess)                                                                                  \
            return gfail(g, DEME_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)
#define GNCCL(call)                                                                                             \
    do {                                                                                                        \
        int _r = (call);                                                                                        \
        if (_r != 0)                                                                                            \
            return gfail(g, DEME_ERR_HIP, "%s failed: %s (%s:%d)", #call, g->api->GetErrorString(_r), __FILE__, __LINE__); \
    } while (0)
}  // namespace

int deme_halo_unique_id(unsigned char* id128) {
    RcclApi* api = rccl_api();
    if (!api->lib || !id128)
        return DEME_ERR_HIP;
    ncclUniqueId_t id;
    if (api->GetUniqueId(&id) != 0)
        return DEME_ERR_HIP;
    memcpy(id128, id.internal, 128);
    return DEME_OK;
}

int deme_halo_group_create(const unsigned char* id128, int rank, int world, int device, deme_halo_group** out) {
    if (!out || world < 1 || rank < 0 || rank >= world)
        return DEME_ERR_INVALID;
    *out = nullptr;
    deme_halo_group* g = new deme_halo_group();
    g->api = rccl_api();
    g->rank = rank, g->world = world, g->device = device;
    *out = g;  // returned even on failure so that deme_halo_group_last_error can report
    if (!g->api->lib)
        return gfail(g, DEME_ERR_HIP, "%s", g->api->err.c_str());
    GHIP(hipSetDevice(device));
    ncclUniqueId_t id;
    if (id128)
        memcpy(id.internal, id128, 128);
    else if (world == 1)
        GNCCL(g->api->GetUniqueId(&id));  // a one-rank communicator: sends to self serve the slabs one process holds
    else
        return gfail(g, DEME_ERR_INVALID, "a communicator of %d ranks needs the unique id rank 0 generated", world);
    GNCCL(g->api->CommInitRank(&g->comm, world, id, rank));
    GHIP(hipStreamCreateWithFlags(&g->xstream, hipStreamNonBlocking));
    GHIP(hipEventCreateWithFlags(&g->evExchanged, hipEventDisableTiming));
    return DEME_OK;
}

void deme_halo_group_destroy(deme_halo_group* g) {
    if (!g)
        return;
    hipSetDevice(g->device);
    if (g->xstream)
        hipStreamSynchronize(g->xstream);
    for (auto& s : g->slabs) {
        if (s.ctx && s.ctx->haloStream)
            hipStreamSynchronize(s.ctx->haloStream);
        for (auto& sd : s.side)
            for (void* p : {sd.sendIds, sd.recvIds, sd.sendBuf, sd.recvBuf})
                if (p)
                    hipFree(p);
        if (s.evPacked)
            hipEventDestroy(s.evPacked);
        if (s.evAcc)
            hipEventDestroy(s.evAcc);
        if (s.evRev)
            hipEventDestroy(s.evRev);
        if (s.pool.base)
            hipFree(s.pool.base);
        for (auto& sd : s.side)
            if (sd.revBuf)
                hipFree(sd.revBuf);
        // the slab's books of global ids (deme_halo_group_set_slab / every migration allocates them anew; a decomposed run that is
        // re-planned -- UpdateClumps, ResortClumps, ReplanSlabs -- destroys and rebuilds its groups each time)
        if (s.geo.ownerGid)
            hipFree(s.geo.ownerGid);
        if (s.geo.sphereGid)
            hipFree(s.geo.sphereGid);
        s.geo.ownerGid = s.geo.sphereGid = nullptr;
    }
    if (g->evRevDone)
        hipEventDestroy(g->evRevDone);
    if (g->sharedSum)
        hipFree(g->sharedSum);
    if (g->agreeBuf)
        hipFree(g->agreeBuf);
    if (g->evReduced)
        hipEventDestroy(g->evReduced);
    if (g->comm)
        g->api->CommDestroy(g->comm);
    if (g->evExchanged)
        hipEventDestroy(g->evExchanged);
    if (g->xstream)
        hipStreamDestroy(g->xstream);
    for (deme_ctx* c : g->ownedCtx)
        deme_ctx_destroy(c);
    delete g;
}

const char* deme_halo_group_last_error(const deme_halo_group* g) { return g ? g->err.c_str() : "null halo group"; }

int deme_halo_group_attach(deme_halo_group* g, deme_ctx* c, int leftRank, deme_ctx* leftLocal, const uint32_t* sendLeft,
                           uint32_t nSendLeft, const uint32_t* recvLeft, uint32_t nRecvLeft, int rightRank, deme_ctx* rightLocal,
                           const uint32_t* sendRight, uint32_t nSendRight, const uint32_t* recvRight, uint32_t nRecvRight) {
    if (!g || !c || !g->comm)
        return DEME_ERR_INVALID;
    if (int rc = check_ready(c))
        return gfail(g, rc, "attach: %s", c->err.c_str());
    if (int rc = ensure_halo_stream(c))
        return gfail(g, rc, "attach: %s", c->err.c_str());
    HaloSlab s;
    s.ctx = c;
    GHIP(hipEventCreateWithFlags(&s.evPacked, hipEventDisableTiming));
    GHIP(hipEventCreateWithFlags(&s.evAcc, hipEventDisableTiming));
    {   // replicated free owners: the same list, in the same order, on every slab of every rank (decomp.py keeps them so)
        const uint32_t n = (uint32_t)c->hShared.size();
        if (!g->slabs.empty() && n != g->nShared)
            return gfail(g, DEME_ERR_INVALID, "attach: this slab has %u replicated free owners, the slabs before it %u", n, g->nShared);
        g->nShared = n;
        if (n) {
            if (ensure(c, c->sharedBuf, (size_t)n * sizeof(AccRec)))
                return gfail(g, c->lastStatus, "attach: %s", c->err.c_str());
            if (!g->sharedSum) {
                GHIP(hipMalloc(&g->sharedSum, (size_t)n * sizeof(AccRec)));
                GHIP(hipEventCreateWithFlags(&g->evReduced, hipEventDisableTiming));
            }
        }
    }
    const int peer[2] = {leftRank, rightRank};
    deme_ctx* local[2] = {leftLocal, rightLocal};
    const uint32_t* sIds[2] = {sendLeft, sendRight};
    const uint32_t* rIds[2] = {recvLeft, recvRight};
    const uint32_t nS[2] = {nSendLeft, nSendRight}, nR[2] = {nRecvLeft, nRecvRight};
    for (int k = 0; k < 2; k++) {
        HaloSide& sd = s.side[k];
        sd.peerRank = peer[k], sd.peerLocal = local[k], sd.nSend = nS[k], sd.nRecv = nR[k];
        if (peer[k] < 0)
            continue;
        if (peer[k] >= g->world || (peer[k] == g->rank) != (local[k] != nullptr))
            return gfail(g, DEME_ERR_INVALID, "attach: neighbour rank %d of %d; a neighbour on this rank must be given as a context", peer[k], g->world);
        for (uint32_t i = 0; i < nS[k]; i++)
            if (sIds[k][i] >= c->nOwners)
                return gfail(g, DEME_ERR_INVALID, "attach: send id %u out of range", sIds[k][i]);
        for (uint32_t i = 0; i < nR[k]; i++)
            if (rIds[k][i] >= c->nOwners)
                return gfail(g, DEME_ERR_INVALID, "attach: receive id %u out of range", rIds[k][i]);
        GHIP(hipMalloc(&sd.sendIds, std::max<size_t>(nS[k], 1) * 4));
        GHIP(hipMalloc(&sd.recvIds, std::max<size_t>(nR[k], 1) * 4));
        GHIP(hipMalloc(&sd.sendBuf, std::max<size_t>(nS[k], 1) * sizeof(GhostRec)));
        GHIP(hipMalloc(&sd.recvBuf, std::max<size_t>(nR[k], 1) * sizeof(GhostRec)));
        std::vector<uint32_t> ts, tr;
        const uint32_t *ps = sIds[k], *pr = rIds[k];
        if (c->permuted) {  // (a context with its own order holds no ghosts, so these lists are empty in practice)
            ts.assign(ps, ps + nS[k]), tr.assign(pr, pr + nR[k]);
            for (auto& v : ts)
                v = c->hE2O[v];
            for (auto& v : tr)
                v = c->hE2O[v];
            ps = ts.data(), pr = tr.data();
        }
        if (nS[k])
            GHIP(hipMemcpy(sd.sendIds, ps, (size_t)nS[k] * 4, hipMemcpyHostToDevice));
        if (nR[k])
            GHIP(hipMemcpy(sd.recvIds, pr, (size_t)nR[k] * 4, hipMemcpyHostToDevice));
        g->bytesPerStep += (uint64_t)nS[k] * sizeof(GhostRec);
    }
    s.nOwnersAtAttach = c->nOwners;
    g->slabs.push_back(s);
    return DEME_OK;
}

// one exchange of every slab's ghost records: pack on each slab's halo stream, all transfers in ONE RCCL group on the exchange
// stream, unpack on each slab's halo stream again
static inline double now_us() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
static int halo_exchange(deme_halo_group* g) {
    const double t0 = now_us();
    for (auto& s : g->slabs) {
        deme_ctx* c = s.ctx;
        GHIP(hipStreamWaitEvent(c->haloStream, c->evStepDone, 0));  // the owners' state of the step just integrated
        for (auto& sd : s.side)
            if (sd.peerRank >= 0 && sd.nSend)
                hipLaunchKernelGGL(k_halo_pack, dim3(grid_for(sd.nSend)), dim3(256), 0, c->haloStream, sd.nSend, (const uint32_t*)sd.sendIds,
                                   c->owners.as<OwnerRec>(), (GhostRec*)sd.sendBuf);
        GHIP(hipEventRecord(s.evPacked, c->haloStream));
        GHIP(hipStreamWaitEvent(g->xstream, s.evPacked, 0));
    }
    auto find = [&](deme_ctx* c) -> HaloSlab* {
        for (auto& s : g->slabs)
            if (s.ctx == c)
                return &s;
        return nullptr;
    };
    for (auto& s : g->slabs) {  // every check that can fail comes BEFORE the group opens: an RCCL group left open hangs what follows
        HaloSide& r = s.side[1];
        if (r.peerRank >= 0 && r.peerLocal) {
            HaloSlab* nb = find(r.peerLocal);
            if (!nb)
                return gfail(g, DEME_ERR_INVALID, "a local neighbour context is not attached to the group");
            HaloSide& l = nb->side[0];
            if (l.nRecv != r.nSend || l.nSend != r.nRecv)
                return gfail(g, DEME_ERR_INVALID, "ghost lists of two neighbouring slabs do not match (%u/%u vs %u/%u)", r.nSend, r.nRecv, l.nRecv, l.nSend);
        }
    }
    const double t1 = now_us();
    g->hostUs[1] += t1 - t0;
    GNCCL(g->api->GroupStart());
#define GNCCL_IN_GROUP(call)                                                                                                     \
    do {                                                                                                                         \
        int _r = (call);                                                                                                         \
        if (_r != 0) {                                                                                                           \
            g->api->GroupEnd();                                                                                                  \
            return gfail(g, DEME_ERR_HIP, "%s failed: %s (%s:%d)", #call, g->api->GetErrorString(_r), __FILE__, __LINE__);        \
        }                                                                                                                        \
    } while (0)
    for (auto& s : g->slabs) {
        // the right-hand edge of every slab (each edge once).  A neighbour in this process: the two transfers go to self, and
        // sends to self meet receives from self in posting order -- so each send is posted right before the receive it feeds
        HaloSide& r = s.side[1];
        if (r.peerRank >= 0) {
            if (r.peerLocal) {
                HaloSide& l = find(r.peerLocal)->side[0];  // (checked above)
                if (r.nSend) {
                    GNCCL_IN_GROUP(g->api->Send(r.sendBuf, (size_t)r.nSend * sizeof(GhostRec), kNcclUint8, g->rank, g->comm, g->xstream));
                    GNCCL_IN_GROUP(g->api->Recv(l.recvBuf, (size_t)l.nRecv * sizeof(GhostRec), kNcclUint8, g->rank, g->comm, g->xstream));
                }
                if (l.nSend) {
                    GNCCL_IN_GROUP(g->api->Send(l.sendBuf, (size_t)l.nSend * sizeof(GhostRec), kNcclUint8, g->rank, g->comm, g->xstream));
                    GNCCL_IN_GROUP(g->api->Recv(r.recvBuf, (size_t)r.nRecv * sizeof(GhostRec), kNcclUint8, g->rank, g->comm, g->xstream));
                }
            } else {
                if (r.nSend)
                    GNCCL_IN_GROUP(g->api->Send(r.sendBuf, (size_t)r.nSend * sizeof(GhostRec), kNcclUint8, r.peerRank, g->comm, g->xstream));
                if (r.nRecv)
                    GNCCL_IN_GROUP(g->api->Recv(r.recvBuf, (size_t)r.nRecv * sizeof(GhostRec), kNcclUint8, r.peerRank, g->comm, g->xstream));
            }
        }
        HaloSide& l = s.side[0];
        if (l.peerRank >= 0 && !l.peerLocal) {  // a left neighbour on another rank (a local one was served as its right edge)
            if (l.nSend)
                GNCCL_IN_GROUP(g->api->Send(l.sendBuf, (size_t)l.nSend * sizeof(GhostRec), kNcclUint8, l.peerRank, g->comm, g->xstream));
            if (l.nRecv)
                GNCCL_IN_GROUP(g->api->Recv(l.recvBuf, (size_t)l.nRecv * sizeof(GhostRec), kNcclUint8, l.peerRank, g->comm, g->xstream));
        }
    }
#undef GNCCL_IN_GROUP
    GNCCL(g->api->GroupEnd());
    GHIP(hipEventRecord(g->evExchanged, g->xstream));
    const double t2 = now_us();
    g->hostUs[2] += t2 - t1;
    for (auto& s : g->slabs) {
        deme_ctx* c = s.ctx;
        GHIP(hipStreamWaitEvent(c->haloStream, g->evExchanged, 0));
        for (auto& sd : s.side)
            if (sd.peerRank >= 0 && sd.nRecv)
                hipLaunchKernelGGL(k_halo_unpack, dim3(grid_for(sd.nRecv)), dim3(256), 0, c->haloStream, sd.nRecv, (const uint32_t*)sd.recvIds,
                                   c->owners.as<OwnerRec>(), (const GhostRec*)sd.recvBuf);
        GHIP(hipEventRecord(c->evHaloDone, c->haloStream));
    }
    g->nExchanges++;
    g->hostUs[3] += now_us() - t2;
    return DEME_OK;
}

// ---- one evaluation per cross-cut contact ----------------------------------------------------------------------------------------
// By default a contact between an own clump and a ghost is evaluated on both ranks (each keeps a history copy, the force on the
// ghost is dropped).  With deme_halo_group_set_cross_contacts(g, 1) the LEFT slab of a cut evaluates it -- the right slab marks its
// left ghosts passive (OWNER_PASSIVE_BIT) and the sweep leaves their pairs with own clumps out, along with every ghost sphere's
// wall and mesh contacts -- and after its ghost-dependent force pass it sends a / alpha of each right ghost's contact sum to the
// ghost's owner, which adds them to the clump's own before integrating ("reverse exchange of the ghost's force").
static int rev_setup_slab(deme_halo_group* g, HaloSlab& s) {
    deme_ctx* c = s.ctx;
    c->crossStale = false;
    if (c->nOwners != s.nOwnersAtAttach)
        return gfail(g, DEME_ERR_INVALID, "cross contacts: the slab's scene changed size since it was attached (%u -> %u owners): its exchange "
                                          "lists are stale, attach it to a new group", s.nOwnersAtAttach, c->nOwners);
    c->pairsOnce = g->pairsOnce;
    c->dp.hasGhosts = (c->hasGhosts ? 1u : 0u) | (c->pairsOnce ? 2u : 0u);
    c->revAcc = nullptr;
    HaloSide &l = s.side[0], &r = s.side[1];
    if (l.peerRank >= 0 && l.nRecv)  // my left ghosts: passive or not
        hipLaunchKernelGGL(k_owner_set_bits, dim3(grid_for(l.nRecv)), dim3(256), 0, c->stream, l.nRecv, (const uint32_t*)l.recvIds,
                           c->owners.as<OwnerRec>(), (uint32_t)OWNER_PASSIVE_BIT, g->pairsOnce ? (uint32_t)OWNER_PASSIVE_BIT : 0u);
    c->listStale = true;  // the rule changes what belongs on the list
    if (!g->pairsOnce)
        return DEME_OK;
    if (!s.evRev)
        GHIP(hipEventCreateWithFlags(&s.evRev, hipEventDisableTiming));
    if (!g->evRevDone)
        GHIP(hipEventCreateWithFlags(&g->evRevDone, hipEventDisableTiming));
    const uint32_t want[2] = {l.peerRank >= 0 ? l.nSend : 0u, r.peerRank >= 0 ? r.nRecv : 0u};
    for (int k = 0; k < 2; k++) {
        HaloSide& sd = s.side[k];
        if (want[k] > sd.revCap) {
            if (sd.revBuf)
                GHIP(hipFree(sd.revBuf));
            sd.revCap = want[k] + want[k] / 4 + 64;
            GHIP(hipMalloc(&sd.revBuf, (size_t)sd.revCap * 32));
        }
    }
    if (ensure(c, c->revSlot, std::max<size_t>(c->nOwners, 1) * 4))
        return gfail(g, c->lastStatus, "cross contacts: %s", c->err.c_str());
    GHIP(hipMemsetAsync(c->revSlot.p, 0xFF, std::max<size_t>(c->nOwners, 1) * 4, c->stream));
    if (want[0]) {
        hipLaunchKernelGGL(k_rev_slots, dim3(grid_for(l.nSend)), dim3(256), 0, c->stream, l.nSend, (const uint32_t*)l.sendIds,
                           c->revSlot.as<uint32_t>());
        GHIP(hipMemsetAsync(l.revBuf, 0, (size_t)l.nSend * 32, c->stream));  // (nothing has been received yet)
        c->revAcc = l.revBuf;
    }
    return DEME_OK;
}

int deme_halo_group_set_cross_contacts(deme_halo_group* g, int evaluateOnce) {
    if (!g || !g->comm)
        return DEME_ERR_INVALID;
    GHIP(hipSetDevice(g->device));
    if (evaluateOnce && g->nShared)
        return gfail(g, DEME_ERR_INVALID, "cross contacts: replicated free owners are summed over both evaluations of a cross-cut contact; not combined with the single evaluation yet");
    g->pairsOnce = evaluateOnce != 0;
    for (auto& s : g->slabs) {
        if (int rc = rev_setup_slab(g, s))
            return rc;
        GHIP(hipStreamSynchronize(s.ctx->stream));
    }
    return DEME_OK;
}

// after every slab's ghost-dependent force pass: pack the right ghosts' sums, one RCCL group (each cut: left slab -> right slab),
// the integrations wait for what they receive
// `waitNow` false: the compute streams are NOT made to wait here -- the caller integrates the owners that expect no share beside the
// exchange and waits on evRevDone before the ones that do (one_step)
static int reverse_exchange(deme_halo_group* g, bool waitNow = true) {
    auto find = [&](deme_ctx* c) -> HaloSlab* {
        for (auto& s : g->slabs)
            if (s.ctx == c)
                return &s;
        return nullptr;
    };
    for (auto& s : g->slabs) {
        deme_ctx* c = s.ctx;
        HaloSide& r = s.side[1];
        if (r.peerRank >= 0 && r.nRecv)
            hipLaunchKernelGGL(k_ghost_acc_pack, dim3(grid_for(r.nRecv)), dim3(256), 0, c->stream, c->dp, r.nRecv, (const uint32_t*)r.recvIds,
                               gather_args(c), c->owners.as<OwnerRec>(), (float4*)r.revBuf);
        GHIP(hipEventRecord(s.evRev, c->stream));
        GHIP(hipStreamWaitEvent(g->xstream, s.evRev, 0));
    }
    for (auto& s : g->slabs) {  // (checks before the group opens)
        HaloSide& r = s.side[1];
        if (r.peerRank >= 0 && r.peerLocal && !find(r.peerLocal))
            return gfail(g, DEME_ERR_INVALID, "a local neighbour context is not attached to the group");
    }
    GNCCL(g->api->GroupStart());
#define GNCCL_IN_GROUP(call)                                                                                                     \
    do {                                                                                                                         \
        int _r = (call);                                                                                                         \
        if (_r != 0) {                                                                                                           \
            g->api->GroupEnd();                                                                                                  \
            return gfail(g, DEME_ERR_HIP, "%s failed: %s (%s:%d)", #call, g->api->GetErrorString(_r), __FILE__, __LINE__);        \
        }                                                                                                                        \
    } while (0)
    for (auto& s : g->slabs) {
        HaloSide& r = s.side[1];
        if (r.peerRank >= 0 && r.nRecv) {
            if (r.peerLocal) {  // to self: the send right before the receive it feeds
                HaloSide& l = find(r.peerLocal)->side[0];
                GNCCL_IN_GROUP(g->api->Send(r.revBuf, (size_t)r.nRecv * 32, kNcclUint8, g->rank, g->comm, g->xstream));
                GNCCL_IN_GROUP(g->api->Recv(l.revBuf, (size_t)l.nSend * 32, kNcclUint8, g->rank, g->comm, g->xstream));
            } else {
                GNCCL_IN_GROUP(g->api->Send(r.revBuf, (size_t)r.nRecv * 32, kNcclUint8, r.peerRank, g->comm, g->xstream));
            }
        }
        HaloSide& l = s.side[0];
        if (l.peerRank >= 0 && !l.peerLocal && l.nSend)
            GNCCL_IN_GROUP(g->api->Recv(l.revBuf, (size_t)l.nSend * 32, kNcclUint8, l.peerRank, g->comm, g->xstream));
    }
#undef GNCCL_IN_GROUP
    GNCCL(g->api->GroupEnd());
    GHIP(hipEventRecord(g->evRevDone, g->xstream));
    if (waitNow)
        for (auto& s : g->slabs)
            GHIP(hipStreamWaitEvent(s.ctx->stream, g->evRevDone, 0));
    g->nRevExchanges++;
    return DEME_OK;
}

// Second half of a step when the scene has replicated free owners (a mesh or an analytical body that moves under contact forces,
// kept on every slab): every slab finishes its force passes and reduces its own spheres' contributions to those owners; the
// per-slab sums are added up -- first across the slabs of this process, in slab order, then across the ranks with one all-reduce --
// and every slab integrates its replica with the same total.  (the only all-reduce of the path.)
static int shared_tail(deme_halo_group* g) {
    const uint32_t n = g->nShared, n4 = 2 * n;
    for (auto& s : g->slabs)  // the rules would see each slab's PARTIAL acceleration of a replicated owner and could decide differently per slab
        if (s.ctx->rulesFn && s.ctx->rulesNeedAcc)
            return gfail(g, DEME_ERR_INVALID, "family-change rules that read accelerations cannot be combined with replicated free owners under decomposition");
    for (auto& s : g->slabs) {
        deme_ctx* c = s.ctx;
        if (int rc = overlap_forces(c))
            return gfail(g, rc, "step (boundary forces): %s", c->err.c_str());
        if (int rc = step_tail_pre(c))
            return gfail(g, rc, "step (reductions): %s", c->err.c_str());
        hipLaunchKernelGGL(k_shared_pack, dim3(grid_for(n4)), dim3(256), 0, c->stream, n, c->sharedIds.as<uint32_t>(), c->acc.as<AccRec>(),
                           c->sharedBuf.as<float4>());
        GHIP(hipEventRecord(s.evAcc, c->stream));
        GHIP(hipStreamWaitEvent(g->xstream, s.evAcc, 0));
    }
    GHIP(hipMemcpyAsync(g->sharedSum, g->slabs[0].ctx->sharedBuf.p, (size_t)n * sizeof(AccRec), hipMemcpyDeviceToDevice, g->xstream));
    for (size_t k = 1; k < g->slabs.size(); k++)
        hipLaunchKernelGGL(k_shared_add, dim3(grid_for(n4)), dim3(256), 0, g->xstream, n4, (float4*)g->sharedSum,
                           g->slabs[k].ctx->sharedBuf.as<float4>());
    GNCCL(g->api->AllReduce(g->sharedSum, g->sharedSum, (size_t)n * 8, kNcclFloat32, kNcclSum, g->comm, g->xstream));
    GHIP(hipEventRecord(g->evReduced, g->xstream));
    g->nReductions++;
    for (auto& s : g->slabs) {
        deme_ctx* c = s.ctx;
        GHIP(hipStreamWaitEvent(c->stream, g->evReduced, 0));
        hipLaunchKernelGGL(k_shared_unpack, dim3(grid_for(n4)), dim3(256), 0, c->stream, n, c->sharedIds.as<uint32_t>(), c->acc.as<AccRec>(),
                           (const float4*)g->sharedSum);
        if (int rc = step_tail_post(c))
            return gfail(g, rc, "step (integration): %s", c->err.c_str());
    }
    return DEME_OK;
}

// nsteps time steps of every slab this process holds, each with its ghost exchange: interior force pass on the compute streams
// while the records travel, the ghost-dependent pass and the integration after they have arrived.  Asynchronous like deme_step.
int deme_halo_group_step(deme_halo_group* g, uint32_t nsteps) {
    if (!g || !g->comm)
        return DEME_ERR_INVALID;
    GHIP(hipSetDevice(g->device));
    auto one_step = [&]() -> int {
        const double t0 = now_us();
        for (auto& s : g->slabs)
            if (int rc = deme_step_overlap_begin(s.ctx, nullptr))
                return gfail(g, rc, "step (interior forces): %s", s.ctx->err.c_str());
        g->hostUs[0] += now_us() - t0;
        if (int rc = halo_exchange(g))
            return rc;
        const double t1 = now_us();
        if (g->pairsOnce) {
            for (auto& s : g->slabs)
                if (int rc = overlap_forces(s.ctx))
                    return gfail(g, rc, "step (boundary forces): %s", s.ctx->err.c_str());
            // The reactions travel while every slab integrates the owners that expect none (all but the clumps on its send-left
            // list); those are integrated when the share has arrived.  (Family rules that read accelerations see the complete sums
            // of every owner before anything is integrated: the unsplit order serves then.)
            static const int splitEnv = getenv("DEME_REV_SPLIT") ? atoi(getenv("DEME_REV_SPLIT")) : 1;
            bool split = splitEnv != 0;
            for (auto& s : g->slabs)
                if (s.ctx->rulesFn)
                    split = false;
            if (int rc = reverse_exchange(g, !split))
                return rc;
            if (!split) {
                for (auto& s : g->slabs)
                    if (int rc = step_tail(s.ctx))
                        return gfail(g, rc, "step (integration): %s", s.ctx->err.c_str());
            } else {
                for (auto& s : g->slabs) {
                    if (int rc = step_tail_pre(s.ctx))
                        return gfail(g, rc, "step (reductions): %s", s.ctx->err.c_str());
                    if (int rc = step_tail_post(s.ctx, true))
                        return gfail(g, rc, "step (integration beside the reverse exchange): %s", s.ctx->err.c_str());
                }
                for (auto& s : g->slabs) {
                    const HaloSide& l = s.side[0];
                    GHIP(hipStreamWaitEvent(s.ctx->stream, g->evRevDone, 0));
                    const bool waits = l.peerRank >= 0 && l.nSend && s.ctx->revAcc;
                    if (int rc = step_tail_later(s.ctx, waits ? (const uint32_t*)l.sendIds : nullptr, waits ? l.nSend : 0u))
                        return gfail(g, rc, "step (integration of the clumps that waited for their share): %s", s.ctx->err.c_str());
                }
            }
        } else if (g->nShared == 0) {
            for (auto& s : g->slabs) {
                s.ctx->inGroupStep = true;
                const int rc = deme_step_overlap_end(s.ctx);
                s.ctx->inGroupStep = false;
                if (rc)
                    return gfail(g, rc, "step (boundary forces, integration): %s", s.ctx->err.c_str());
            }
        } else if (int rc = shared_tail(g)) {
            return rc;
        }
        g->hostUs[3] += now_us() - t1;
        return DEME_OK;
    };
    for (auto& s : g->slabs)  // a scene re-uploaded under the single-evaluation rule: passive marks and reaction slots again
        if (s.ctx->crossStale && g->pairsOnce)
            if (int rc = rev_setup_slab(g, s))
                return rc;
    auto clear_snap = [&]() {  // a failed step must not leave a snapshot request behind for the next call
        for (auto& s : g->slabs)
            s.ctx->snapPending = false;
    };
    for (uint32_t i = 0; i < nsteps; i++) {
        // asynchronous detection (deme_set_async_detection on the slabs' contexts): every slab decides for itself -- a detection
        // is local, the exchange of a step does not depend on it.  The slabs whose lists retire D steps from now take their owner
        // snapshot inside the next step (after its ghosts arrive), the D steps are enqueued -- exchanges included: everything is
        // stream-ordered --, and while the GPU works through them the host drives part 1 of those slabs on their detection streams.
        std::vector<deme_ctx*> starters;
        uint32_t D = 0;
        if (g->nShared == 0)
            for (auto& s : g->slabs)
                if (async_detection_can_start(s.ctx, nsteps - i) && (starters.empty() || s.ctx->asyncLead == D)) {
                    D = s.ctx->asyncLead;
                    starters.push_back(s.ctx);
                }
        if (!starters.empty()) {
            for (deme_ctx* c : starters)
                c->snapPending = true;
            for (uint32_t d = 0; d < D; d++)
                if (int rc = one_step()) {
                    clear_snap();
                    return rc;
                }
            std::vector<uint64_t> nC(starters.size(), 0);
            for (size_t k = 0; k < starters.size(); k++)
                if (int rc = async_part1(starters[k], &nC[k]))
                    return gfail(g, rc, "asynchronous detection: %s", starters[k]->err.c_str());
            for (size_t k = 0; k < starters.size(); k++)
                if (int rc = async_part2(starters[k], nC[k]))
                    return gfail(g, rc, "asynchronous detection: %s", starters[k]->err.c_str());
            i += D - 1;
            continue;
        }
        if (int rc = one_step())
            return rc;
    }
    return DEME_OK;
}

int deme_halo_group_exchange(deme_halo_group* g) {  // ghosts refreshed without a step (after uploads; tests)
    if (!g || !g->comm)
        return DEME_ERR_INVALID;
    GHIP(hipSetDevice(g->device));
    for (auto& s : g->slabs)
        if (s.ctx->evStepDone)
            GHIP(hipEventRecord(s.ctx->evStepDone, s.ctx->stream));
    if (int rc = halo_exchange(g))
        return rc;
    for (auto& s : g->slabs)
        GHIP(hipStreamWaitEvent(s.ctx->stream, s.ctx->evHaloDone, 0));
    return DEME_OK;
}

int deme_halo_group_sync(deme_halo_group* g) {
    if (!g)
        return DEME_ERR_INVALID;
    for (auto& s : g->slabs) {
        GHIP(hipStreamSynchronize(s.ctx->stream));
        if (s.ctx->haloStream)
            GHIP(hipStreamSynchronize(s.ctx->haloStream));
    }
    if (g->xstream)
        GHIP(hipStreamSynchronize(g->xstream));
    return DEME_OK;
}

int deme_halo_group_host_time(deme_halo_group* g, double us[4], int reset) {
    if (!g || !us)
        return DEME_ERR_INVALID;
    for (int k = 0; k < 4; k++) {
        us[k] = g->hostUs[k];
        if (reset)
            g->hostUs[k] = 0;
    }
    return DEME_OK;
}

int deme_halo_group_stats(const deme_halo_group* g, uint64_t* exchanges, uint64_t* bytesSentPerStep) {
    if (!g)
        return DEME_ERR_INVALID;
    if (exchanges)
        *exchanges = g->nExchanges;
    if (bytesSentPerStep)
        *bytesSentPerStep = g->bytesPerStep;
    return DEME_OK;
}

int deme_halo_group_comm_count(const deme_halo_group* g, int* ranks) {
    if (!g || !g->comm || !ranks)
        return DEME_ERR_INVALID;
    return g->api->CommCount(g->comm, ranks) == 0 ? DEME_OK : DEME_ERR_HIP;
}

#include "deme_migrate_host.inc"

int deme_get_counts(deme_ctx* c, DemeCounts* out) {
    if (!c || !out)
        return DEME_ERR_INVALID;
    memset(out, 0, sizeof(*out));
    out->nContacts = c->nContacts;
    out->nPrevContacts = c->nPrev;
    out->nBinSphereTouches = c->nInc;
    out->nActiveBins = c->nActiveBins;
    out->nSteps = c->nSteps;
    out->nDetections = c->nDetections;
    out->maxSpheresInBin = c->maxInBin;
    out->lastStatus = c->lastStatus;
    return DEME_OK;
}

int deme_download_bin_incidence(deme_ctx* c, uint32_t* bins, uint32_t* sph, size_t cap) {
    if (int rc = check_ready(c))
        return rc;
    if (cap < c->nInc)
        return fail(c, DEME_ERR_INVALID, "buffer too small: need %llu", (unsigned long long)c->nInc);
    std::vector<uint32_t> hb;
    uint32_t* binsHost = bins;
    if (c->permuted && sph && !bins) {  // (the bins are needed to put each bin's spheres in the caller's order)
        hb.resize(c->nInc);
        binsHost = hb.data();
    }
    if (c->nInc) {
        if (binsHost)
            HIPCK(hipMemcpyAsync(binsHost, c->incKeys[1].p, c->nInc * 4, hipMemcpyDeviceToHost, c->stream));
        if (sph)
            HIPCK(hipMemcpyAsync(sph, c->incVals[1].p, c->nInc * 4, hipMemcpyDeviceToHost, c->stream));
    }
    HIPCK(hipStreamSynchronize(c->stream));
    if (c->permuted && sph) {  // caller's sphere ids, ascending within every bin
        for (size_t i = 0; i < c->nInc; i++)
            sph[i] = c->hS2E[sph[i]];
        for (size_t b = 0; b < c->nInc;) {
            size_t e = b + 1;
            while (e < c->nInc && binsHost[e] == binsHost[b])
                e++;
            std::sort(sph + b, sph + e);
            b = e;
        }
    }
    return DEME_OK;
}

int deme_download_contacts(deme_ctx* c, uint32_t* idA, uint32_t* idB, uint8_t* type, uint32_t* mapping, size_t cap) {
    if (int rc = check_ready(c))
        return rc;
    const size_t n = c->nContacts;
    if (cap < n)
        return fail(c, DEME_ERR_INVALID, "buffer too small: need %zu", n);
    // the list in the caller's ids and canonical order (the engine's own when it keeps the caller's numbering)
    if (int rc = order_current_view(c))
        return rc;
    const std::vector<uint64_t>& k = c->viewKeys;
    if (mapping && n && c->seeded) {  // a seeded list (deme_seed_contacts, a renewed order) has no previous list to map to
        for (size_t i = 0; i < n; i++)
            mapping[i] = 0xFFFFFFFFu;
    } else if (mapping && n) {
        std::vector<uint32_t> m(n);
        HIPCK(hipMemcpyAsync(m.data(), c->mapping.p, n * 4, hipMemcpyDeviceToHost, c->stream));
        HIPCK(hipStreamSynchronize(c->stream));
        if (!c->permuted) {
            memcpy(mapping, m.data(), n * 4);
        } else {  // row i of the caller's view came from which row of the caller's view of the PREVIOUS list
            std::vector<uint64_t> pk;
            std::vector<uint32_t> pperm;
            if (int rc = order_view_of(c, c->keysSorted[c->keysCur ^ 1].as<uint64_t>(), c->nPrev, pk, pperm))
                return rc;
            std::vector<uint32_t> pinv(c->nPrev);
            for (size_t i = 0; i < c->nPrev; i++)
                pinv[pperm[i]] = (uint32_t)i;
            for (size_t i = 0; i < n; i++) {
                const uint32_t from = m[c->viewPerm[i]];
                mapping[i] = from < c->nPrev ? pinv[from] : from;
            }
        }
    }
    for (size_t i = 0; i < n; i++) {
        const uint32_t cls = key_class(k[i]);
        if (idA)
            idA[i] = key_a(k[i]);
        if (idB)
            idB[i] = key_b(k[i]);
        if (type) {
            if (cls == DEME_KEY_CLASS_SS)
                type[i] = DEME_SPHERE_SPHERE_CONTACT;
            else if (cls == DEME_KEY_CLASS_SM)
                type[i] = DEME_SPHERE_MESH_CONTACT;
            else
                type[i] = (c->hObjType[key_b(k[i])] == DEME_ANAL_OBJ_TYPE_PLANE) ? DEME_SPHERE_PLANE_CONTACT
                                                                                   : DEME_SPHERE_CYL_CONTACT;
        }
    }
    return DEME_OK;
}

int deme_download_contact_wildcard(deme_ctx* c, uint32_t w, float* out, size_t cap) {
    if (int rc = check_ready(c))
        return rc;
    const uint32_t nW = c->hp.nContactWildcards;
    if (w >= nW)
        return fail(c, DEME_ERR_INVALID, "wildcard %u out of range (%u)", w, nW);
    if (c->mapFresh)
        if (int rc = do_migrate(c))
            return rc;
    const size_t n = c->nContacts;
    if (cap < n)
        return fail(c, DEME_ERR_INVALID, "buffer too small: need %zu", n);
    std::vector<float> h(n * nW);
    if (n)
        HIPCK(hipMemcpyAsync(h.data(), c->wc[c->wcCur].p, n * nW * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCK(hipStreamSynchronize(c->stream));
    if (c->permuted) {  // rows in the caller's list order
        if (int rc = order_current_view(c))
            return rc;
        for (size_t i = 0; i < n; i++)
            out[i] = h[(size_t)c->viewPerm[i] * nW + w];
        return DEME_OK;
    }
    for (size_t i = 0; i < n; i++)
        out[i] = h[i * nW + w];
    return DEME_OK;
}

int deme_upload_contact_wildcard(deme_ctx* c, uint32_t w, const float* in, size_t n) {
    if (int rc = check_ready(c))
        return rc;
    const uint32_t nW = c->hp.nContactWildcards;
    if (w >= nW || n != c->nContacts)
        return fail(c, DEME_ERR_INVALID, "wildcard upload: index %u of %u, %zu values for %llu contacts", w, nW, n,
                    (unsigned long long)c->nContacts);
    if (c->mapFresh)
        if (int rc = do_migrate(c))
            return rc;
    std::vector<float> h(n * nW);
    if (n)
        HIPCK(hipMemcpyAsync(h.data(), c->wc[c->wcCur].p, n * nW * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCK(hipStreamSynchronize(c->stream));
    if (c->permuted)
        if (int rc = order_current_view(c))
            return rc;
    for (size_t i = 0; i < n; i++)
        h[(c->permuted ? (size_t)c->viewPerm[i] : i) * nW + w] = in[i];
    if (n)
        HIPCK(hipMemcpyAsync(c->wc[c->wcCur].p, h.data(), n * nW * 4, hipMemcpyHostToDevice, c->stream));
    HIPCK(hipStreamSynchronize(c->stream));
    return DEME_OK;
}

int deme_seed_contacts(deme_ctx* c, const uint32_t* idA, const uint32_t* idB, const uint8_t* type, const float* wildcards,
                       size_t n) {
    if (int rc = check_ready(c))
        return rc;
    const uint32_t nW = c->hp.nContactWildcards;
    if (n && (!idA || !idB || !type || (nW && !wildcards)))
        return fail(c, DEME_ERR_INVALID, "deme_seed_contacts: null input");
    std::vector<uint64_t> keys(n);
    std::vector<uint8_t> flip(n, 0);
    for (size_t i = 0; i < n; i++) {
        const uint64_t cls = (type[i] == 1) ? DEME_KEY_CLASS_SS : (type[i] == 2) ? DEME_KEY_CLASS_SM : DEME_KEY_CLASS_SA;
        const uint32_t nB = (cls == DEME_KEY_CLASS_SS) ? c->dp.nSpheres : (cls == DEME_KEY_CLASS_SM) ? c->dp.nTri : c->dp.nAnal;
        if (idA[i] >= c->dp.nSpheres || idB[i] >= nB)
            return fail(c, DEME_ERR_INVALID, "deme_seed_contacts: pair %zu (%u, %u, type %u) is out of range", i, idA[i], idB[i],
                        (unsigned)type[i]);
        uint32_t a = idA[i], b = idB[i];
        if (cls == DEME_KEY_CLASS_SS) {
            if (a == b)
                return fail(c, DEME_ERR_INVALID, "deme_seed_contacts: pair %zu joins sphere %u to itself", i, a);
            if (a > b) {  // the sweep lists a sphere pair smaller id first (ss_key): a pair given the other way round is stored
                          // canonically, and the B-to-A vector history of the built-in model changes sign with the roles
                if (c->hp.forceModel == DEME_FORCE_CUSTOM && nW)
                    return fail(c, DEME_ERR_INVALID,
                                "deme_seed_contacts: pair %zu (%u, %u) must be given smaller sphere id first (the sign rule of a user "
                                "model's wildcards is not known to the library)", i, a, b);
                std::swap(a, b);
                flip[i] = 1;
            }
        }
        keys[i] = order_key_in(c, make_key(cls, a, b));  // (roles by the caller's ids; the engine's slots name the spheres)
    }
    std::vector<uint32_t> perm(n);
    for (size_t i = 0; i < n; i++)
        perm[i] = (uint32_t)i;
    std::sort(perm.begin(), perm.end(), [&](uint32_t x, uint32_t y) { return keys[x] < keys[y]; });
    std::vector<uint64_t> ks(n);
    std::vector<float> ws(n * (size_t)nW);
    for (size_t i = 0; i < n; i++) {
        ks[i] = keys[perm[i]];
        if (i && ks[i] == ks[i - 1])
            return fail(c, DEME_ERR_INVALID, "deme_seed_contacts: duplicate pair (%u, %u)", idA[perm[i]], idB[perm[i]]);
        for (uint32_t w = 0; w < nW; w++)
            ws[i * nW + w] = wildcards[(size_t)perm[i] * nW + w];
        if (flip[perm[i]] && c->hp.forceModel == DEME_FORCE_HERTZIAN)  // delta_tan_x/y/z (FullHertzianForceModel.cu) point B -> A
            for (uint32_t w = 0; w < 3 && w < nW; w++)
                ws[i * nW + w] = -ws[i * nW + w];
    }
    if (n > c->cntCap)
        if (int rc = grow_contact_arena(c, n + n / 4 + 1024))
            return rc;
    if (n) {
        HIPCK(hipMemcpyAsync(c->keysSorted[c->keysCur].p, ks.data(), n * 8, hipMemcpyHostToDevice, c->stream));
        if (nW)
            HIPCK(hipMemcpyAsync(c->wc[c->wcCur].p, ws.data(), n * nW * 4, hipMemcpyHostToDevice, c->stream));
    }
    HIPCK(hipStreamSynchronize(c->stream));
    c->nContacts = n;
    c->listSerial++;
    c->nPrev = 0;
    c->nWcStored = n;
    c->haveList = true;
    c->seeded = true;
    c->mapFresh = false;
    c->conValid = false;
    return DEME_OK;
}

int deme_set_record_contacts(deme_ctx* c, int enable) {
    if (!c)
        return DEME_ERR_INVALID;
    hipSetDevice(c->device);  // (a process may hold contexts on several devices: deme_multi)
    c->record = enable != 0;
    if (c->record && c->cntCap)
        for (int k = 0; k < 4; k++)
            if (int rc = ensure(c, c->rec[k], c->cntCap * 12))
                return rc;
    return DEME_OK;
}

int deme_download_contact_records(deme_ctx* c, float* force, float* torqueOnly, float* cpA, float* cpB, size_t cap) {
    if (int rc = check_ready(c))
        return rc;
    if (!c->record)
        return fail(c, DEME_ERR_INVALID, "contact recording is off (deme_set_record_contacts)");
    if (c->seeded)
        return fail(c, DEME_ERR_INVALID, "the list is a seed (deme_seed_contacts, or the engine's order was just renewed): its contacts have not been evaluated yet -- step or detect first");
    const size_t n = c->nContacts;
    if (cap < n)
        return fail(c, DEME_ERR_INVALID, "buffer too small: need %zu", n);
    float* dst[4] = {force, torqueOnly, cpA, cpB};
    if (c->permuted) {
        if (int rc = order_current_view(c))
            return rc;
        std::vector<float> h(n * 3);
        for (int k = 0; k < 4; k++)
            if (dst[k] && n) {
                HIPCK(hipMemcpyAsync(h.data(), c->rec[k].p, n * 12, hipMemcpyDeviceToHost, c->stream));
                HIPCK(hipStreamSynchronize(c->stream));
                for (size_t i = 0; i < n; i++)
                    memcpy(dst[k] + 3 * i, h.data() + 3 * (size_t)c->viewPerm[i], 12);
            }
        return DEME_OK;
    }
    for (int k = 0; k < 4; k++)
        if (dst[k] && n)
            HIPCK(hipMemcpyAsync(dst[k], c->rec[k].p, n * 12, hipMemcpyDeviceToHost, c->stream));
    HIPCK(hipStreamSynchronize(c->stream));
    return DEME_OK;
}

int deme_download_sphere_geometry(deme_ctx* c, double* X, double* Y, double* Z, float* R, size_t cap) {
    if (int rc = check_ready(c))
        return rc;
    const size_t n = c->nSpheres;
    if (cap < n)
        return fail(c, DEME_ERR_INVALID, "buffer too small: need %zu", n);
    if (c->geoStale)
        return fail(c, DEME_ERR_INVALID, "the engine's order was renewed after the last detection: the sphere geometry of the new slots is computed by the next one");
    std::vector<GeoRec> h(n);
    if (n)
        HIPCK(hipMemcpyAsync(h.data(), c->geo.p, n * sizeof(GeoRec), hipMemcpyDeviceToHost, c->stream));
    HIPCK(hipStreamSynchronize(c->stream));
    for (size_t k = 0; k < n; k++) {
        const size_t i = order_sphere_out(c, (uint32_t)k);
        if (X) X[i] = h[k].x;
        if (Y) Y[i] = h[k].y;
        if (Z) Z[i] = h[k].z;
        if (R) R[i] = h[k].r;
    }
    return DEME_OK;
}

static size_t wc_array_len(deme_ctx* c, uint32_t kind) {
    return kind == 0 ? c->nOwners : kind == 1 ? c->nSpheres : kind == 2 ? c->nTri : c->nAnal;
}

int deme_compile_force_model_ex(deme_ctx* c, const char* src, size_t len, const char* const* wildcardNames, uint32_t nWildcards,
                                const char* const* ownerNames, uint32_t nOwnerWc, const char* const* geoNames, uint32_t nGeoWc,
                                const char* prerequisites) {
    if (!c || !src)
        return DEME_ERR_INVALID;
    if (!c->haveScene)
        return fail(c, DEME_ERR_INVALID, "upload the scene first: material tables are compiled into the model");
    if (nWildcards != c->hp.nContactWildcards)
        return fail(c, DEME_ERR_INVALID, "%u wildcard names given but DemeParams.nContactWildcards is %u", nWildcards,
                    c->hp.nContactWildcards);
    if (nOwnerWc > 8 || nGeoWc > 8)
        return fail(c, DEME_ERR_INVALID, "at most 8 owner and 8 geometry wildcards are supported");
    std::vector<std::string> names, onames, gnames;
    for (uint32_t i = 0; i < nWildcards; i++)
        names.emplace_back(wildcardNames[i] ? wildcardNames[i] : "");
    for (uint32_t i = 0; i < nOwnerWc; i++)
        onames.emplace_back(ownerNames[i] ? ownerNames[i] : "");
    for (uint32_t i = 0; i < nGeoWc; i++)
        gnames.emplace_back(geoNames[i] ? geoNames[i] : "");
    std::string gen, err;
    if (deme_jit::generate_source(std::string(src, len), names, prerequisites ? prerequisites : "", c->mt, gen, err, onames, gnames))
        return fail(c, DEME_ERR_COMPILE, "%s", err.c_str());
    const size_t key = std::hash<std::string>{}(gen);
    auto it = c->jitCache.find(key);
    if (it == c->jitCache.end()) {
        std::vector<char> code;
        std::string log;
        if (deme_jit::compile(gen, code, log))
            return fail(c, DEME_ERR_COMPILE, "force model failed to compile:\n%.900s", log.c_str());
        it = c->jitCache.emplace(key, std::move(code)).first;
    }
    HIPCK(hipStreamSynchronize(c->stream));
    // wildcard arrays: existing ones keep their values, new ones start at zero
    for (uint32_t kind = 0; kind < 4; kind++) {
        const uint32_t want = kind == 0 ? nOwnerWc : nGeoWc;
        const size_t n = wc_array_len(c, kind);
        for (uint32_t k = 0; k < want; k++)
            if (!c->userWc[kind][k].p && n) {
                if (int rc = ensure(c, c->userWc[kind][k], n * 4))
                    return rc;
                HIPCK(hipMemsetAsync(c->userWc[kind][k].p, 0, n * 4, c->stream));
            }
    }
    c->nOwnerWc = nOwnerWc, c->nGeoWc = nGeoWc;
    if (c->customMod) {
        (void)hipModuleUnload(c->customMod);
        c->customMod = nullptr;
        c->customFn[0] = c->customFn[1] = c->customFn[2] = nullptr;
        c->customTileFn[0] = c->customTileFn[1] = c->customTileFn[2] = c->customTileFn[3] = nullptr;
    }
    HIPCK(hipModuleLoadData(&c->customMod, it->second.data()));
    HIPCK(hipModuleGetFunction(&c->customFn[0], c->customMod, "deme_custom_forces_ss"));
    HIPCK(hipModuleGetFunction(&c->customFn[1], c->customMod, "deme_custom_forces_sm"));
    bool tileAll = true;
    for (int k = 0; k < 4; k++)
        if (hipModuleGetFunction(&c->customTileFn[k], c->customMod, deme_jit::kTileEntry[k]) != hipSuccess)
            tileAll = false;
    if (!tileAll)  // all four (tile / big-tile fallback, with and without mesh records) or none: a list with a tile that does not fit
                   // LDS, or a scene with a mesh, must never find its entry missing -- the general kernel evaluates the lists then
        c->customTileFn[0] = c->customTileFn[1] = c->customTileFn[2] = c->customTileFn[3] = nullptr;
    c->listStale = true;  // the list structures depend on which kernel evaluates the list
    return DEME_OK;
}

int deme_compile_force_model(deme_ctx* c, const char* src, size_t len, const char* const* wildcardNames, uint32_t nWildcards,
                             const char* prerequisites) {
    return deme_compile_force_model_ex(c, src, len, wildcardNames, nWildcards, nullptr, 0, nullptr, 0, prerequisites);
}

int deme_upload_wildcard_array(deme_ctx* c, uint32_t kind, uint32_t index, const float* in, size_t n) {
    if (int rc = check_ready(c))
        return rc;
    if (kind > 3 || index >= 8 || !in)
        return fail(c, DEME_ERR_INVALID, "wildcard array: kind %u index %u", kind, index);
    if (n != wc_array_len(c, kind))
        return fail(c, DEME_ERR_INVALID, "wildcard array of kind %u holds %zu values, %zu given", kind, wc_array_len(c, kind), n);
    if (int rc = ensure(c, c->userWc[kind][index], std::max<size_t>(n, 1) * 4))
        return rc;
    // (owner and sphere wildcard arrays are indexed by the ids user code sees -- the caller's -- and stay in the caller's order
    // whatever order the engine keeps the owners in: deme_order.inc)
    if (n)
        HIPCK(hipMemcpyAsync(c->userWc[kind][index].p, in, n * 4, hipMemcpyHostToDevice, c->stream));
    HIPCK(hipStreamSynchronize(c->stream));
    return DEME_OK;
}

int deme_download_wildcard_array(deme_ctx* c, uint32_t kind, uint32_t index, float* out, size_t cap) {
    if (int rc = check_ready(c))
        return rc;
    if (kind > 3 || index >= 8 || !out || !c->userWc[kind][index].p)
        return fail(c, DEME_ERR_INVALID, "wildcard array: kind %u index %u does not exist", kind, index);
    const size_t n = wc_array_len(c, kind);
    if (cap < n)
        return fail(c, DEME_ERR_INVALID, "buffer too small: need %zu", n);
    if (n)
        HIPCK(hipMemcpyAsync(out, c->userWc[kind][index].p, n * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCK(hipStreamSynchronize(c->stream));
    return DEME_OK;
}

int deme_compile_prescriptions(deme_ctx* c, const char* velCases, const char* posCases, const char* accCases) {
    if (!c)
        return DEME_ERR_INVALID;
    hipSetDevice(c->device);  // (a process may hold contexts on several devices: deme_multi)
    HIPCK(hipStreamSynchronize(c->stream));
    if (c->prescMod) {
        (void)hipModuleUnload(c->prescMod);
        c->prescMod = nullptr;
        c->prescFn = nullptr;
    }
    c->prescDirty = true;
    const std::string v = velCases ? velCases : "", ps = posCases ? posCases : "", a = accCases ? accCases : "";
    if (v.empty() && ps.empty() && a.empty())
        return DEME_OK;
    std::string gen;
    deme_jit::generate_prescribe_source(v, ps, a, gen);
    const size_t key = std::hash<std::string>{}(gen);
    auto it = c->jitCache.find(key);
    if (it == c->jitCache.end()) {
        std::vector<char> code;
        std::string log;
        if (deme_jit::compile(gen, code, log))
            return fail(c, DEME_ERR_COMPILE, "family prescriptions failed to compile:\n%.900s", log.c_str());
        it = c->jitCache.emplace(key, std::move(code)).first;
    }
    HIPCK(hipModuleLoadData(&c->prescMod, it->second.data()));
    HIPCK(hipModuleGetFunction(&c->prescFn, c->prescMod, "deme_prescribe"));
    return DEME_OK;
}

int deme_compile_family_rules(deme_ctx* c, const char* rules) {
    if (!c)
        return DEME_ERR_INVALID;
    hipSetDevice(c->device);  // (a process may hold contexts on several devices: deme_multi)
    HIPCK(hipStreamSynchronize(c->stream));
    if (c->rulesMod) {
        (void)hipModuleUnload(c->rulesMod);
        c->rulesMod = nullptr;
        c->rulesFn = nullptr;
    }
    c->prescDirty = true;
    const std::string r = rules ? rules : "";
    if (r.find_first_not_of(" \t\n") == std::string::npos)
        return DEME_OK;
    // whole-word scan for the contact acceleration, as the reference scans force models for their ingredients
    auto mentions = [&](const std::string& w) {
        for (size_t pos = r.find(w); pos != std::string::npos; pos = r.find(w, pos + 1)) {
            const bool l = pos == 0 || !(isalnum((unsigned char)r[pos - 1]) || r[pos - 1] == '_');
            const size_t e = pos + w.size();
            const bool rr = e >= r.size() || !(isalnum((unsigned char)r[e]) || r[e] == '_');
            if (l && rr)
                return true;
        }
        return false;
    };
    c->rulesNeedAcc = mentions("acc") || mentions("accX") || mentions("accY") || mentions("accZ");
    std::string gen;
    deme_jit::generate_family_rules_source(r, gen);
    const size_t key = std::hash<std::string>{}(gen);
    auto it = c->jitCache.find(key);
    if (it == c->jitCache.end()) {
        std::vector<char> code;
        std::string log;
        if (deme_jit::compile(gen, code, log))
            return fail(c, DEME_ERR_COMPILE, "family change rules failed to compile:\n%.900s", log.c_str());
        it = c->jitCache.emplace(key, std::move(code)).first;
    }
    HIPCK(hipModuleLoadData(&c->rulesMod, it->second.data()));
    HIPCK(hipModuleGetFunction(&c->rulesFn, c->rulesMod, "deme_family_changes"));
    return DEME_OK;
}

int deme_add_owner_acc(deme_ctx* c, uint32_t owner, uint32_t n, const float* acc, const float* angAcc) {
    if (int rc = check_ready(c))
        return rc;
    if ((uint64_t)owner + n > c->nOwners || (!acc && !angAcc))
        return fail(c, DEME_ERR_INVALID, "deme_add_owner_acc: owners [%u, %u) out of range or nothing to set", owner, owner + n);
    if (c->hNextAcc.size() != c->nOwners) {
        c->hNextAcc.assign(c->nOwners, AccRec{});
        if (int rc = ensure(c, c->nextAcc, std::max<size_t>(c->nOwners, 1) * sizeof(AccRec)))
            return rc;
        HIPCK(hipMemsetAsync(c->nextAcc.p, 0, (size_t)c->nOwners * sizeof(AccRec), c->stream));
    }
    uint32_t lo = 0xFFFFFFFFu, hi = 0u;
    for (uint32_t k = 0; k < n; k++) {
        const uint32_t slot = order_owner_in(c, owner + k);
        AccRec& r = c->hNextAcc[slot];
        if (acc)
            r.ax = acc[3 * k], r.ay = acc[3 * k + 1], r.az = acc[3 * k + 2];
        if (angAcc)
            r.lx = angAcc[3 * k], r.ly = angAcc[3 * k + 1], r.lz = angAcc[3 * k + 2];
        lo = std::min(lo, slot), hi = std::max(hi, slot + 1u);
    }
    if (n)  // one copy of the touched range of slots (the caller's range is not a range of slots on a reordered context: the host
            // mirror holds every record, so what lies between the touched ones is written back unchanged)
        HIPCK(hipMemcpyAsync(c->nextAcc.as<AccRec>() + lo, c->hNextAcc.data() + lo, (size_t)(hi - lo) * sizeof(AccRec),
                             hipMemcpyHostToDevice, c->stream));
    HIPCK(hipStreamSynchronize(c->stream));
    c->nextAccPending = true;
    return DEME_OK;
}

int deme_mark_persistent_contacts(deme_ctx* c, int mode, uint32_t N1, uint32_t N2, int mark) {
    if (int rc = check_ready(c))
        return rc;
    if (mode < 0 || mode > 3)
        return fail(c, DEME_ERR_INVALID, "deme_mark_persistent_contacts: mode %d is not one of 0 (all), 1 (either), 2 (both), 3 (pair)", mode);
    if (c->hp.nContactWildcards == 0)  // DEM/APIPrivate.cpp:38-43
        return fail(c, DEME_ERR_INVALID,
                    "persistent contacts cannot be marked with a history-less force model (persistency is part of the history); add a placeholder wildcard");
    const size_t nC = c->haveList ? c->nContacts : 0;
    std::vector<uint64_t> hit;
    if (nC) {
        if (int rc = ensure(c, c->stage, nC))
            return rc;
        hipLaunchKernelGGL(k_persist_flags, dim3(grid_for(nC)), dim3(256), 0, c->stream, c->dp, (uint32_t)nC,
                           c->keysSorted[c->keysCur].as<uint64_t>(), c->spheres.as<SphereRec>(), c->owners.as<OwnerRec>(), mode, N1, N2,
                           c->stage.as<uint8_t>());
        std::vector<uint8_t> flag(nC);
        std::vector<uint64_t> keys(nC);
        HIPCK(hipMemcpyAsync(flag.data(), c->stage.p, nC, hipMemcpyDeviceToHost, c->stream));
        HIPCK(hipMemcpyAsync(keys.data(), c->keysSorted[c->keysCur].p, nC * 8, hipMemcpyDeviceToHost, c->stream));
        HIPCK(hipStreamSynchronize(c->stream));
        for (size_t i = 0; i < nC; i++)
            if (flag[i])
                hit.push_back(keys[i]);  // ascending already
    }
    std::vector<uint64_t> out;
    if (mark)
        std::set_union(c->hPersist.begin(), c->hPersist.end(), hit.begin(), hit.end(), std::back_inserter(out));
    else
        std::set_difference(c->hPersist.begin(), c->hPersist.end(), hit.begin(), hit.end(), std::back_inserter(out));
    c->hPersist.swap(out);
    if (!c->hPersist.empty()) {
        if (int rc = ensure(c, c->persistKeys, c->hPersist.size() * 8))
            return rc;
        HIPCK(hipMemcpyAsync(c->persistKeys.p, c->hPersist.data(), c->hPersist.size() * 8, hipMemcpyHostToDevice, c->stream));
        HIPCK(hipStreamSynchronize(c->stream));
    }
    return DEME_OK;
}

int deme_download_persistent_contacts(deme_ctx* c, uint32_t* idA, uint32_t* idB, uint8_t* type, size_t cap) {
    if (int rc = check_ready(c))
        return rc;
    const size_t n = c->hPersist.size();
    if (cap < n)
        return fail(c, DEME_ERR_INVALID, "buffer too small: need %zu", n);
    std::vector<uint64_t> keys(c->hPersist);
    if (c->permuted) {
        for (auto& k : keys)
            k = order_key_out(c, k);
        std::sort(keys.begin(), keys.end());
    }
    for (size_t i = 0; i < n; i++) {
        const uint64_t k = keys[i];
        const uint32_t cls = key_class(k);
        if (idA)
            idA[i] = key_a(k);
        if (idB)
            idB[i] = key_b(k);
        if (type)
            type[i] = cls == DEME_KEY_CLASS_SS   ? DEME_SPHERE_SPHERE_CONTACT
                      : cls == DEME_KEY_CLASS_SM ? DEME_SPHERE_MESH_CONTACT
                      : (c->hObjType[key_b(k)] == DEME_ANAL_OBJ_TYPE_PLANE ? DEME_SPHERE_PLANE_CONTACT : DEME_SPHERE_CYL_CONTACT);
    }
    return DEME_OK;
}

int deme_upload_persistent_contacts(deme_ctx* c, const uint32_t* idA, const uint32_t* idB, const uint8_t* type, size_t n) {
    if (int rc = check_ready(c))
        return rc;
    if (n && (!idA || !idB || !type))
        return fail(c, DEME_ERR_INVALID, "deme_upload_persistent_contacts: null input");
    if (n && c->hp.nContactWildcards == 0)
        return fail(c, DEME_ERR_INVALID, "persistent contacts cannot be marked with a history-less force model");
    std::vector<uint64_t> keys(n);
    for (size_t i = 0; i < n; i++) {
        const uint64_t cls = (type[i] == 1) ? DEME_KEY_CLASS_SS : (type[i] == 2) ? DEME_KEY_CLASS_SM : DEME_KEY_CLASS_SA;
        const uint32_t nB = (cls == DEME_KEY_CLASS_SS) ? c->dp.nSpheres : (cls == DEME_KEY_CLASS_SM) ? c->dp.nTri : c->dp.nAnal;
        if (idA[i] >= c->dp.nSpheres || idB[i] >= nB)
            return fail(c, DEME_ERR_INVALID, "deme_upload_persistent_contacts: pair %zu (%u, %u, type %u) is out of range", i, idA[i],
                        idB[i], (unsigned)type[i]);
        keys[i] = order_key_in(c, make_key(cls, idA[i], idB[i]));
    }
    std::sort(keys.begin(), keys.end());
    keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
    c->hPersist.swap(keys);
    if (!c->hPersist.empty()) {
        if (int rc = ensure(c, c->persistKeys, c->hPersist.size() * 8))
            return rc;
        HIPCK(hipMemcpyAsync(c->persistKeys.p, c->hPersist.data(), c->hPersist.size() * 8, hipMemcpyHostToDevice, c->stream));
        HIPCK(hipStreamSynchronize(c->stream));
    }
    return DEME_OK;
}

int deme_num_persistent_contacts(deme_ctx* c, size_t* n) {
    if (!c || !n)
        return DEME_ERR_INVALID;
    *n = c->hPersist.size();
    return DEME_OK;
}

int deme_change_family(deme_ctx* c, uint32_t from, uint32_t to) {
    if (int rc = check_ready(c))
        return rc;
    if (from > 255 || to > 255)
        return fail(c, DEME_ERR_INVALID, "family numbers must not be larger than 255");
    if (c->nOwners)
        hipLaunchKernelGGL(k_change_family, dim3(grid_for(c->nOwners)), dim3(256), 0, c->stream, c->owners.as<OwnerRec>(),
                           (uint32_t)c->nOwners, from, to);
    c->prescDirty = true;
    c->listStale = true;  // pairs may have become unmasked
    return DEME_OK;
}

int deme_set_family_material(deme_ctx* c, uint32_t family, uint32_t material, int kind) {
    if (int rc = check_ready(c))
        return rc;
    if (family > 255 || material >= c->nMat || (kind != 0 && kind != 1))
        return fail(c, DEME_ERR_INVALID, "deme_set_family_material: family %u, material %u of %u, kind %d", family, material, c->nMat, kind);
    if (kind == 0 && c->nSpheres)
        hipLaunchKernelGGL(k_family_material_spheres, dim3(grid_for(c->nSpheres)), dim3(256), 0, c->stream, c->nSpheres,
                           c->spheres.as<SphereRec>(), c->owners.as<OwnerRec>(), family, material);
    if (kind == 1 && c->nTri)
        hipLaunchKernelGGL(k_family_material_tris, dim3(grid_for(c->nTri)), dim3(256), 0, c->stream, c->nTri, c->tris.as<TriRec>(),
                           c->owners.as<OwnerRec>(), family, material);
    c->listStale = true;  // the per-contact gather records hold the materials
    return DEME_OK;
}

int deme_device_memory(deme_ctx* c, size_t* usedBytes, size_t* totalBytes) {
    if (!c)
        return DEME_ERR_INVALID;
    size_t fr = 0, tot = 0;
    HIPCK(hipSetDevice(c->device));
    HIPCK(hipMemGetInfo(&fr, &tot));
    if (usedBytes)
        *usedBytes = tot - fr;
    if (totalBytes)
        *totalBytes = tot;
    return DEME_OK;
}

int deme_jit_probe_ex(const char* src, const char* const* wildcardNames, uint32_t nWildcards, const char* const* ownerNames,
                      uint32_t nOwnerWc, const char* const* geoNames, uint32_t nGeoWc, const char* prerequisites, char* log, size_t logCap) {
    deme_jit::MaterialTables mt;
    mt.nMat = 2;
    mt.E = {1e8f, 1e9f}, mt.nu = {0.3f, 0.3f};
    mt.CoR = {0.5f, 0.6f, 0.6f, 0.7f}, mt.mu = {0.2f, 0.3f, 0.3f, 0.4f}, mt.Crr = {0.f, 0.f, 0.f, 0.f};
    std::vector<std::string> names, onames, gnames;
    for (uint32_t i = 0; i < nWildcards; i++)
        names.emplace_back(wildcardNames[i] ? wildcardNames[i] : "");
    for (uint32_t i = 0; i < nOwnerWc; i++)
        onames.emplace_back(ownerNames[i] ? ownerNames[i] : "");
    for (uint32_t i = 0; i < nGeoWc; i++)
        gnames.emplace_back(geoNames[i] ? geoNames[i] : "");
    std::string gen, err, clog;
    std::vector<char> code;
    int rc = deme_jit::generate_source(src ? src : "", names, prerequisites ? prerequisites : "", mt, gen, err, onames, gnames);
    if (!rc) {
        rc = deme_jit::compile(gen, code, clog);
        err = clog;
    }
    if (log && logCap) {
        snprintf(log, logCap, "%s", err.c_str());
    }
    return rc ? DEME_ERR_COMPILE : DEME_OK;
}

int deme_jit_probe(const char* src, const char* const* wildcardNames, uint32_t nWildcards, const char* prerequisites,
                   char* log, size_t logCap) {
    return deme_jit_probe_ex(src, wildcardNames, nWildcards, nullptr, 0, nullptr, 0, prerequisites, log, logCap);
}

int deme_set_timing(deme_ctx* c, int enable) {
    if (!c)
        return DEME_ERR_INVALID;
    c->timing = enable > 0 ? enable : 0;
    return DEME_OK;
}
int deme_kernel_time_reset(deme_ctx* c) {
    if (!c)
        return DEME_ERR_INVALID;
    hipSetDevice(c->device);  // (a process may hold contexts on several devices: deme_multi)
    hipStreamSynchronize(c->stream);
    drain_timers(c);
    for (auto& kv : c->timers) {
        kv.second.total_ms = 0;
        kv.second.launches = 0;
        kv.second.seen = 0;
    }
    return DEME_OK;
}
namespace {
struct FMax {
    __host__ __device__ float operator()(float a, float b) const { return a > b ? a : b; }
};
struct FMin {
    __host__ __device__ float operator()(float a, float b) const { return a < b ? a : b; }
};
// fills c->stage with the per-element quantity; returns the element count through n.  region >= 0: elements outside the
// compiled region (deme_compile_region) get `identity`
int inspect_fill(deme_ctx* c, uint32_t q, float identity, size_t& n, int region = -1) {
    if (q > DEME_INSPECT_CLUMP_VOLUME)
        return fail(c, DEME_ERR_INVALID, "unknown inspection quantity %u", q);
    if (region >= (int)c->regionFn.size())
        return fail(c, DEME_ERR_INVALID, "unknown inspection region %d", region);
    if (q == DEME_INSPECT_CLUMP_VOLUME && !c->haveVolumes)
        return fail(c, DEME_ERR_INVALID, "clump_volume needs the templates' volumes (deme_upload_volumes)");
    const bool perSphere = q <= DEME_INSPECT_CLUMP_MAX_ABSV;
    n = perSphere ? c->nSpheres : c->nOwners;
    if (int rc = ensure(c, c->stage, std::max<size_t>(n, 1) * 4 + 16))
        return rc;
    if (n == 0)
        return DEME_OK;
    if (perSphere)
        hipLaunchKernelGGL(k_inspect_sphere, dim3(grid_for(n)), dim3(256), 0, c->stream, c->dp, c->owners.as<OwnerRec>(),
                           c->spheres.as<SphereRec>(), q, identity, c->stage.as<float>());
    else
        hipLaunchKernelGGL(k_inspect_owner, dim3(grid_for(n)), dim3(256), 0, c->stream, c->dp, c->owners.as<OwnerRec>(),
                           (uint32_t)c->nOwnerClumps, q, identity, c->volumes.as<float>(), c->stage.as<float>());
    if (region >= 0) {
        DevParams dp = c->dp;
        const OwnerRec* owners = c->owners.as<OwnerRec>();
        const SphereRec* spheres = c->spheres.as<SphereRec>();
        uint32_t nn = (uint32_t)n, ps = perSphere ? 1u : 0u;
        float* values = c->stage.as<float>();
        void* args[] = {&dp, &owners, &spheres, &nn, &ps, &identity, &values};
        HIPCK(hipModuleLaunchKernel(c->regionFn[region], grid_for(n), 1, 1, 256, 1, 1, 0, c->stream, args, nullptr));
    }
    return DEME_OK;
}
}  // namespace

int deme_inspect_region(deme_ctx* c, uint32_t q, int region, float* out) {
    if (int rc = check_ready(c))
        return rc;
    if (!out || q == DEME_INSPECT_ABSV)
        return fail(c, DEME_ERR_INVALID, "deme_inspect: quantity %u has no reduction (use deme_inspect_values)", q);
    const bool isMax = q == DEME_INSPECT_CLUMP_MAX_Z || q == DEME_INSPECT_CLUMP_MAX_ABSV || q == DEME_INSPECT_MAX_ABSV;
    const bool isMin = q == DEME_INSPECT_CLUMP_MIN_Z;
    const float identity = isMax ? -3.402823466e38f : isMin ? 3.402823466e38f : 0.f;
    size_t n = 0;
    if (int rc = inspect_fill(c, q, identity, n, region))
        return rc;
    float* in = c->stage.as<float>();
    float* res = in + n;  // one spare slot behind the values
    size_t need = 0;
    if (isMax) {
        HIPCK(rocprim::reduce(nullptr, need, in, res, identity, n, FMax(), c->stream));
    } else if (isMin) {
        HIPCK(rocprim::reduce(nullptr, need, in, res, identity, n, FMin(), c->stream));
    } else {
        HIPCK(rocprim::reduce(nullptr, need, in, res, identity, n, rocprim::plus<float>(), c->stream));
    }
    if (int rc = ensure(c, c->sortTmp, need))
        return rc;
    need = c->sortTmp.bytes;
    if (isMax) {
        HIPCK(rocprim::reduce(c->sortTmp.p, need, in, res, identity, n, FMax(), c->stream));
    } else if (isMin) {
        HIPCK(rocprim::reduce(c->sortTmp.p, need, in, res, identity, n, FMin(), c->stream));
    } else {
        HIPCK(rocprim::reduce(c->sortTmp.p, need, in, res, identity, n, rocprim::plus<float>(), c->stream));
    }
    HIPCK(hipMemcpyAsync(out, res, 4, hipMemcpyDeviceToHost, c->stream));
    HIPCK(hipStreamSynchronize(c->stream));
    return DEME_OK;
}

int deme_inspect(deme_ctx* c, uint32_t q, float* out) { return deme_inspect_region(c, q, -1, out); }

int deme_compile_region(deme_ctx* c, const char* code, int* regionId) {
    if (!c || !regionId)
        return DEME_ERR_INVALID;
    hipSetDevice(c->device);  // (a process may hold contexts on several devices: deme_multi)
    const std::string r = code ? code : "";
    auto word = [&](const std::string& w) {
        for (size_t pos = r.find(w); pos != std::string::npos; pos = r.find(w, pos + 1)) {
            const bool l = pos == 0 || !(isalnum((unsigned char)r[pos - 1]) || r[pos - 1] == '_');
            const size_t e = pos + w.size();
            const bool rr = e >= r.size() || !(isalnum((unsigned char)r[e]) || r[e] == '_');
            if (l && rr)
                return true;
        }
        return false;
    };
    if (!(word("X") || word("Y") || word("Z")) || !word("return"))  // DEM/AuxClasses.cpp:209-219
        return fail(c, DEME_ERR_INVALID,
                    "an inspection region must return a bool that is a result of logical operations involving X, Y and Z");
    std::string gen;
    deme_jit::generate_region_source(r, gen);
    const size_t key = std::hash<std::string>{}(gen);
    auto it = c->jitCache.find(key);
    if (it == c->jitCache.end()) {
        std::vector<char> bin;
        std::string log;
        if (deme_jit::compile(gen, bin, log))
            return fail(c, DEME_ERR_COMPILE, "inspection region failed to compile:\n%.900s", log.c_str());
        it = c->jitCache.emplace(key, std::move(bin)).first;
    }
    hipModule_t mod = nullptr;
    hipFunction_t fn = nullptr;
    HIPCK(hipModuleLoadData(&mod, it->second.data()));
    HIPCK(hipModuleGetFunction(&fn, mod, "deme_region_filter"));
    c->regionMod.push_back(mod);
    c->regionFn.push_back(fn);
    *regionId = (int)c->regionFn.size() - 1;
    return DEME_OK;
}

int deme_upload_volumes(deme_ctx* c, const float* volumes, size_t n) {
    if (int rc = check_ready(c))
        return rc;
    if (!volumes || n != c->nMassProps)
        return fail(c, DEME_ERR_INVALID, "deme_upload_volumes: expected %u values (one per mass-property entry)", c->nMassProps);
    if (int rc = ensure(c, c->volumes, std::max<size_t>(n, 1) * 4))
        return rc;
    HIPCK(hipMemcpyAsync(c->volumes.p, volumes, n * 4, hipMemcpyHostToDevice, c->stream));
    HIPCK(hipStreamSynchronize(c->stream));
    c->haveVolumes = true;
    return DEME_OK;
}

int deme_inspect_values(deme_ctx* c, uint32_t q, float* out, size_t cap) {
    if (int rc = check_ready(c))
        return rc;
    size_t n = 0;
    if (int rc = inspect_fill(c, q, 0.f, n))
        return rc;
    if (!out || cap < n)
        return fail(c, DEME_ERR_INVALID, "buffer too small: need %zu", n);
    if (n)
        HIPCK(hipMemcpyAsync(out, c->stage.p, n * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCK(hipStreamSynchronize(c->stream));
    if (c->permuted && n) {  // per sphere or per owner: back into the caller's numbering
        const std::vector<uint32_t>& m = n == c->nSpheres && n != c->nOwners ? c->hS2E : (q <= DEME_INSPECT_CLUMP_MAX_ABSV ? c->hS2E : c->hO2E);
        std::vector<float> tmp(out, out + n);
        for (size_t k = 0; k < n; k++)
            out[m[k]] = tmp[k];
    }
    return DEME_OK;
}

int deme_kernel_time_ms(deme_ctx* c, const char* name, double* avg_ms, uint64_t* launches) {
    if (!c || !name)
        return DEME_ERR_INVALID;
    hipSetDevice(c->device);  // (a process may hold contexts on several devices: deme_multi)
    hipStreamSynchronize(c->stream);
    drain_timers(c);
    auto it = c->timers.find(name);
    const double tot = it == c->timers.end() ? 0.0 : it->second.total_ms;
    const uint64_t n = it == c->timers.end() ? 0 : it->second.launches;
    if (avg_ms)
        *avg_ms = n ? tot / (double)n : 0.0;
    if (launches)
        *launches = n;
    return DEME_OK;
}

int deme_halo_pack(deme_ctx* c, const uint32_t* d_ids, uint32_t n, void* d_buf) {
    if (int rc = check_ready(c))
        return rc;
    if (n)
        hipLaunchKernelGGL(k_halo_pack, dim3(grid_for(n)), dim3(256), 0, c->stream, n, d_ids, c->owners.as<OwnerRec>(),
                           (GhostRec*)d_buf);
    return DEME_OK;
}
int deme_halo_unpack(deme_ctx* c, const uint32_t* d_ids, uint32_t n, const void* d_buf) {
    if (int rc = check_ready(c))
        return rc;
    if (n)
        hipLaunchKernelGGL(k_halo_unpack, dim3(grid_for(n)), dim3(256), 0, c->stream, n, d_ids, c->owners.as<OwnerRec>(),
                           (const GhostRec*)d_buf);
    return DEME_OK;
}

}  // extern "C"

// (templates inside: after the extern "C" block; its entry points take C linkage from their declarations in deme_hip.h)
#include "deme_decomp.inc"
