// deme_tile_p.h -- the owner-tile force pass as PERSISTENT workgroups (round 6; opt-in: DEME_TILE_PERSIST=1).
//
// Same tiles, same staging, same rounds, same pulls, same outputs and the same arithmetic as k_tile_forces (deme_tile.h; physics:
// kernel/DEMCalcForceKernels.cu:44-267 with FullHertzianForceModel.cu / FrictionlessHertzianForceModel.cu) -- bit for bit
// (tools/persist_compare.py, tests/test_tile_persistent.py).  What changed is who runs a tile and when its loads go out.  The phase
// stamps of round 6 (profiles/r06/tile_phase_stamps_plain.txt) showed a tile living 10.3 us of which 4.1 us pass before its first
// round starts -- two memory latencies in a row (the ids of the foreign owners, then their records) -- and only ~925 of the 1024
// workgroup slots of the chip occupied at any time: a slot stays empty for ~1 us between the end of one workgroup and the start of
// the next.  Here a workgroup stays on its slot and takes tile after tile, always knowing the NEXT tile's number a tile ahead:
// behind the staging barrier of tile k the scalars of tile k + 1 (contact range, counts, origin: SGPRs) and the ids of its foreign
// owners (two VGPRs) go out, so that at the start of tile k + 1 every load it needs -- local records, foreign records, streams,
// lists -- goes out at once: the chain is ONE latency deep.  The small tables are copied to LDS once per workgroup.
//
// What it measured (profiles/r06/force_pass_ceiling.md, tile_phase_stamps_persistent.txt): start -> staged 4.1 -> 2.5 us, every slot
// full -- and the rounds 2.7 -> 3.1 us, because the pass is bound by VALU issue (60 us of it in an 88 us kernel), not by the waiting
// this form removes: 99 us against 89 us.  Kept as an option and as the measurement; k_tile_forces is the default.
#pragma once
#include "deme_tile.h"

#pragma clang fp contract(fast)

namespace deme_dev {

struct RawOwner {  // a 64-byte owner record as it lies in memory: decoded where it is used (staging), not where it is loaded --
    uint4 a, b, c, d;  // load_owner's field extraction right behind the load made the wavefront wait for the record before
};                     // issuing anything else
__device__ inline RawOwner load_owner_raw(const OwnerRec* owners, uint32_t o) {
    const uint4* q = reinterpret_cast<const uint4*>(owners + o);
    RawOwner r;
    r.a = q[0], r.b = q[1], r.c = q[2], r.d = q[3];
    return r;
}
__device__ inline OwnerRec owner_of(const RawOwner& r) {
    OwnerRec o;
    __builtin_memcpy(&o, &r, sizeof(OwnerRec));
    return o;
}

// the scalars of one tile (uniform over the workgroup: scalar loads, SGPRs)
struct TileScal {
    uint32_t nH, c0, c1, nL, skip;
    int64_t u0x, u0y, u0z;
};
// (read through the CONSTANT address space: none of these arrays is written while the force pass runs, and a load the compiler
// cannot prove unclobbered -- here: any load behind the first store of the persistent loop -- would become a vector load)
template <typename T>
__device__ inline T const_load(const T* ptr, size_t i) {
    typedef const T __attribute__((address_space(4))) * cptr;
    return ((cptr)(uintptr_t)ptr)[i];
}
__device__ inline TileScal tile_scalars(const TileArgs& a, const uint32_t t) {
    TileScal S;
    const uint32_t o0 = t * DEME_TILE_NB;
    const uint32_t nLoc = min((uint32_t)DEME_TILE_NB, a.nOwners - o0);
    S.nH = const_load(a.hCount, t);
    S.c0 = const_load(a.aStart, o0), S.c1 = const_load(a.aStart, o0 + nLoc);
    S.nL = const_load(a.lCount, t);
    S.u0x = const_load(a.org, 3 * (size_t)t), S.u0y = const_load(a.org, 3 * (size_t)t + 1), S.u0z = const_load(a.org, 3 * (size_t)t + 2);
    uint32_t skip = const_load(a.tileBig, t);
    if (a.tileMode)
        skip |= (const_load(a.tileMode, t) & (1u << a.pass)) ? 0u : 1u;
    S.skip = skip;
    return S;
}

#ifndef DEME_TILE_P_PARTS
#define DEME_TILE_P_PARTS 32u  // partitions of a full launch (one tile counter each)
#endif
#define DEME_TILE_P_CTR_WORDS ((DEME_TILE_P_PARTS + 1u) * 32u)  // counters of one launch: 128 bytes apart, then the workgroups-through word
#ifndef DEME_TILE_P_OCC
#define DEME_TILE_P_OCC 1
#endif

// a.tileCtr[32 k]: tiles of partition k handed out beyond every workgroup's first two; a.tileCtr[32 parts]: workgroups that are
// through.  All zero at launch; the last workgroup to leave sets them back.
template <int MODEL, bool MESH, bool REC = false>
__global__ __launch_bounds__(DEME_TILE_T, DEME_TILE_P_OCC) void k_tile_forces_p(const DevParams p, const TileArgs a) {
    extern __shared__ uint4 tileLds[];
    const uint32_t RSZ = a.rs16;
#define DEME_TILE_P_LDS(base)                                                                          \
    uint4* const sOwn = (base);                                                                        \
    float4* const recA4 = reinterpret_cast<float4*>(sOwn + (DEME_TILE_NB + a.hCap) * RSZ);             \
    float4* const recT = recA4 + DEME_TILE_RSLOTS;                                                     \
    float2* const recA2 = reinterpret_cast<float2*>(recT + DEME_TILE_RSLOTS);                          \
    uint16_t* const sALo = reinterpret_cast<uint16_t*>(recA2 + DEME_TILE_RSLOTS);                      \
    uint16_t* const sLLo = sALo + (DEME_TILE_NB + 1);                                                  \
    uint16_t* const sLPos = sALo + DEME_TILE_BOUNDS_BYTES / 2u;
    uint16_t* sLPos0;
    float4 *recA40, *recT0;
    float2* recA20;
    {
        DEME_TILE_P_LDS(tileLds)
        sLPos0 = sLPos, recA40 = recA4, recT0 = recT, recA20 = recA2;
        (void)sLLo;
    }
    const uint32_t tid = threadIdx.x;
    const uint32_t nTiles = a.nTiles;
    // ---- once per workgroup: the small tables and the zero slot of the contribution arrays
    TileTables T;
    uint4* const tabBase = reinterpret_cast<uint4*>(reinterpret_cast<char*>(sLPos0) + ((a.lCap * 2u + 15u) & ~15u));
    const uint32_t nTab16 = a.nComp + p.nMat * p.nMat * 2u + a.nAnal * 4u;
    uint32_t* sNext;  // two words behind the tables: the number of the tile after the next, handed from thread 0 to everybody
    {
        float4* sComp = reinterpret_cast<float4*>(tabBase);
        MatPair* sMat = reinterpret_cast<MatPair*>(sComp + a.nComp);
        AnalObj* sAnal = reinterpret_cast<AnalObj*>(sMat + p.nMat * p.nMat);
        float* sMass = reinterpret_cast<float*>(sAnal + a.nAnal);
        float* sFam = sMass + ((a.nMass + 3u) & ~3u);
        T.comp = sComp, T.mat = sMat, T.anal = sAnal, T.mass = sMass, T.fam = sFam;
        sNext = reinterpret_cast<uint32_t*>(sFam + (p.familyTrivial ? 0u : 256u));
        if (tid < nTab16) {
            const uint32_t k = tid;
            tabBase[tid] = k < a.nComp ? reinterpret_cast<const uint4*>(p.comp)[k]
                           : (k < a.nComp + p.nMat * p.nMat * 2u ? reinterpret_cast<const uint4*>(p.matPair)[k - a.nComp]
                                                                  : reinterpret_cast<const uint4*>(p.anal)[k - a.nComp - p.nMat * p.nMat * 2u]);
        }
        if (tid < a.nMass)
            sMass[tid] = p.massProps[tid].x;
        if (!p.familyTrivial)
            sFam[tid & 255u] = p.familyExtra[tid & 255u];
        if (tid == DEME_TILE_T - 1u)
            recA40[DEME_TILE_T] = make_float4(0, 0, 0, 0), recT0[DEME_TILE_T] = make_float4(0, 0, 0, 0), recA20[DEME_TILE_T] = make_float2(0, 0);
    }
    const float4* const wc4 = reinterpret_cast<const float4*>(a.wc);
    constexpr int NWU = MODEL == 2 ? DEME_JIT_NW : 1;

    // ---- which tiles this workgroup takes.  The tiles are dealt to a.ctrParts partitions (tile t belongs to partition t % parts),
    // the workgroups likewise (workgroup b to partition b % parts; the host launches a multiple of `parts` workgroups, or parts = 1):
    // one counter per partition, 128 bytes apart -- a single counter for the whole chip answers one atomic per ~15 ns, 7 800 tiles
    // were 117 us of counter alone (measured: profiles/r06/persistent_first_attempt.txt).  A workgroup's first two tiles are
    // fixed (its own rank in the partition, then that plus the partition's workgroups); from the third on the number comes from
    // the partition's counter, asked for behind the staging barrier of one tile and handed to the workgroup at the staging
    // barrier of the next -- a whole tile later, at a point where the wavefront has just waited for its records anyway.
    const uint32_t parts = a.ctrParts, part = blockIdx.x % parts, wgPerPart = gridDim.x / parts;
    uint32_t* const ctr = a.tileCtr + part * 32u;
#ifndef DEME_TILE_P_CONTIG
#define DEME_TILE_P_CONTIG 0  // 1: a partition is a contiguous range of tiles (neighbouring tiles on one XCD: their shared owner records in one L2)
#endif
#if DEME_TILE_P_CONTIG
    const uint32_t chunk = (nTiles + parts - 1u) / parts;
    auto tile_of = [&](uint32_t idx) { return idx < chunk ? part * chunk + idx : 0xFFFFFFFFu; };
#else
    auto tile_of = [&](uint32_t idx) { return part + parts * idx; };
#endif
    uint32_t t = tile_of(blockIdx.x / parts);  // (the host launches at most nTiles workgroups)
    TileScal S = tile_scalars(a, t);
    uint32_t id0 = 0u, id1 = 0u;
    {
        const uint32_t nLoc = min((uint32_t)DEME_TILE_NB, a.nOwners - t * DEME_TILE_NB);
        const uint32_t* hl = a.hList + (size_t)t * DEME_TILE_HMAX;
        const uint32_t h0 = tid - nLoc, h1 = tid + DEME_TILE_T - nLoc;
        if (tid >= nLoc && h0 < DEME_TILE_HMAX)
            id0 = hl[h0];
        if (h1 < DEME_TILE_HMAX)
            id1 = hl[h1];
    }
    asm volatile("" : "+v"(id0), "+v"(id1));  // (waited for here, once: see the note in the epilogue)
    __syncthreads();  // (the tables are in LDS)
    uint32_t pend = tile_of(blockIdx.x / parts + wgPerPart);  // thread 0: the tile after the current one, to be told at the staging barrier
    for (uint32_t it = 0;; it++) {
        bool haveN = false;
        TileScal SN = S;
        uint32_t idn0 = 0u, idn1 = 0u, tN = 0xFFFFFFFFu;
        uint32_t fetched = 0u;  // (thread 0 only)
        // What tile tN will need first goes out behind the staging barrier of tile t: its scalars (SGPRs), the ids of its foreign
        // owners (two VGPRs) -- and the question for the number of the tile after it.
        auto hand_over_and_prefetch = [&]() __attribute__((always_inline)) {
            if (tid == 0)
                sNext[it & 1u] = pend;
            __syncthreads();
            tN = __builtin_amdgcn_readfirstlane(sNext[it & 1u]);
            haveN = tN < nTiles;
            if (haveN) {
                SN = tile_scalars(a, tN);
                const uint32_t nLocN = min((uint32_t)DEME_TILE_NB, a.nOwners - tN * DEME_TILE_NB);
                const uint32_t* hlN = a.hList + (size_t)tN * DEME_TILE_HMAX;
                const uint32_t h0N = tid - nLocN, h1N = tid + DEME_TILE_T - nLocN;
                if (tid >= nLocN && h0N < DEME_TILE_HMAX)
                    idn0 = hlN[h0N];
                if (h1N < DEME_TILE_HMAX)
                    idn1 = hlN[h1N];
                if (tid == 0)
                    fetched = atomicAdd(ctr, 1u);
            }
        };
        if (!S.skip) {
            uint32_t tl = tid;  // (the thread's number, opaque per tile: what is derived from it -- a dozen stream and list addresses -- is
            asm volatile("" : "+v"(tl));  // formed per tile, not once per workgroup and kept in registers through every round)
            uint32_t zOff = 0u;           // ... and likewise the addresses of the LDS areas
            asm volatile("" : "+s"(zOff));
            DEME_TILE_P_LDS(tileLds + zOff)
            // the pulling side of this thread: threads 0 .. NB - 1 take the A runs, NB .. 2 NB - 1 the local-B lists
            const uint32_t po = tl % DEME_TILE_NB;
            const bool sideA = tl < DEME_TILE_NB, sideB = !sideA && tl < 2 * DEME_TILE_NB;
            const uint32_t o0 = t * DEME_TILE_NB;
            const uint32_t nLoc = min((uint32_t)DEME_TILE_NB, a.nOwners - o0);
            const uint32_t nH = S.nH, c0 = S.c0, c1 = S.c1, nL = S.nL;
            const int64_t u0x = S.u0x, u0y = S.u0y, u0z = S.u0z;
            TILE_STAMP(0);
            // ---- every load of the tile at once
            const uint32_t h0 = tl - nLoc, h1 = tl + DEME_TILE_T - nLoc;  // my foreign slots (meaningful when < nH)
            const uint32_t* const hl = a.hList + (size_t)t * DEME_TILE_HMAX;
            // (every thread loads a record -- a thread without one the tile's first: a value defined on one side of a branch only would
            // keep its sixteen registers through the whole loop over the tiles)
            const RawOwner raw0 = load_owner_raw(a.owners, tl < nLoc ? o0 + tl : (h0 < nH ? id0 : o0));
            uint2 inf[DEME_TILE_DEPTH];
            float4 hist[DEME_TILE_DEPTH];
            float uwv[DEME_TILE_DEPTH][NWU];
            uint32_t rbase[DEME_TILE_DEPTH];
#pragma unroll
            for (int d = 0; d < DEME_TILE_DEPTH; d++) {
                const uint32_t cd = c0 + tl + d * DEME_TILE_T;
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
                    rbase[d] = a.rankC[cd - (tl & 63u)];
                }
            }
            uint32_t bA = 0, bL = 0;
            if (tl <= DEME_TILE_NB) {
                const uint32_t o = min(tl, nLoc);
                bA = a.aStart[o0 + o], bL = (a.lOff + (size_t)t * (DEME_TILE_NB + 1))[o];
            }
            uint32_t lp[DEME_TILE_LREG];
#pragma unroll
            for (int k = 0; k < DEME_TILE_LREG; k++) {
                const uint32_t i = tl + k * DEME_TILE_T;
                lp[k] = (i < nL) ? (uint32_t)a.lPos[c0 + i] : 0u;
            }
            // ---- staging: the tile's owners, its halo, the owners' run bounds and local-B lists
            {
                if (tl < nLoc + nH) {
                    const OwnerRec r = owner_of(raw0);
                    const uint32_t slot = tl < nLoc ? tl : DEME_TILE_NB + h0;
                    tile_stage<MODEL>(p, T.mass[r.inertiaOff], r, u0x, u0y, u0z, sOwn + slot * RSZ, (slot >> 3) & a.swz);
                }
                if (nLoc + nH > (uint32_t)DEME_TILE_T) {  // (a halo beyond one record per thread: rare, and not worth 16 registers held
                    if (h1 < nH) {                          // through the prologue of every tile -- loaded here, behind the others)
                        const OwnerRec r = owner_of(load_owner_raw(a.owners, id1));
                        tile_stage<MODEL>(p, T.mass[r.inertiaOff], r, u0x, u0y, u0z, sOwn + (DEME_TILE_NB + h1) * RSZ, ((DEME_TILE_NB + h1) >> 3) & a.swz);
                    }
                }
                if (tl <= DEME_TILE_NB)
                    sALo[tl] = (uint16_t)(bA - c0), sLLo[tl] = (uint16_t)bL;
#pragma unroll
                for (int k = 0; k < DEME_TILE_LREG; k++)
                    if (tl + k * DEME_TILE_T < nL)
                        sLPos[tl + k * DEME_TILE_T] = (uint16_t)lp[k];
            }
            hand_over_and_prefetch();  // (with the staging barrier)
            TILE_STAMP(2);
            uint32_t plo = sideB ? sLLo[po] : sALo[po];
            const uint32_t phi = (sideA || sideB) ? (sideB ? sLLo[po + 1] : sALo[po + 1]) : plo;
            // the six sums of this thread's owner and side as three register pairs (the pulls add with v_pk_add_f32):
            //   A side: (F.x F.y) (F.z tA.x) (tA.y tA.z);  B side: (-F.x -F.y) (-F.z tB.x) (tB.y tB.z)
            v2f s01 = {0.f, 0.f}, s23 = {0.f, 0.f}, s45 = {0.f, 0.f};
            const uint32_t nCt = c1 - c0;
            for (uint32_t rlo = 0; rlo < nCt; rlo += DEME_TILE_T) {  // stage 0 is the current round; the stages are rotated after it
                const uint32_t c = c0 + rlo + tl;
                bool crossing = false;
                float4 x4, x2;  // (read only where `crossing` was set)
                if (c < c1) {
                    const uint2 ci = inf[0];
                    float4 h = hist[0];
                    const uint32_t slotA = ci.x & 1023u, slotB = (ci.x >> 10) & 1023u;
                    const TileOwner A = tile_read<MODEL>(sOwn, slotA * RSZ, (slotA >> 3) & a.swz), B = tile_read<MODEL>(sOwn, slotB * RSZ, (slotB >> 3) & a.swz);
                    f3 force, tA, tB;
                    if (MESH && ((ci.x >> 20) & 3u) == DEME_KEY_CLASS_SM) {  // (rare, and only in tiles along the mesh: the loads sit behind a branch)
                        const float4 a4 = a.conA4[c], b4 = a.conB4[c];
                        const float2 a2 = a.conA2[c], b2 = a.conB2[c];
                        force = mk3(a4.x, a4.y, a4.z), tA = mk3(a4.w, a2.x, a2.y), tB = mk3(b4.w, b2.x, b2.y);
                    } else {
                        if (MODEL == 2) {
                            float uw[NWU];
#pragma unroll
                            for (int k = 0; k < NWU; k++)
                                uw[k] = uwv[0][k];
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
#else
                        ki_sink(h.x + h.y + h.z + h.w);
#endif
                    }
                    recA4[tl] = make_float4(force.x, force.y, force.z, tA.x);
                    recA2[tl] = make_float2(tA.y, tA.z);
                    if (slotB < DEME_TILE_NB) {
                        recT[tl] = make_float4(tB.y, tB.z, tB.x, 0.f);
                    } else if (ci.x & (1u << 22)) {
                        crossing = true;
                        x4 = make_float4(-force.x, -force.y, -force.z, tB.x), x2 = make_float4(tB.y, tB.z, 0.f, 0.f);
                    }
                }
                {   // the wavefront's crossing contacts write consecutive records
                    const uint64_t m = __ballot(crossing);
                    if (crossing) {
                        const uint32_t k = rbase[0] + (uint32_t)__popcll(m & ((1ull << (tl & 63u)) - 1ull));
#if DEME_TILE_KI & 32
                        ki_sink(x4.x + x4.y + x4.z + x4.w + x2.x + x2.y + (float)k);
#elif DEME_REC24
                        float2* const r24 = reinterpret_cast<float2*>(a.rec32) + 3 * (size_t)k;  // (-F.x -F.y) (-F.z tB.x) (tB.y tB.z)
                        stream_store(r24, make_float2(x4.x, x4.y));
                        stream_store(r24 + 1, make_float2(x4.z, x4.w));
                        stream_store(r24 + 2, make_float2(x2.x, x2.y));
#else
                        stream_store(a.rec32 + 2 * (size_t)k, x4);
                        stream_store(a.rec32 + 2 * (size_t)k + 1, x2);
#endif
                    }
                }
#pragma unroll
                for (int q = 0; q + 1 < DEME_TILE_DEPTH; q++) {
                    inf[q] = inf[q + 1], hist[q] = hist[q + 1], rbase[q] = rbase[q + 1];
#pragma unroll
                    for (int k = 0; k < NWU; k++)
                        uwv[q][k] = uwv[q + 1][k];
                }
                {   // stage DEPTH - 1 takes the round DEPTH rounds ahead
                    constexpr int d = DEME_TILE_DEPTH - 1;
                    const uint32_t cd = c + DEME_TILE_DEPTH * DEME_TILE_T;
                    if (cd < c1) {
                        inf[d] = stream_load(a.tInfo + cd);
                        if (MODEL == 0)
                            hist[d] = stream_load(wc4 + cd);
                        if (MODEL == 2 && DEME_JIT_HAS_WC) {
#pragma unroll
                            for (int k = 0; k < NWU; k++)
                                uwv[d][k] = stream_load(a.wc + (size_t)cd * NWU + k);
                        }
                        rbase[d] = a.rankC[cd - (tl & 63u)];
                    }
                }
                __syncthreads();
                const uint32_t rhi = rlo + DEME_TILE_T;
                if (sideA) {  // my A run's part of this round: positions [plo, min(phi, rhi)); a missing entry reads the zero slot
                    const uint32_t e = min(phi, rhi);
                    while (plo < e) {
                        float4 v4[DEME_TILE_PULLW];
                        float2 v2[DEME_TILE_PULLW];
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
                        plo += used;
                        if (used < (uint32_t)DEME_TILE_PULLW)
                            break;
                    }
                }
                __syncthreads();
                TILE_STAMP(min(3u + rlo / DEME_TILE_T, 10u));
            }
            // A-side sum + B-side sum, through LDS (the contribution arrays are free now); the tile after the next for everybody
            if (sideB) {
                recA4[po] = make_float4(s01.x, s01.y, s23.x, s23.y);
                recA2[po] = make_float2(s45.x, s45.y);
            }
            // (the next tile's ids went out behind the staging barrier, rounds ago: telling the compiler HERE that they have arrived --
            // it waits for whatever is outstanding, the last round's stores, which are about through -- spares the wait at the next
            // tile's start, where the tile sums stored below would have to drain first)
            asm volatile("" : "+v"(idn0), "+v"(idn1));
            __syncthreads();
            if (sideA && po < nLoc) {
                const float4 b4 = recA4[po];
                const float2 b2 = recA2[po];
#if DEME_TILE_KI & 64
                ki_sink(s01.x + b4.x + s01.y + b4.y + s23.x + b4.z + s23.y + b4.w + s45.x + b2.x + s45.y + b2.y);
#elif DEME_REC24
                float2* const t24 = reinterpret_cast<float2*>(a.tSum) + 3 * (size_t)(o0 + po);  // (F.x F.y) (F.z t.x) (t.y t.z)
                t24[0] = make_float2(s01.x + b4.x, s01.y + b4.y);
                t24[1] = make_float2(s23.x + b4.z, s23.y + b4.w);
                t24[2] = make_float2(s45.x + b2.x, s45.y + b2.y);
#else
                a.tSum[2 * (size_t)(o0 + po)] = make_float4(s01.x + b4.x, s01.y + b4.y, s23.x + b4.z, 0.f);
                a.tSum[2 * (size_t)(o0 + po) + 1] = make_float4(s23.y + b4.w, s45.x + b2.x, s45.y + b2.y, 0.f);
#endif
            }
#if DEME_TILE_STAMPS
            TILE_STAMP(11);
            if (a.stamps && tl == 0) {
                uint32_t hw, xcc;
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
                a.stamps[(size_t)t * 16u + 12u] = (unsigned long long)hw | ((unsigned long long)xcc << 32);
                a.stamps[(size_t)t * 16u + 13u] = nCt;
                a.stamps[(size_t)t * 16u + 14u] = nH;
            }
#endif
        } else {  // (a tile of the other pass, or one that k_tile_forces_big takes: nothing but the hand-over)
            hand_over_and_prefetch();
            asm volatile("" : "+v"(idn0), "+v"(idn1));
        }
        if (!haveN)
            break;
        t = tN;
        S = SN, id0 = idn0, id1 = idn1;
        pend = tile_of(2u * wgPerPart + fetched);  // (thread 0; read at the next staging barrier)
    }
    // ---- the last workgroup to leave sets the counters back for the next launch
    if (tid == 0) {
        uint32_t* const doneCtr = a.tileCtr + parts * 32u;
        const uint32_t done = atomicAdd(doneCtr, 1u);
        if (done == gridDim.x - 1u) {
            for (uint32_t k = 0; k < parts; k++)
                atomicExch(a.tileCtr + k * 32u, 0u);
            atomicExch(doneCtr, 0u);
        }
    }
}

}  // namespace deme_dev
