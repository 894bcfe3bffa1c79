// deme_tile_step.h -- the whole time step in ONE kernel: k_tile_step<MODEL> (fast arithmetic mode, built-in models, scenes without
// ghosts / meshes / prescriptions; round 4).
//
// k_tile_forces leaves a 32-byte record for every contact whose B owner lives in another tile (27 % of the contacts of a packed bed
// in the engine's box-shaped tiles) and a 32-byte sum per owner, and k_integrate reads the owners again, gathers those records
// and writes the owners back: 42 us of the step's 135 for 242 MB of traffic that exists only because a tile cannot finish its own
// owners.  Here every tile is CLOSED: it evaluates its own contacts (A owner in the tile) AND the contacts that hold one of its
// owners as B from another tile -- those are evaluated twice, once by each tile, each keeping only its own owners' side --, so the
// sums of its 128 owners are complete when its rounds are over and the owner thread integrates in place of writing a sum:
// reference semantics of DEMCalcForceKernels.cu:44-267 + DEMIntegrationKernels.cu:100-236 in one pass.  No crossing records, no
// per-owner sums, no second kernel.
//   * Two evaluations of one contact must agree bit for bit (Newton's third law, and the history: only A's tile writes it).  They do:
//     both tiles stage the two owners from the same records, positions are taken relative to ONE origin (the world's, in fp64:
//     1e-16 m at a metre) instead of the tile's, and tile_contact is a pure function of its inputs.
//   * Owners are double-buffered (a tile writes its owners' new records while its neighbours may still be staging the old ones),
//     and so is the history (B's tile reads the value of the step's start while A's tile writes the new one): the context swaps
//     both after every step.  The buffers of the step before stay valid until the next step -- a state download that wants a / alpha
//     replays the step's force evaluation on them (`dry`: nothing stored but the accelerations).
//   * The incoming contacts of a tile are the (B owner, record) pairs the per-detection sort leaves in owner order already: owner
//     o's incoming contacts are a contiguous run of the tile's stream behind its own contacts, pulled like an A run.  Contacts whose
//     B owner is fixed (walls) are not listed: nobody integrates their sum.
#pragma once
#include "deme_tile.h"

#ifndef DEME_JIT

#pragma clang fp contract(fast)

namespace deme_dev {

struct StepArgs {
    TileArgs t;                 // what the tile pass takes (owners = the records of the step's start, wc = the history of the step's start)
    OwnerRec* ownersNext;       // ... and where the step's results go
    float* wcNext;
    const uint2* tInfoIn;       // per incoming contact: the tile pass's 8-byte record in the frame of B's tile
    const uint32_t* inContact;  // ... and its index in the list (its history is read from there)
    const uint32_t* inStart;    // per owner (+1): first incoming contact
    AccRec* acc;                // dry runs: a / alpha of every owner
    const AccRec* nextAcc;      // accelerations a script added for this step (deme_add_owner_acc), or null
    uint32_t dry;
};

template <int MODEL>
__global__ __launch_bounds__(DEME_TILE_T, DEME_TILE_OCC) void k_tile_step(const DevParams p, const StepArgs sa) {
    const TileArgs& a = sa.t;
    extern __shared__ uint4 tileLds[];
    uint4* const sOwn = tileLds;
    const uint32_t RSZ = a.rs16;
    float4* const recA4 = reinterpret_cast<float4*>(sOwn + (DEME_TILE_NB + a.hCap) * RSZ);
    float4* const recT = recA4 + DEME_TILE_RSLOTS;
    float2* const recA2 = reinterpret_cast<float2*>(recT + DEME_TILE_RSLOTS);
    uint16_t* const sALo = reinterpret_cast<uint16_t*>(recA2 + DEME_TILE_RSLOTS);  // (a tile's contact range is at most DEME_TILE_CMAX long)
    uint16_t* const sLLo = sALo + (DEME_TILE_NB + 1);
    uint16_t* const sLPos = sALo + DEME_TILE_BOUNDS_BYTES / 2u;  // (16-byte aligned)
    const uint32_t t = tile_block_id(a.xcdGroup);
    if (t >= a.nTiles)
        return;
    const uint32_t tid = threadIdx.x;
    const uint32_t o0 = t * DEME_TILE_NB;
    const uint32_t nLoc = min((uint32_t)DEME_TILE_NB, a.nOwners - o0);
    // ---- loads that need only the tile number (see k_tile_forces)
    const uint32_t* hl = a.hList + (size_t)t * DEME_TILE_HMAX;
    const uint32_t h0 = tid - nLoc, h1 = tid + DEME_TILE_T - nLoc;
    uint32_t id0 = (tid >= nLoc && h0 < DEME_TILE_HMAX) ? hl[h0] : 0u;
    uint32_t id1 = (h1 < DEME_TILE_HMAX) ? hl[h1] : 0u;
    OwnerRec rec0;
    if (tid < nLoc)
        rec0 = load_owner(a.owners, o0 + tid);
    TileTables T;
    uint4* const tabBase = reinterpret_cast<uint4*>(reinterpret_cast<char*>(sLPos) + ((a.lCap * 2u + 15u) & ~15u));
    const uint32_t nTab16 = a.nComp + p.nMat * p.nMat * 2u + a.nAnal * 4u;
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
    const uint32_t nH = a.hCount[t];
    const uint32_t c0 = a.aStart[o0], c1 = a.aStart[o0 + nLoc];
    const uint32_t k0 = sa.inStart[o0], k1 = sa.inStart[o0 + nLoc];
    const uint32_t nOwnC = c1 - c0, nCt = nOwnC + (k1 - k0);  // the tile's stream: its own contacts, then the incoming ones
    const uint16_t* const lOffT = a.lOff + (size_t)t * (DEME_TILE_NB + 1);
    const uint32_t nL = a.lCount[t];
    const float4* wc4 = reinterpret_cast<const float4*>(a.wc);
    uint2 inf[DEME_TILE_DEPTH];
    float4 hist[DEME_TILE_DEPTH];
    auto fetch = [&](const int d, const uint32_t q) __attribute__((always_inline)) {  // stage d takes stream position q
        inf[d] = make_uint2(0, 0), hist[d] = make_float4(0, 0, 0, 0);
        if (q < nOwnC) {
            inf[d] = stream_load(a.tInfo + c0 + q);
            if (MODEL == 0)
                hist[d] = stream_load(wc4 + c0 + q);
        } else if (q < nCt) {
            const uint32_t k = k0 + (q - nOwnC);
            inf[d] = stream_load(sa.tInfoIn + k);
            if (MODEL == 0)
                hist[d] = wc4[sa.inContact[k]];  // (a gather: the history lives with A's tile)
        }
    };
#pragma unroll
    for (int d = 0; d < DEME_TILE_DEPTH; d++)
        fetch(d, tid + d * DEME_TILE_T);
    uint32_t bA = 0, bL = 0, bI = 0;
    if (tid <= DEME_TILE_NB) {
        const uint32_t o = min(tid, nLoc);
        bA = a.aStart[o0 + o], bL = lOffT[o], bI = sa.inStart[o0 + o];
    }
    uint32_t lp[DEME_TILE_LREG];
#pragma unroll
    for (int k = 0; k < DEME_TILE_LREG; k++) {
        const uint32_t i = tid + k * DEME_TILE_T;
        lp[k] = (i < nL) ? (uint32_t)a.lPos[c0 + i] : 0u;
    }
    OwnerRec rec1;
    if (tid >= nLoc && h0 < nH)
        rec0 = load_owner(a.owners, id0);
    if (h1 < nH)
        rec1 = load_owner(a.owners, id1);
    if (tid < nTab16)
        tabBase[tid] = tab16;
    if (tid < a.nMass)
        const_cast<float*>(T.mass)[tid] = tabMass;
    if (!p.familyTrivial)
        const_cast<float*>(T.fam)[tid & 255u] = tabFam;
    __syncthreads();
    {
        // ONE origin for every tile: the world's (fp64 positions: 1e-16 m at a metre).  Two tiles that evaluate the same contact
        // stage the same numbers, so their results are the same bits.
        if (tid < nLoc + nH)
            tile_stage<MODEL>(p, T.mass[rec0.inertiaOff], rec0, 0, 0, 0, sOwn + (tid < nLoc ? tid : DEME_TILE_NB + h0) * RSZ, ((tid < nLoc ? tid : DEME_TILE_NB + h0) >> 3) & a.swz);
        if (h1 < nH)
            tile_stage<MODEL>(p, T.mass[rec1.inertiaOff], rec1, 0, 0, 0, sOwn + (DEME_TILE_NB + h1) * RSZ, ((DEME_TILE_NB + h1) >> 3) & a.swz);
        if (tid <= DEME_TILE_NB)
            sALo[tid] = (uint16_t)(bA - c0), sLLo[tid] = (uint16_t)bL;
        if (tid == DEME_TILE_T - 1u)
            recA4[DEME_TILE_T] = make_float4(0, 0, 0, 0), recT[DEME_TILE_T] = make_float4(0, 0, 0, 0), recA2[DEME_TILE_T] = make_float2(0, 0);
#pragma unroll
        for (int k = 0; k < DEME_TILE_LREG; k++)
            if (tid + k * DEME_TILE_T < nL)
                sLPos[tid + k * DEME_TILE_T] = (uint16_t)lp[k];
    }
    __syncthreads();
    const uint32_t po = tid % DEME_TILE_NB;
    const bool sideA = tid < DEME_TILE_NB, sideB = !sideA && tid < 2 * DEME_TILE_NB;
    uint32_t plo = sideB ? sLLo[po] : sALo[po];
    const uint32_t phi = (sideA || sideB) ? (sideB ? sLLo[po + 1] : sALo[po + 1]) : plo;
    // my owner's incoming run (side B): stream positions [ilo, ihi)
    __shared__ uint32_t sIn[DEME_TILE_NB + 1];
    if (tid <= DEME_TILE_NB)
        sIn[tid] = nOwnC + (bI - k0);
    __syncthreads();
    uint32_t ilo = sideB ? sIn[po] : 0u;
    const uint32_t ihi = sideB ? sIn[po + 1] : 0u;
    v2f s01 = {0.f, 0.f}, s23 = {0.f, 0.f}, s45 = {0.f, 0.f};
    float4* const wcOut = reinterpret_cast<float4*>(sa.wcNext);
    for (uint32_t rlo = 0; rlo < nCt; rlo += DEME_TILE_T) {
        const uint32_t q = rlo + tid;
        if (q < nCt) {
            const uint2 ci = inf[0];
            float4 h = hist[0];
            const uint32_t slotA = ci.x & 1023u, slotB = (ci.x >> 10) & 1023u;
            const TileOwner A = tile_read<MODEL>(sOwn, slotA * RSZ, (slotA >> 3) & a.swz), B = tile_read<MODEL>(sOwn, slotB * RSZ, (slotB >> 3) & a.swz);
            f3 force, tA, tB;
            tile_contact<MODEL>(p, T, ci, A, B, h, force, tA, tB);
            if (MODEL == 0 && q < nOwnC && !sa.dry)
                stream_store(wcOut + c0 + q, h);  // (the new history: written by A's tile only)
            recA4[tid] = make_float4(force.x, force.y, force.z, tA.x);
            recA2[tid] = make_float2(tA.y, tA.z);
            if (slotB < DEME_TILE_NB)
                recT[tid] = make_float4(tB.y, tB.z, tB.x, 0.f);
        }
#pragma unroll
        for (int d = 0; d + 1 < DEME_TILE_DEPTH; d++)
            inf[d] = inf[d + 1], hist[d] = hist[d + 1];
        fetch(DEME_TILE_DEPTH - 1, q + DEME_TILE_DEPTH * DEME_TILE_T);
        __syncthreads();
        const uint32_t rhi = rlo + DEME_TILE_T;
        if (sideA) {
            const uint32_t e = min(phi, rhi);
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
        } else if (sideB) {
            // my local-B list's entries of this round (own contacts of the tile that hold my owner as B) ...
            while (plo < phi) {
                uint32_t pos[4];
#pragma unroll
                for (int k = 0; k < 4; k++)
                    pos[k] = (plo + k < phi) ? (uint32_t)sLPos[plo + k] : 0xFFFFFFFFu;
                if (pos[0] >= rhi)
                    break;
                float4 v4[4], vt[4];
                uint32_t used = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const bool in = pos[k] < rhi;
                    const uint32_t i = in ? pos[k] - rlo : (uint32_t)DEME_TILE_T;
                    v4[k] = recA4[i], vt[k] = recT[i];
                    used += in ? 1u : 0u;
                }
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    s01 -= v2f{v4[k].x, v4[k].y};
                    s23.x -= v4[k].z;
                    s23.y += vt[k].z;
                    s45 += v2f{vt[k].x, vt[k].y};
                }
                plo += used;
                if (used < 4u)
                    break;
            }
            // ... then my incoming run's part of this round (contacts of OTHER tiles' owners with my owner as B): positions
            // [ilo, min(ihi, rhi)), all behind the tile's own contacts -- the order of an owner's sum is fixed: own list, then incoming
            const uint32_t e = min(ihi, rhi);
            while (ilo < e) {
                float4 v4[4], vt[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const uint32_t i = (ilo + k < e) ? ilo + k - rlo : (uint32_t)DEME_TILE_T;
                    v4[k] = recA4[i], vt[k] = recT[i];
                }
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    s01 -= v2f{v4[k].x, v4[k].y};
                    s23.x -= v4[k].z;
                    s23.y += vt[k].z;
                    s45 += v2f{vt[k].x, vt[k].y};
                }
                ilo = min(ilo + 4u, e);
            }
        }
        __syncthreads();
    }
    // ---- the owners' sums are complete: integrate (k_integrate's per-owner part), or -- dry run -- leave a / alpha
    if (sideB) {
        recA4[po] = make_float4(s01.x, s01.y, s23.x, s23.y);
        recA2[po] = make_float2(s45.x, s45.y);
    }
    __syncthreads();
    if (sideA && po < nLoc) {
        const float4 b4 = recA4[po];
        const float2 b2 = recA2[po];
        const uint32_t o = o0 + po;
        OwnerRec r = load_owner(a.owners, o);  // (again: 64 bytes from L2; kept in registers it would cost the loop 16 VGPRs)
        const uint32_t fflags = p.familyFlags[fam_of(r.family)];
        const bool fixed = (fflags & 1u) != 0;
        float4 acc4 = make_float4(s01.x + b4.x, s01.y + b4.y, s23.x + b4.z, 0.f);
        float4 al4 = make_float4(s23.y + b4.w, s45.x + b2.x, s45.y + b2.y, 0.f);
        if (!fixed)
            acc_from_world(p, r, acc4, al4);
        else
            acc4 = make_float4(0, 0, 0, 0), al4 = acc4;
        if (sa.dry) {
            float4* ap = reinterpret_cast<float4*>(sa.acc + o);
            ap[0] = acc4, ap[1] = al4;
        } else {
            GatherArgs g{};
            g.nextAcc = sa.nextAcc;
            PrescArgs pa{nullptr, nullptr};
            integrate_owner(p, r, acc4, al4, o, fflags, fixed, g, pa);
            store_owner(sa.ownersNext, o, r);
        }
    }
}

// ---- per-detection builders of the incoming lists --------------------------------------------------------------------------------
// (1) per owner: how many contacts hold it as B from another tile -- none for an owner nobody integrates (fixed family)
__global__ __launch_bounds__(256) void k_in_count(const DevParams p, const OwnerRec* __restrict__ owners, const uint32_t* __restrict__ rStart,
                                                  uint32_t* __restrict__ inCnt) {
    const uint32_t o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o > p.nOwners)
        return;
    uint32_t n = 0;
    if (o < p.nOwners) {
        const uint32_t fw = owners[o].family;
        if (!(p.familyFlags[fam_of(fw)] & 1u))
            n = rStart[o + 1] - rStart[o];
    }
    inCnt[o] = n;
}
// (2) one workgroup per tile: the 8-byte gather records of its incoming contacts in ITS frame -- B's slot is the owner's place in the
// tile, A's the place of its (foreign) owner in the tile's halo list, which is extended by the owners only incoming contacts name
__global__ __launch_bounds__(256) void k_tile_incoming(const DevParams p, uint32_t nOwners, const uint4* __restrict__ info,
                                                       const uint32_t* __restrict__ rStart, const uint32_t* __restrict__ rIdx,
                                                       const uint32_t* __restrict__ recContact, const uint32_t* __restrict__ inStart,
                                                       uint32_t* __restrict__ hList, const uint32_t* __restrict__ hCount,
                                                       uint32_t* __restrict__ hCountIn, uint2* __restrict__ tInfoIn, uint32_t* __restrict__ inContact, RangeCounters* rc) {
    __shared__ uint32_t table[DEME_TILE_HASH];
    __shared__ uint16_t slotTab[DEME_TILE_HASH];
    __shared__ uint32_t sIn[DEME_TILE_NB + 1], sR[DEME_TILE_NB + 1];
    __shared__ uint32_t nU;
    const uint32_t t = blockIdx.x, tid = threadIdx.x;
    const uint32_t o0 = t * DEME_TILE_NB, o1 = min(o0 + (uint32_t)DEME_TILE_NB, nOwners), nLoc = o1 - o0;
    for (uint32_t i = tid; i < DEME_TILE_HASH; i += 256)
        table[i] = 0xFFFFFFFFu;
    const uint32_t nH0 = hCount[t];
    if (tid == 0)
        nU = nH0;
    if (tid <= nLoc)
        sIn[tid] = inStart[o0 + tid], sR[tid] = rStart[o0 + tid];
    __syncthreads();
    uint32_t* hl = hList + (size_t)t * DEME_TILE_HMAX;
    for (uint32_t i = tid; i < nH0; i += 256) {  // the halo the tile's own contacts gave it
        const uint32_t ob = hl[i];
        uint32_t h = (ob * 2654435761u) >> 22;
        while (atomicCAS(&table[h], 0xFFFFFFFFu, ob) != 0xFFFFFFFFu)
            h = (h + 1u) & (DEME_TILE_HASH - 1u);
        slotTab[h] = (uint16_t)i;
    }
    __syncthreads();
    const uint32_t e0 = sIn[0], e1 = sIn[nLoc];
    // pass 1: owners that only incoming contacts name join the halo
    for (uint32_t e = e0 + tid; e < e1; e += 256) {
        uint32_t lo = 0, hi = nLoc;  // the owner of entry e: the last o with sIn[o] <= e
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (sIn[mid] <= e)
                lo = mid;
            else
                hi = mid;
        }
        const uint32_t c = recContact[rIdx[sR[lo] + (e - sIn[lo])]];
        const uint32_t oa = info[c].x & 0x3FFFFFFFu;
        uint32_t h = (oa * 2654435761u) >> 22;
        while (*(volatile uint32_t*)&nU <= DEME_TILE_HMAX) {
            const uint32_t old = atomicCAS(&table[h], 0xFFFFFFFFu, oa);
            if (old == 0xFFFFFFFFu) {
                const uint32_t slot = atomicAdd(&nU, 1u);
                slotTab[h] = (uint16_t)slot;
                if (slot < DEME_TILE_HMAX)
                    hl[slot] = oa;
                break;
            }
            if (old == oa)
                break;
            h = (h + 1u) & (DEME_TILE_HASH - 1u);
        }
    }
    __syncthreads();
    // (the open form -- k_tile_forces + k_integrate -- keeps its own count: the extension sits BEHIND the entries it stages)
    if (nU > DEME_TILE_HMAX) {  // the closed tile does not fit: the list keeps the open form
        if (tid == 0)
            atomicAdd(&rc->nBigIn, 1u);
        return;
    }
    if (tid == 0) {
        hCountIn[t] = nU;
        atomicMax(&rc->tileMaxHaloIn, nU);
    }
    // pass 2: the records
    for (uint32_t e = e0 + tid; e < e1; e += 256) {
        uint32_t lo = 0, hi = nLoc;
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (sIn[mid] <= e)
                lo = mid;
            else
                hi = mid;
        }
        const uint32_t c = recContact[rIdx[sR[lo] + (e - sIn[lo])]];
        const uint4 ci = info[c];
        const uint32_t oa = ci.x & 0x3FFFFFFFu, cls = ci.x >> 30;
        uint32_t h = (oa * 2654435761u) >> 22;
        while (table[h] != oa)
            h = (h + 1u) & (DEME_TILE_HASH - 1u);
        const uint32_t slotA = DEME_TILE_NB + slotTab[h];
        const uint32_t matA = ci.z >> 16, compA = ci.z & 0xFFFFu;
        const uint32_t matB = (cls == DEME_KEY_CLASS_SS) ? (ci.w >> 16) : 0u, wB = (cls == DEME_KEY_CLASS_SS) ? (ci.w & 0xFFFFu) : ci.w;
        tInfoIn[e] = make_uint2(tile_info_x(slotA, lo, cls, 0u, matA, matB), compA | (wB << 16));
        inContact[e] = c;
    }
}

}  // namespace deme_dev

#pragma clang fp contract(off)

#endif  // DEME_JIT
