// deme_migrate.h -- device side of deme_halo_group_migrate (slab decomposition, SURVEY 8e "migration at re-bin time").
//
// A slab's owners are laid out [own clumps | ghosts from the left | ghosts from the right | replicated owners (walls, meshes)],
// its spheres clump-major.  A migration keeps the slab edges and talks to the face neighbours only:
//   1. an own clump whose centre crossed a face leaves for that neighbour with its record, its template ids and the history of
//      every contact it takes part in (rows in GLOBAL sphere ids, sphere pairs smaller id first, B->A vector wildcards negated
//      where the local list stored the pair the other way round);
//   2. every slab then sends the clumps it owns within the halo of a shared face: the neighbour's new ghosts;
//   3. the slab is re-assembled in place on the device -- [stay | arrived left | arrived right | ghosts left | ghosts right |
//      replicated] -- and its contact history re-seeded from what it had and what arrived: every known row whose spheres are
//      present and that involves an own clump, the slab's own rows first.
// Nothing goes through the host but counts.  The algorithm is decomp.migrate_neighbours (the numpy statement the tests compare
// with); the kernels below are its steps.
#pragma once
#include "deme_device.h"

namespace deme_dev {

struct __attribute__((aligned(8))) MigClump {  // one travelling clump: 72 bytes
    OwnerRec rec;
    uint32_t gid;   // global clump id
    uint32_t nsph;
};
struct MigSphere {  // its spheres follow in a second stream, clump-major
    uint32_t gid;   // global sphere id
    uint16_t comp, mat;
};
#define DEME_MIG_MAXW 8u
struct MigRowHead {  // a history row: header + nW floats
    uint32_t gA, gB, cls;  // cls: contact class (2 bits) | DEME_MIG_PERSIST_BIT
};
#define DEME_MIG_PERSIST_BIT 0x100u

struct MigCount {  // per class (0 stay, 1 to the left, 2 to the right): clumps and spheres; a scan element
    uint32_t c[3], s[3];
};
struct MigCountPlus {
    __host__ __device__ MigCount operator()(const MigCount& a, const MigCount& b) const {
        MigCount r;
        for (int k = 0; k < 3; k++)
            r.c[k] = a.c[k] + b.c[k], r.s[k] = a.s[k] + b.s[k];
        return r;
    }
};

// first sphere of every owner (spheres are clump-major): start[o] for o = 0 .. nOwners
__global__ __launch_bounds__(256) void k_mig_first_sphere(uint32_t nS, const SphereRec* __restrict__ spheres, uint32_t nOwners,
                                                          uint32_t* __restrict__ start) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (nS == 0) {
        if (i <= nOwners)
            start[i] = 0;
        return;
    }
    if (i >= nS)
        return;
    const uint32_t cur = spheres[i].owner;
    const uint32_t first = (i == 0) ? 0u : spheres[i - 1].owner + 1u;
    for (uint32_t o = first; o <= cur && o <= nOwners; o++)
        start[o] = i;
    if (i == nS - 1)
        for (uint32_t o = cur + 1; o <= nOwners; o++)
            start[o] = nS;
}

// a clump centre's coordinate along the axis the slabs were cut along (0 / 1 / 2; deme_halo_group_set_axis)
__device__ inline double mig_world_coord(const OwnerRec& r, const DevParams& p, int axis) {
    if (axis == 0)
        return (double)(r.voxelID & (((uint64_t)1 << p.nvXp2) - 1)) * p.voxelSize + (double)r.locX * p.l + (double)p.LBFX;
    if (axis == 1)
        return (double)((r.voxelID >> p.nvXp2) & (((uint64_t)1 << p.nvYp2) - 1)) * p.voxelSize + (double)r.locY * p.l + (double)p.LBFY;
    return (double)(r.voxelID >> (p.nvXp2 + p.nvYp2)) * p.voxelSize + (double)r.locZ * p.l + (double)p.LBFZ;
}

// dest[o]: 0 stays, 1 leaves to the left, 2 to the right, 3 a ghost (dropped: the neighbours send their ghosts anew);
// cnt[o] = the scan element (cnt[nClumps] = zero closes the scan)
__global__ __launch_bounds__(256) void k_mig_classify(const DevParams p, const OwnerRec* __restrict__ owners, uint32_t nOwn,
                                                      uint32_t nClumps, int axis, double xLo, double xHi, const uint32_t* __restrict__ firstSph,
                                                      uint8_t* __restrict__ dest, MigCount* __restrict__ cnt) {
    const uint32_t o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o > nClumps)
        return;
    MigCount m{};
    if (o < nClumps) {
        uint8_t d = 3;
        if (o < nOwn) {
            const double x = mig_world_coord(owners[o], p, axis);
            d = x < xLo ? 1 : (x >= xHi ? 2 : 0);
            m.c[d] = 1, m.s[d] = firstSph[o + 1] - firstSph[o];
        }
        dest[o] = d;
    }
    cnt[o] = m;
}

// class d's clumps into their packet (d = 1, 2) -- or, for d = 0, straight into the new arrays (base offsets 0)
__global__ __launch_bounds__(256) void k_mig_pack_clumps(uint32_t nClumps, const OwnerRec* __restrict__ owners,
                                                         const SphereRec* __restrict__ spheres, const uint32_t* __restrict__ ownerGid,
                                                         const uint32_t* __restrict__ sphereGid, const uint32_t* __restrict__ firstSph,
                                                         const uint8_t* __restrict__ dest, const MigCount* __restrict__ pos, uint32_t d,
                                                         MigClump* __restrict__ outC, MigSphere* __restrict__ outS) {
    const uint32_t o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= nClumps || dest[o] != d)
        return;
    const uint32_t fs = firstSph[o], ns = firstSph[o + 1] - fs;
    MigClump m;
    m.rec = owners[o];
    m.rec.family &= ~(OWNER_GHOST_BIT | OWNER_PASSIVE_BIT);
    m.gid = ownerGid[o];
    m.nsph = ns;
    outC[pos[o].c[d]] = m;
    for (uint32_t k = 0; k < ns; k++) {
        const SphereRec sr = load_sphere(spheres, fs + k);
        MigSphere ms;
        ms.gid = sphereGid[fs + k], ms.comp = sr.comp, ms.mat = sr.mat;
        outS[pos[o].s[d] + k] = ms;
    }
}

// a packet section into the new arrays: clumps [ownerBase ..), spheres [sphBase ..); sphOff = exclusive scan of the packet's nsph
__global__ __launch_bounds__(256) void k_mig_unpack(uint32_t n, const MigClump* __restrict__ inC, const MigSphere* __restrict__ inS,
                                                    const uint32_t* __restrict__ sphOff, uint32_t ownerBase, uint32_t sphBase,
                                                    uint32_t ghostBit, OwnerRec* __restrict__ owners, SphereRec* __restrict__ spheres,
                                                    uint32_t* __restrict__ ownerGid, uint32_t* __restrict__ sphereGid) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    MigClump m = inC[i];
    m.rec.family = (m.rec.family & ~(OWNER_GHOST_BIT | OWNER_PASSIVE_BIT)) | ghostBit;
    owners[ownerBase + i] = m.rec;
    ownerGid[ownerBase + i] = m.gid;
    const uint32_t s0 = sphOff[i];
    for (uint32_t k = 0; k < m.nsph; k++) {
        const MigSphere ms = inS[s0 + k];
        SphereRec sr;
        sr.owner = ownerBase + i, sr.comp = ms.comp, sr.mat = ms.mat;
        *reinterpret_cast<uint2*>(spheres + sphBase + s0 + k) = make_uint2(sr.owner, (uint32_t)sr.comp | ((uint32_t)sr.mat << 16));
        sphereGid[sphBase + s0 + k] = ms.gid;
    }
}
__global__ __launch_bounds__(256) void k_mig_nsph(uint32_t n, const MigClump* __restrict__ inC, uint32_t* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= n)
        out[i] = i < n ? inC[i].nsph : 0u;
}

// the slab's whole history in global form; toL / toR: the row travels with a clump that leaves
__global__ __launch_bounds__(256) void k_mig_rows(uint32_t nC, uint32_t nW, uint32_t flipMask, const uint64_t* __restrict__ keys,
                                                  const float* __restrict__ wc, const SphereRec* __restrict__ spheres,
                                                  const uint32_t* __restrict__ sphereGid, const uint8_t* __restrict__ dest,
                                                  uint32_t nClumps, MigRowHead* __restrict__ head, float* __restrict__ rw,
                                                  uint32_t* __restrict__ toL, uint32_t* __restrict__ toR,
                                                  const uint64_t* __restrict__ persist, uint32_t nPersist) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c > nC)
        return;
    if (c == nC) {
        toL[c] = 0, toR[c] = 0;
        return;
    }
    const uint64_t k = keys[c];
    const uint32_t A = key_a(k), B = key_b(k), cls = key_class(k);
    const uint32_t oA = spheres[A].owner;
    uint32_t gA = sphereGid[A], gB = B, oB = 0xFFFFFFFFu;
    bool flip = false;
    if (cls == DEME_KEY_CLASS_SS) {
        oB = spheres[B].owner;
        gB = sphereGid[B];
        if (gA > gB) {
            const uint32_t t = gA;
            gA = gB, gB = t;
            flip = true;
        }
    }
    uint32_t mark = 0;  // a persistent contact (deme_mark_persistent_contacts) keeps its mark wherever its clumps go: bit 8 of the class word
    if (nPersist) {
        uint32_t lo = 0, hi = nPersist;
        while (lo < hi) {
            const uint32_t mid = lo + ((hi - lo) >> 1);
            if (persist[mid] < k)
                lo = mid + 1;
            else
                hi = mid;
        }
        mark = (lo < nPersist && persist[lo] == k) ? DEME_MIG_PERSIST_BIT : 0u;
    }
    head[c] = MigRowHead{gA, gB, cls | mark};
    for (uint32_t w = 0; w < nW; w++) {
        const float v = wc[(size_t)c * nW + w];
        rw[(size_t)c * nW + w] = (flip && ((flipMask >> w) & 1u)) ? -v : v;
    }
    const uint8_t dA = oA < nClumps ? dest[oA] : 0, dB = (oB != 0xFFFFFFFFu && oB < nClumps) ? dest[oB] : 0;
    toL[c] = (dA == 1 || dB == 1) ? 1u : 0u;
    toR[c] = (dA == 2 || dB == 2) ? 1u : 0u;
}
__global__ __launch_bounds__(256) void k_mig_pack_rows(uint32_t nC, uint32_t nW, const MigRowHead* __restrict__ head,
                                                       const float* __restrict__ rw, const uint32_t* __restrict__ flag,
                                                       const uint32_t* __restrict__ pos, MigRowHead* __restrict__ outH,
                                                       float* __restrict__ outW) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nC || !flag[c])
        return;
    const uint32_t j = pos[c];
    outH[j] = head[c];
    for (uint32_t w = 0; w < nW; w++)
        outW[(size_t)j * nW + w] = rw[(size_t)c * nW + w];
}

// the own clumps of the re-assembled slab that lie within the halo of a face: flags for the two ghost packets
__global__ __launch_bounds__(256) void k_mig_ghost_flags(const DevParams p, const OwnerRec* __restrict__ owners, uint32_t nOwn, int axis,
                                                         double xLo, double xHi, double halo, int hasLeft, int hasRight,
                                                         const uint32_t* __restrict__ firstSphNew, uint8_t* __restrict__ dest,
                                                         MigCount* __restrict__ cnt) {
    // classes here: 1 = to the left neighbour, 2 = to the right one (a clump can be both in a slab thinner than two halos: it
    // then goes to the left with class 1 and to the right through the second flag array of the host loop -- see k_mig_classify2)
    const uint32_t o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o > nOwn)
        return;
    MigCount m{};
    if (o < nOwn) {
        const double x = mig_world_coord(owners[o], p, axis);
        const uint32_t ns = firstSphNew[o + 1] - firstSphNew[o];
        uint8_t d = 0;
        if (hasLeft && x < xLo + halo)
            d |= 1, m.c[1] = 1, m.s[1] = ns;
        if (hasRight && x >= xHi - halo)
            d |= 2, m.c[2] = 1, m.s[2] = ns;
        dest[o] = d;
    }
    cnt[o] = m;
}
// ghost packets: like k_mig_pack_clumps, for the bit-coded classes of k_mig_ghost_flags; also the send list (new owner ids)
__global__ __launch_bounds__(256) void k_mig_pack_ghosts(uint32_t nOwn, const OwnerRec* __restrict__ owners,
                                                         const SphereRec* __restrict__ spheres, const uint32_t* __restrict__ ownerGid,
                                                         const uint32_t* __restrict__ sphereGid, const uint32_t* __restrict__ firstSph,
                                                         const uint8_t* __restrict__ dest, const MigCount* __restrict__ pos, uint32_t d,
                                                         MigClump* __restrict__ outC, MigSphere* __restrict__ outS,
                                                         uint32_t* __restrict__ sendIds) {
    const uint32_t o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= nOwn || !(dest[o] & d))
        return;
    const uint32_t fs = firstSph[o], ns = firstSph[o + 1] - fs;
    MigClump m;
    m.rec = owners[o];
    m.gid = ownerGid[o];
    m.nsph = ns;
    const uint32_t j = pos[o].c[d];
    outC[j] = m;
    sendIds[j] = o;
    for (uint32_t k = 0; k < ns; k++) {
        const SphereRec sr = load_sphere(spheres, fs + k);
        MigSphere ms;
        ms.gid = sphereGid[fs + k], ms.comp = sr.comp, ms.mat = sr.mat;
        outS[pos[o].s[d] + k] = ms;
    }
}

// user wildcard arrays (deme_compile_force_model_ex: up to 8 per-owner and 8 per-sphere float arrays) travel beside the clump packets
// in two more streams: [clump][k] and [sphere][k], in the packets' own order
struct MigWcPtrs {
    float* p[8];
};
__global__ __launch_bounds__(256) void k_mig_pack_wc(uint32_t nClumps, const uint8_t* __restrict__ dest, const MigCount* __restrict__ pos,
                                                     uint32_t d, int bitCoded, const uint32_t* __restrict__ firstSph, uint32_t nOW,
                                                     uint32_t nGW, MigWcPtrs ow, MigWcPtrs gw, float* __restrict__ outO,
                                                     float* __restrict__ outS) {
    const uint32_t o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= nClumps)
        return;
    if (bitCoded ? !(dest[o] & d) : dest[o] != d)
        return;
    const uint32_t j = pos[o].c[d];
    for (uint32_t k = 0; k < nOW; k++)
        outO[(size_t)j * nOW + k] = ow.p[k][o];
    if (nGW) {
        const uint32_t fs = firstSph[o], ns = firstSph[o + 1] - fs, sj = pos[o].s[d];
        for (uint32_t s = 0; s < ns; s++)
            for (uint32_t k = 0; k < nGW; k++)
                outS[(size_t)(sj + s) * nGW + k] = gw.p[k][fs + s];
    }
}
// sphOff: the exclusive scan of the packet's sphere counts over n + 1 entries (sphOff[n] = the packet's spheres)
__global__ __launch_bounds__(256) void k_mig_unpack_wc(uint32_t n, const float* __restrict__ inO, const float* __restrict__ inS,
                                                       const uint32_t* __restrict__ sphOff, uint32_t ownerBase, uint32_t sphBase, uint32_t nOW,
                                                       uint32_t nGW, MigWcPtrs ow, MigWcPtrs gw) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    for (uint32_t k = 0; k < nOW; k++)
        ow.p[k][ownerBase + i] = inO[(size_t)i * nOW + k];
    if (nGW) {
        const uint32_t s0 = sphOff[i], ns = sphOff[i + 1] - s0;
        for (uint32_t s = 0; s < ns; s++)
            for (uint32_t k = 0; k < nGW; k++)
                gw.p[k][sphBase + s0 + s] = inS[(size_t)(s0 + s) * nGW + k];
    }
}
__global__ __launch_bounds__(256) void k_mig_copy_extras_wc(uint32_t n, uint32_t oldBase, uint32_t newBase, uint32_t nOW, MigWcPtrs oldW,
                                                            MigWcPtrs newW) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    for (uint32_t k = 0; k < nOW; k++)
        newW.p[k][newBase + i] = oldW.p[k][oldBase + i];
}

__global__ __launch_bounds__(256) void k_mig_copy_extras(uint32_t n, const OwnerRec* __restrict__ oldOwners, const uint32_t* __restrict__ oldGid,
                                                         uint32_t oldBase, uint32_t newBase, OwnerRec* __restrict__ owners,
                                                         uint32_t* __restrict__ ownerGid) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    owners[newBase + i] = oldOwners[oldBase + i];
    ownerGid[newBase + i] = oldGid[oldBase + i];
}
__global__ __launch_bounds__(256) void k_mig_shift_anal(uint32_t n, AnalObj* __restrict__ anal, int shift) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        anal[i].owner = (uint32_t)((int)anal[i].owner + shift);
}
__global__ __launch_bounds__(256) void k_mig_shift_tris(uint32_t n, uint32_t* __restrict__ tris, int shift) {  // TriRec: owner at word 9
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        tris[12 * (size_t)i + 9] = (uint32_t)((int)tris[12 * (size_t)i + 9] + shift);
}

__global__ __launch_bounds__(256) void k_mig_iota_gid(uint32_t n, const uint32_t* __restrict__ gid, uint32_t* __restrict__ keys,
                                                      uint32_t* __restrict__ vals) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        keys[i] = gid[i], vals[i] = i;
}
__device__ inline uint32_t mig_lookup(const uint32_t* sortedGid, const uint32_t* sortedIdx, uint32_t n, uint32_t g) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (sortedGid[mid] < g)
            lo = mid + 1;
        else
            hi = mid;
    }
    return (lo < n && sortedGid[lo] == g) ? sortedIdx[lo] : 0xFFFFFFFFu;
}
// candidate rows (the slab's own first, then what arrived) -> local keys in the re-assembled numbering; key = all ones: dropped
__global__ __launch_bounds__(256) void k_mig_localise(uint32_t nRows, uint32_t nW, uint32_t flipMask, const MigRowHead* __restrict__ head,
                                                      float* __restrict__ rw, const uint32_t* __restrict__ sortedGid,
                                                      const uint32_t* __restrict__ sortedIdx, uint32_t nS,
                                                      const SphereRec* __restrict__ spheres, uint32_t nOwn, uint64_t* __restrict__ keys,
                                                      uint32_t* __restrict__ rowIdx) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nRows)
        return;
    rowIdx[i] = i;
    MigRowHead h = head[i];
    h.cls &= 3u;  // (bit 8 is the persistent mark: k_mig_seed reads it from the row)
    uint32_t la = mig_lookup(sortedGid, sortedIdx, nS, h.gA), lb = h.gB;
    bool ok = la != 0xFFFFFFFFu;
    bool own = ok && spheres[la].owner < nOwn;
    if (h.cls == DEME_KEY_CLASS_SS) {
        lb = mig_lookup(sortedGid, sortedIdx, nS, h.gB);
        ok = ok && lb != 0xFFFFFFFFu;
        own = ok && (own || spheres[lb].owner < nOwn);
    }
    if (!ok || !own) {
        keys[i] = ~0ull;
        return;
    }
    if (h.cls == DEME_KEY_CLASS_SS && la > lb) {
        const uint32_t t = la;
        la = lb, lb = t;
        for (uint32_t w = 0; w < nW; w++)
            if ((flipMask >> w) & 1u)
                rw[(size_t)i * nW + w] = -rw[(size_t)i * nW + w];
    }
    keys[i] = make_key(h.cls, la, lb);
}
// after the stable sort by key: keep the first row of every key (the slab's own history wins), drop the all-ones tail
__global__ __launch_bounds__(256) void k_mig_keep_flags(uint32_t n, const uint64_t* __restrict__ keys, uint32_t* __restrict__ keep) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n)
        return;
    keep[i] = (i < n && keys[i] != ~0ull && (i == 0 || keys[i] != keys[i - 1])) ? 1u : 0u;
}
__global__ __launch_bounds__(256) void k_mig_seed(uint32_t n, uint32_t nW, const uint64_t* __restrict__ keys, const uint32_t* __restrict__ rowIdx,
                                                  const uint32_t* __restrict__ keep, const uint32_t* __restrict__ pos,
                                                  const float* __restrict__ rw, uint64_t* __restrict__ outKeys, float* __restrict__ outWc,
                                                  const MigRowHead* __restrict__ head, uint8_t* __restrict__ outMark) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !keep[i])
        return;
    const uint32_t j = pos[i], r = rowIdx[i];
    outKeys[j] = keys[i];
    if (outMark)
        outMark[j] = (head[r].cls & DEME_MIG_PERSIST_BIT) ? 1 : 0;
    for (uint32_t w = 0; w < nW; w++)
        outWc[(size_t)j * nW + w] = rw[(size_t)r * nW + w];
}
__global__ __launch_bounds__(256) void k_mig_iota(uint32_t n, uint32_t base, uint32_t* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        out[i] = base + i;
}

}  // namespace deme_dev
