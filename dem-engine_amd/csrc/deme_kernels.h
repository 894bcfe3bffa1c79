// deme_kernels.h -- contact-detection, margin and integration kernels (gfx950).
//
// Pipeline of one contact-detection update (replaces algorithms/DEMCubContactDetection.cu:38-1123,
// which runs ~7 radix sorts, 6 scans, 3 run-length encodes and >=10 host syncs):
//   k_sphere_prep   sphere world geometry + bin span + sphere-analytical contacts   (1 pass)
//   exclusive scan  of per-sphere bin counts                                         (rocPRIM)
//   k_fill_incidence (bin, sphere) pairs in sphere order
//   radix sort      pairs by bin (only ceil(log2(nBins)) key bits)                   (rocPRIM)
//   k_sweep         LDS-staged chunks of the bin-sorted list, all pairs inside a bin,
//                   contact-point-in-this-bin de-duplication, wavefront ballot/prefix
//                   compaction into 64-bit contact keys (single sweep, no count pass)
//   radix sort      contact keys -> canonical order (type class, A, B)               (rocPRIM)
//   k_history       new->previous index by binary search over the previous sorted keys
#pragma once
#include "deme_device.h"

namespace deme_dev {

#define DEME_NULL_BINID_DEV 0xFFFFFFFFu
// device status bits (host reads them at sync points)
#define DEME_ST_VELOCITY 1u
#define DEME_ST_NONFINITE 2u
#define DEME_ST_INCIDENCE 4u  // a sphere touches more than DEME_MAX_BINS_PER_SPHERE bins: its count would not be trustworthy
#define DEME_MAX_BINS_PER_SPHERE 0xFFFFFFu

struct DetectCounters {  // one 64-byte block of device counters, zeroed per detection
    unsigned long long nContactsRaw;  // keys appended (may exceed capacity: then nothing was written past it)
    unsigned int nActiveBins;
    unsigned int maxInBin;
    unsigned int status;
    unsigned int maxCount;  // largest number of bins one sphere touches (sizing check of the 32-bit incidence offsets)
    unsigned int pad[10];
};

// Where the emitting kernels (k_sphere_prep, k_sweep, k_tri_sweep) append their raw contact keys.  A reservation is an atomic add
// on a counter, and atomics on ONE address retire at ~12 ns each whoever issues them (tools/probes/atomic_probe.hip: 32 768
// workgroups with one reservation each take 400 us on one counter, 67 us on 8, 19 us on 64): with one counter the 32 000 windows
// of k_sweep could not finish in less than 0.4 ms whatever else they did.  So a large arena is cut into DEME_KEY_SEGS equal
// segments with a counter each (128 bytes apart), a workgroup appends to segment blockIdx & segMask, and k_compact_keys closes the
// gaps afterwards (the host has read the counters by then: it needs the total anyway).  Small arenas keep one segment.
#define DEME_KEY_SEGS 64u
#define DEME_KEY_SEG_STRIDE 16u  // in counters (8 bytes each)
struct KeyArena {
    uint64_t* keys;
    unsigned long long* ctr;  // segment g counts at ctr[g * DEME_KEY_SEG_STRIDE]
    uint64_t segCap;          // slots per segment
    uint32_t segMask;         // number of segments - 1 (0: the whole arena is one segment)
};
__device__ inline uint32_t arena_seg(const KeyArena& a) {
    return blockIdx.x & a.segMask;
}
__device__ inline unsigned long long arena_reserve(const KeyArena& a, uint32_t seg, unsigned long long n) {
    return atomicAdd(&a.ctr[seg * DEME_KEY_SEG_STRIDE], n);
}
__device__ inline void arena_store(const KeyArena& a, uint32_t seg, unsigned long long slot, uint64_t key) {
    if (slot < a.segCap)  // (a full segment drops the key; its counter keeps counting and the host grows the arena and repeats)
        a.keys[(uint64_t)seg * a.segCap + slot] = key;
}
// one thread per slot of the segmented arena: the occupied slots move to their place in the contiguous list.  The counts are read
// on the device (every workgroup adds up the 64 of them for its segment's start), so the kernel can be enqueued BEFORE the host
// reads them: the GPU works through that read-back.  xOff: first slot-block this launch covers (the host launches the rest of a
// segment that turned out longer than its estimate).
__global__ __launch_bounds__(256) void k_compact_keys(const uint64_t* __restrict__ in, uint64_t segCap,
                                                      const unsigned long long* __restrict__ segCtr, uint32_t xOff,
                                                      uint64_t* __restrict__ out) {
    __shared__ unsigned long long sStart, sCount;
    const uint32_t seg = blockIdx.y;
    if (threadIdx.x < 64) {
        static_assert(DEME_KEY_SEGS == 64, "one lane per segment");
        const unsigned long long cnt = min(segCtr[threadIdx.x * DEME_KEY_SEG_STRIDE], (unsigned long long)segCap);
        unsigned long long inc = cnt;
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned long long u = __shfl_up(inc, d);
            if ((int)threadIdx.x >= d)
                inc += u;
        }
        if (threadIdx.x == seg)
            sStart = inc - cnt, sCount = cnt;
    }
    __syncthreads();
    const uint64_t i = ((uint64_t)xOff + blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < sCount)
        out[sStart + i] = in[(uint64_t)seg * segCap + i];
}

// ---------------------------------------------------------------------------
// kernel/DEMMiscKernels.cu:37-61 (computeMarginFromAbsv) with the "absv" inspector
// (DEM/AuxClasses.cpp:54-61) and the max-velocity check of DEM/kT.cpp:136-149.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_margins(const DevParams p, OwnerRec* owners, uint32_t drift,
                                                 DetectCounters* ctr) {
    const uint32_t o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= p.nOwners)
        return;
    OwnerRec* r = owners + o;
    const double vx = r->vx, vy = r->vy, vz = r->vz;
    float absv = (float)sqrt(vx * vx + vy * vy + vz * vz);
    if (!isfinite(absv))
        atomicOr(&ctr->status, DEME_ST_NONFINITE);
    else if (absv > p.errOutVel)
        atomicOr(&ctr->status, DEME_ST_VELOCITY);
    if (absv > p.approxMaxVel)
        absv = p.approxMaxVel;
    const float extra = p.familyTrivial ? 0.f : p.familyExtra[fam_of(r->family)];
    r->margin = (float)((double)(absv * p.expSafetyMulti + p.expSafetyAdder) * (double)p.h * (double)drift + (double)extra);
}

__global__ __launch_bounds__(256) void k_set_margins(uint32_t n, OwnerRec* owners, const float* m, const uint32_t* __restrict__ o2e) {
    const uint32_t o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o < n)
        owners[o].margin = m[o2e ? o2e[o] : o];
}

// ---------------------------------------------------------------------------
// Per sphere: world position (fp64) and inflated radii, bin span, sphere-analytical contacts.
// kernel/DEMBinSphereKernels.cu:11-131 (getNumberOfBinsEachSphereTouches) + the analytical half of
// :133-278 (populateBinSphereTouchingPairs), fused: one pass, contacts appended as keys.
// ---------------------------------------------------------------------------
struct ObjWorld {
    double x, y, z;
    float dx, dy, dz;
    float size1, nsign, margin;
    uint32_t type, family;
};

__global__ __launch_bounds__(256) void k_sphere_prep(const DevParams p, const OwnerRec* __restrict__ owners,
                                                     const SphereRec* __restrict__ spheres, GeoRec* __restrict__ geo,
                                                     uint4* __restrict__ binLo, uint2* __restrict__ binN,
                                                     uint32_t* __restrict__ counts, const KeyArena ar, DetectCounters* ctr,
                                                     uint16_t* __restrict__ sphFam) {
    __shared__ ObjWorld sObj[64];
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = s < p.nSpheres;
    d3 pos{0, 0, 0};
    double rBin = 0;
    uint32_t fam = 0, myCount = 0;
    bool ghostOnce = false;
    if (valid) {
        const SphereRec sr = load_sphere(spheres, s);
        const OwnerRec o = load_owner(owners, sr.owner);
        const float4 c = p.comp[sr.comp];
        const d3 op = decode_pos(o.voxelID, o.locX, o.locY, o.locZ, p);
        const f3 rel = rot_apply(rot_coeffs(o.qw, o.qx, o.qy, o.qz), mk3(c.x, c.y, c.z));
        pos = {op.x + (double)rel.x, op.y + (double)rel.y, op.z + (double)rel.z};
        rBin = (double)c.w;
        rBin += o.margin;  // fp64 sum, DEMBinSphereKernels.cu:36
        float rSweep = c.w;
        rSweep += o.margin;  // fp32 sum, DEMContactKernels_SphereSphere.cu:40
        fam = fam_of(o.family);
        ghostOnce = ghost_of(o.family) && (p.hasGhosts & 2u);  // (its own rank lists a ghost sphere's wall contacts)
        GeoRec g;
        // The owner number the sweeps see is the CALLER's (engine-side order, deme_order.inc): it serves their same-owner test and
        // decides which sphere of a pair is A -- spheres are clump-major in the caller's numbering too.  The family word they need
        // when masks, margins or ghosts are in play travels per sphere (the sweeps have no use for the engine's owner slot then).
        g.x = pos.x, g.y = pos.y, g.z = pos.z, g.r = rSweep, g.owner = p.o2e ? p.o2e[sr.owner] : sr.owner;
        geo[s] = g;
        if (!(p.familyTrivial && !p.hasGhosts))
            sphFam[s] = (uint16_t)o.family;
        uint32_t lx, hx, ly, hy, lz, hz;
        bin_range(pos.x, rBin, p.binSize, p.nbX, lx, hx);
        bin_range(pos.y, rBin, p.binSize, p.nbY, ly, hy);
        bin_range(pos.z, rBin, p.binSize, p.nbZ, lz, hz);
        const uint32_t nx = hx - lx + 1, ny = hy - ly + 1, nz = hz - lz + 1;
        binLo[s] = make_uint4(lx, ly, lz, nx);
        binN[s] = make_uint2(ny, nz);
        const uint64_t n64 = (uint64_t)nx * ny * nz;  // the product itself may not fit 32 bits (huge margin, tiny bins)
        if (n64 > DEME_MAX_BINS_PER_SPHERE)
            atomicOr(&ctr->status, DEME_ST_INCIDENCE);
        myCount = (n64 > DEME_MAX_BINS_PER_SPHERE) ? 0u : (uint32_t)n64;
        counts[s] = myCount;
    }
    // largest per-sphere count, recorded only when it is large enough for nSpheres such counts to overflow the 32-bit offsets
    // of the scan (never in a sanely binned scene: no atomic, no extra load on the normal path); the host then checks the total
    if (myCount > 0xFFFFFFFFu / max(p.nSpheres, 1u))
        atomicMax(&ctr->maxCount, myCount);
    if (blockIdx.x == 0 && threadIdx.x == 0)
        counts[p.nSpheres] = 0;  // scan sentinel: offsets[nSpheres] = total

    for (uint32_t ob0 = 0; ob0 < p.nAnal; ob0 += 64) {
        __syncthreads();
        if (threadIdx.x < 64 && ob0 + threadIdx.x < p.nAnal) {
            const AnalObj ob = p.anal[ob0 + threadIdx.x];
            const OwnerRec o = load_owner(owners, ob.owner);
            const d3 op = decode_pos(o.voxelID, o.locX, o.locY, o.locZ, p);
            const RotM m = rot_coeffs(o.qw, o.qx, o.qy, o.qz);
            const f3 rp = rot_apply(m, mk3(ob.relx, ob.rely, ob.relz));
            const f3 rd = rot_apply(m, mk3(ob.rotx, ob.roty, ob.rotz));
            ObjWorld w;
            w.x = op.x + (double)rp.x, w.y = op.y + (double)rp.y, w.z = op.z + (double)rp.z;
            w.dx = rd.x, w.dy = rd.y, w.dz = rd.z;
            w.size1 = ob.size1, w.nsign = ob.normal, w.margin = o.margin;
            w.type = ob.type, w.family = fam_of(o.family);
            sObj[threadIdx.x] = w;
        }
        __syncthreads();
        {
            // every lane of the wavefront runs the loop (an idle one never hits): the hits of one object are appended with ONE
            // reservation per wavefront -- the spheres along a wall are neighbours in the list, and same-address atomics serialise
            // at ~10 ns each (one per sphere-wall contact made this kernel three times as long as its memory traffic)
            const uint32_t nHere = min(64u, p.nAnal - ob0);
            const uint32_t lane = threadIdx.x & 63u;
            for (uint32_t k = 0; k < nHere; k++) {
                const ObjWorld w = sObj[k];
                bool hit = valid && !ghostOnce;
                float thres = 0.f;
                if (hit && !p.familyTrivial) {
                    if (p.familyMasks[mask_pair(fam, w.family)] != 0)
                        hit = false;
                    const float ea = p.familyExtra[fam], eb = p.familyExtra[w.family];
                    thres = (ea < eb) ? ea : eb;
                }
                if (hit) {
                    d3 cp;
                    f3 nr;
                    double depth;
                    const uint32_t t = sphere_entity(pos, (float)rBin, w.type, {w.x, w.y, w.z}, mk3(w.dx, w.dy, w.dz),
                                                     w.size1, w.nsign, w.margin, cp, nr, depth);
                    hit = t && depth > (double)thres;
                }
                const unsigned long long m = __ballot(hit);
                if (m) {
                    const int leader = __ffsll((long long)m) - 1;
                    unsigned long long slot0 = 0;
                    if ((int)lane == leader)
                        slot0 = arena_reserve(ar, arena_seg(ar), (unsigned long long)__popcll(m));
                    slot0 = __shfl(slot0, leader);
                    const unsigned long long slot = slot0 + (unsigned long long)__popcll(m & ((1ull << lane) - 1ull));
                    if (hit)
                        arena_store(ar, arena_seg(ar), slot, make_key(DEME_KEY_CLASS_SA, s, ob0 + k));
                }
            }
        }
    }
}

// kernel/DEMBinSphereKernels.cu:181-211: z-y-x loop nest, sphere-major output
__global__ __launch_bounds__(256) void k_fill_incidence(const DevParams p, const uint4* __restrict__ binLo,
                                                        const uint2* __restrict__ binN,
                                                        const uint32_t* __restrict__ offsets,
                                                        uint32_t* __restrict__ keys, uint32_t* __restrict__ vals,
                                                        uint64_t cap) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= p.nSpheres)
        return;
    const uint4 lo = binLo[s];
    const uint2 n = binN[s];
    if ((uint64_t)lo.w * n.x * n.y > DEME_MAX_BINS_PER_SPHERE)
        return;  // (counted as zero and flagged by k_sphere_prep; the host stops the detection when it reads the flag)
    uint64_t off = offsets[s];
    for (uint32_t k = lo.z; k < lo.z + n.y; k++)
        for (uint32_t j = lo.y; j < lo.y + n.x; j++)
            for (uint32_t i = lo.x; i < lo.x + lo.w; i++) {
                if (off < cap) {
                    keys[off] = i + j * p.nbX + k * p.nbX * p.nbY;  // binIDFrom3Indices, DEMHelperKernels.cuh:339-347
                    vals[off] = s;
                }
                off++;
            }
}

// ---------------------------------------------------------------------------
// Bin sweep.  The bin-sorted incidence list is cut into windows of SW_T entries; the workgroup of window
// w owns every bin whose FIRST entry lies in the window (so a bin is never split between workgroups and
// no look-back through global memory is needed).  The owned range [start, end) holds at most 2*SW_T
// entries and is staged in LDS (two entries per thread).  Inside LDS a bin is the segment [s, e) found
// by binary search over the sorted keys; its n(n-1)/2 pairs are spread evenly over its n entries by
// cyclic pairing: entry k tests partners k+1 .. k+floor((n-1)/2) (mod n), plus k+n/2 for k < n/2 when n
// is even -- every unordered pair exactly once, equal work for all lanes of a bin.  A bin with more than
// SW_T entries (only possible with very coarse bins) is handled by its owner in SW_T x SW_T tiles.
// Decision arithmetic: DEMContactKernels_SphereSphere.cu:57-89 (calcContactPoint) and :177-216 -- the pair
// counts iff the contact point's bin is this bin -- with the cheap fp64 distance test first.
// Output: wavefront ballot + prefix compaction into an LDS buffer, one global reservation per flush,
// coalesced copy-out (replaces the count -> scan -> fill double sweep of the reference).
// ---------------------------------------------------------------------------
#ifndef SW_T
#define SW_T 256
#endif
#ifndef SW_WPB
#define SW_WPB 1  // windows per workgroup (measured: 1 -> 2.30 ms, 4 -> 2.35 ms, 16 -> 2.51 ms per detection)
#endif
#define SW_W (2 * SW_T)
#define SW_OUT 256  // with the rest of SweepLDS: 19.6 KB, eight workgroups per CU
#define SW_FLUSH 128

struct SweepLDS {
    float4 f[SW_W];  // fp32 copy for the pair loop's pre-filter: position relative to the range's first entry, inflated radius
    uint32_t owner[SW_W], sph[SW_W], bin[SW_W];
    uint16_t fam[SW_W];  // family word: family | ghost bit (9 bits)
    uint32_t queue[SW_T / 64][128];  // per wavefront: pairs that passed the distance test, waiting for the exact test
    uint64_t out[SW_OUT];
    unsigned long long gBase;
    unsigned long long headMask[SW_W / 64];  // per 64 entries of the range: which of them are first entries of a bin
    uint32_t nOut;
    uint32_t start, endIdx, giant;
    uint32_t pop;  // largest bin population this workgroup has seen
    uint32_t nBins, nextBin;
    uint16_t binStart[SW_W + 2];  // first entry of the b-th bin of the range; binStart[nBins] = end of the range
};

// The cheap half of pair_test: different owners and centres closer than the sum of the inflated radii.
__device__ inline bool pair_near(double ax, double ay, double az, float ar, uint32_t ao, double bx, double by, double bz, float br,
                                 uint32_t bo) {
    if (ao == bo)
        return false;
    const double rA = (double)ar, rB = (double)br;
    const double d2 = (ax - bx) * (ax - bx) + (ay - by) * (ay - by) + (az - bz) * (az - bz);
    return !(d2 > (rA + rB) * (rA + rB));
}

// fp32 pre-filter of the pair loop: conservative (never rejects a pair the fp64 test accepts).  Positions are relative to the
// corner of the bin the pair is tested in, so their fp32 rounding is ~1e-9 m for millimetre-to-centimetre bins.  The slack is a
// sum of per-entry terms, so each entry carries its share in w (sweep_radius): 0.5e-6 m plus the fp32 rounding its coordinates can
// carry at their magnitude (ulp = 1.2e-7 x |value|; taken 3x) -- conservative for metre-sized bins and bodies as well -- times
// (1 + 1e-5) for the rounding of the test itself.
__device__ inline float sweep_radius(float x, float y, float z, float r) {
    const float mag = fabsf(x) + fabsf(y) + fabsf(z) + r;
    return (r + 0.5e-6f + 4e-7f * mag) * 1.00001f;
}
__device__ inline bool pair_near_f(float4 a, uint32_t ao, float4 b, uint32_t bo) {
    const float dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
    const float rs = a.w + b.w;
    return (ao != bo) & (dx * dx + dy * dy + dz * dz <= rs * rs);
}

__device__ inline bool pair_test(const DevParams& p, double ax, double ay, double az, float ar, uint32_t ao,
                                 uint32_t af, double bx, double by, double bz, float br, uint32_t bo, uint32_t bf,
                                 uint32_t bin) {
    if (ao == bo)
        return false;
    // distance test first (the reference evaluates everything and ANDs the flags: same outcome)
    const double rA = (double)ar, rB = (double)br;
    const double d2 = (ax - bx) * (ax - bx) + (ay - by) * (ay - by) + (az - bz) * (az - bz);
    if (d2 > (rA + rB) * (rA + rB))
        return false;
    float am = 0.f;
    if (ghost_of(af) && ghost_of(bf))
        return false;  // both are copies of clumps other ranks own: the pair is theirs
    if ((passive_of(af) && !ghost_of(bf)) || (passive_of(bf) && !ghost_of(af)))
        return false;  // an own clump against a passive ghost: the ghost's rank evaluates the pair and sends the reaction
    if (!p.familyTrivial) {
        af = fam_of(af), bf = fam_of(bf);
        if (p.familyMasks[mask_pair(af, bf)] != 0)
            return false;
        const float ea = p.familyExtra[af], eb = p.familyExtra[bf];
        am = (ea < eb) ? ea : eb;
    }
    d3 cp;
    f3 n;
    double depth;
    bool in = spheres_overlap(ax, ay, az, rA, bx, by, bz, rB, cp, n, depth);
    in = in && (depth > (double)am);
    return in && (point_bin(cp.x, cp.y, cp.z, p) == bin);
}

// wave-wide append of `key` for lanes with `hit` (all lanes of the wave must call this together)
__device__ inline void sweep_emit(SweepLDS& L, bool hit, uint64_t key, uint32_t lane, const KeyArena& ar) {
    const unsigned long long m = __ballot(hit);
    if (!m)
        return;
    const int leader = __ffsll((long long)m) - 1;
    uint32_t pos0 = 0;
    if ((int)lane == leader)
        pos0 = atomicAdd(&L.nOut, (uint32_t)__popcll(m));
    pos0 = __shfl(pos0, leader);
    if (hit) {
        const uint32_t pos = pos0 + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        if (pos < SW_OUT) {
            L.out[pos] = key;
        } else {  // LDS buffer full: spill straight to global
            arena_store(ar, arena_seg(ar), arena_reserve(ar, arena_seg(ar), 1ull), key);
        }
    }
}

// Which sphere of a pair is A: the one with the smaller id IN THE CALLER'S NUMBERING (the reference's i < j loop runs over its
// own numbering, and the choice matters: the contact point is computed from B's side).  When the engine keeps the spheres in a
// spatial order of its own (DevParams::o2e, deme_order.inc) the sweep stages the CALLER's owner numbers in LDS and takes the roles
// from them (spheres are clump-major in the caller's numbering too, and a pair of one owner is no pair), so that lists, contact
// points and histories are those of the caller's numbering whatever the internal one is.

// all lanes of one wavefront: exact test of the first `cnt` queued pairs (entry i | entry q << 16) and emission of the hits
__device__ inline void sweep_confirm(const DevParams& p, SweepLDS& L, const GeoRec* __restrict__ geo, const uint32_t* wq, uint32_t cnt,
                                     uint32_t lane, const KeyArena& ar) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // LDS operations of one wavefront complete in order: only the
    __builtin_amdgcn_wave_barrier();                         // compiler has to be kept from reordering
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    bool hit = false;
    uint64_t key = 0;
    if (lane < cnt) {
        const uint32_t e = wq[lane];
        // A is the entry that comes first in the bin's list (= the smaller sphere id: the list is stably sorted), as in the
        // reference's i < j loop.  The order matters: the contact point is computed from B's side, and a point within
        // rounding of a bin face must fall on the same side in every bin that tests the pair -- the cyclic pairing meets
        // the two entries in either order, which (measured: once per ~1e9 pair evaluations) dropped or doubled a contact.
        const uint32_t e0 = e & 0xFFFFu, e1 = e >> 16;
        uint32_t i = min(e0, e1), q = max(e0, e1);
        // the fp64 records: only the ~3 % that pass the pre-filter need them
        if (L.owner[i] > L.owner[q]) {  // the staged owner numbers are the caller's: the smaller one's sphere is A (in the caller's own
            const uint32_t t = i;       // order entries i < q never swap: spheres are clump-major)
            i = q, q = t;
        }
        const GeoRec ga = geo[L.sph[i]], gb = geo[L.sph[q]];
        hit = pair_test(p, ga.x, ga.y, ga.z, ga.r, L.owner[i], L.fam[i], gb.x, gb.y, gb.z, gb.r, L.owner[q], L.fam[q], L.bin[q]);
        if (hit)
            key = make_key(DEME_KEY_CLASS_SS, L.sph[i], L.sph[q]);
    }
    sweep_emit(L, hit, key, lane, ar);
}

// block-wide flush of the LDS output buffer (call from uniform control flow)
__device__ inline void sweep_flush(SweepLDS& L, uint32_t t, const KeyArena& ar) {
    const uint32_t nOut = min(L.nOut, (uint32_t)SW_OUT);
    const uint32_t seg = arena_seg(ar);
    if (t == 0)
        L.gBase = nOut ? arena_reserve(ar, seg, (unsigned long long)nOut) : 0ull;
    __syncthreads();
    for (uint32_t q = t; q < nOut; q += SW_T)
        arena_store(ar, seg, L.gBase + q, L.out[q]);
    __syncthreads();
    if (t == 0)
        L.nOut = 0;
    __syncthreads();
}

__global__ __launch_bounds__(SW_T) void k_sweep(const DevParams p, const uint32_t* __restrict__ keys,
                                                const uint32_t* __restrict__ sphIds, uint32_t P,
                                                const GeoRec* __restrict__ geo, const uint16_t* __restrict__ sphFam,
                                                const KeyArena ar, uint2* __restrict__ winStats) {
    __shared__ SweepLDS L;
    const uint32_t t = threadIdx.x;
    const uint32_t lane = t & 63u;
    const uint32_t nWin = (P + SW_T - 1) / SW_T;
    if (t == 0)
        L.nOut = 0;
    for (uint32_t win = blockIdx.x; win < nWin; win += gridDim.x) {
        const uint32_t base = win * SW_T;
        __syncthreads();  // everyone is done with the previous window's LDS contents
        // ---- keys of [base, base + 2*SW_T) into LDS; heads mark first entries of bins
        uint32_t k0 = DEME_NULL_BINID_DEV, k1 = DEME_NULL_BINID_DEV;
        if (base + t < P)
            k0 = keys[base + t];
        if (base + SW_T + t < P)
            k1 = keys[base + SW_T + t];
        L.bin[t] = k0;
        L.bin[SW_T + t] = k1;
        if (t == 0) {
            L.start = SW_W;
            L.endIdx = SW_W;
            L.giant = 0;
            L.pop = 0;
        }
        __syncthreads();
        const uint32_t prev0 = (t == 0) ? (base == 0 ? DEME_NULL_BINID_DEV : keys[base - 1]) : L.bin[t - 1];
        const bool head0 = (base + t < P) && (k0 != prev0);
        const bool head1 = (base + SW_T + t >= P) ? (base + SW_T + t == P) : (k1 != L.bin[SW_T + t - 1]);
        if (head0)
            atomicMin(&L.start, t);  // first bin that begins in my window
        if (head1)
            atomicMin(&L.endIdx, SW_T + t);  // first bin (or list end) at or after the next window
        if (base + t == P)
            atomicMin(&L.endIdx, t);  // the list ends inside my own window
        __syncthreads();
        const uint32_t start = L.start;
        if (start >= SW_T) {  // no bin begins here: an earlier workgroup owns everything in this window
            if (t == 0)
                winStats[win] = make_uint2(0u, 0u);
            continue;
        }
        uint32_t end = L.endIdx;
        uint32_t giantStart = SW_W;
        if (end >= SW_W) {
            // the last bin that begins in my window runs past the LDS window: a giant bin.  Bins before it
            // are complete and handled in LDS; the giant one goes through the tiled path below.
            // its first entry = the last head inside [start, SW_T): found with a max-reduction
            if (head0)
                atomicMax(&L.giant, t);
            __syncthreads();
            giantStart = L.giant;
            end = giantStart;
        }
        const uint32_t n_rng = end - start;  // <= 2*SW_T - 1 entries, all complete bins
        // ---- stage geometry of my range (two entries per thread)
        __syncthreads();
        for (uint32_t q = t; q < n_rng; q += SW_T) {
            const uint32_t sph = sphIds[base + start + q];
            const GeoRec g = geo[sph];
            // fp32 copy relative to the corner of the entry's OWN bin (pairs are only formed inside a bin, so both partners share
            // the origin): magnitudes stay below a bin edge plus a radius whatever the size of the domain
            const uint32_t bq = keys[base + start + q];
            const uint32_t nbXY = p.nbX * p.nbY;
            const uint32_t iz = fast_div(bq, nbXY, p.mNbXY), rem = bq - iz * nbXY;
            const uint32_t iy = fast_div(rem, p.nbX, p.mNbX), ix = rem - iy * p.nbX;
            const float fx = (float)(g.x - (double)ix * p.binSize), fy = (float)(g.y - (double)iy * p.binSize),
                        fz = (float)(g.z - (double)iz * p.binSize);
            L.f[q] = make_float4(fx, fy, fz, sweep_radius(fx, fy, fz, g.r));
            // (the owner number of the geometry record is the caller's: k_sphere_prep.  Looking it up here -- in this loop or in a
            // pass of its own -- cost 45 us of the kernel's 340 even with no order to translate: measured, profiles/r04)
            L.owner[q] = g.owner, L.sph[q] = sph;
            L.fam[q] = (uint16_t)((p.familyTrivial && !p.hasGhosts) ? 0u : sphFam[sph]);
        }
        // keys were loaded at offset `start`: shift so that L.bin[q] matches entry q of the range
        uint32_t b0 = DEME_NULL_BINID_DEV, b1 = DEME_NULL_BINID_DEV;
        if (t < n_rng)
            b0 = L.bin[start + t];
        if (SW_T + t < n_rng)
            b1 = L.bin[start + SW_T + t];
        __syncthreads();
        L.bin[t] = b0;
        L.bin[SW_T + t] = b1;
        __syncthreads();
        // ---- bins of the range: entry q is a head if its key differs from its left neighbour's; one ballot per 64 entries
        for (uint32_t q = t; q < SW_W; q += SW_T) {
            const bool head = q < n_rng && (q == 0 || L.bin[q] != L.bin[q - 1]);
            const unsigned long long hm = __ballot(head);
            if (lane == 0)
                L.headMask[q >> 6] = hm;
        }
        __syncthreads();
        // ---- list of the range's bins: the b-th head begins bin b
        for (uint32_t q = t; q < SW_W; q += SW_T) {
            const unsigned long long hm = L.headMask[q >> 6];
            if ((hm >> lane) & 1ull) {
                uint32_t idx = (uint32_t)__popcll(hm & ((1ull << lane) - 1ull));
                for (uint32_t ch = 0; ch < (q >> 6); ch++)
                    idx += (uint32_t)__popcll(L.headMask[ch]);
                L.binStart[idx] = (uint16_t)q;
            }
        }
        if (t == 0) {
            uint32_t nb = 0;
            for (uint32_t ch = 0; ch < SW_W / 64; ch++)
                nb += (uint32_t)__popcll(L.headMask[ch]);
            L.binStart[nb] = (uint16_t)n_rng;
            L.nBins = nb;
            L.nextBin = 0;
        }
        __syncthreads();
        // ---- a wavefront per bin, a lane per PAIR.  The n(n-1)/2 pairs of a bin of n entries are numbered f = (m-1) n + k: entry k
        // with the entry m places further on (cyclically), m = 1 .. (n-1)/2, and for even n the first n/2 entries once more with
        // m = n/2 -- every unordered pair exactly once.  Lane l takes f = l, l + 64, ...; k and m advance by 64 mod n and 64 div n.
        // (The first version gave every ENTRY a lane and ran the wavefront for the trips of its largest bin: with four or five
        // bins of different sizes per wavefront two thirds of the lane-trips were idle, and the kernel is bound by instruction issue.)
        // Bins are handed out through an LDS counter: their work goes with n^2.
        uint32_t* wq = L.queue[t >> 6];
        uint32_t qn = 0;  // entries in my wavefront's queue (wave-uniform)
        uint32_t popMine = 0;
        const uint32_t nBinsHere = L.nBins;
        // Two phases (only ~3 % of the pairs of a bin pass the distance test, but with 64 lanes almost every trip has a survivor):
        // the loop runs the fp32 distance test and queues the survivors per wavefront; the exact test -- sqrt, divisions,
        // contact-point bin -- then runs on full wavefronts of survivors.
        auto enqueue = [&](bool near, uint32_t packed) {
            const unsigned long long nm = __ballot(near);
            if (nm) {
                if (near)
                    wq[qn + (uint32_t)__popcll(nm & ((1ull << lane) - 1ull))] = packed;
                qn += (uint32_t)__popcll(nm);  // wave-uniform
                if (qn >= 64u) {
                    sweep_confirm(p, L, geo, wq, 64u, lane, ar);
                    const uint32_t rest = qn - 64u;
                    uint32_t carry = 0;
                    if (lane < rest)
                        carry = wq[64u + lane];
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    if (lane < rest)
                        wq[lane] = carry;
                    qn = rest;
                }
            }
        };
        for (;;) {
            uint32_t b = 0;
            if (lane == 0)
                b = atomicAdd(&L.nextBin, 1u);
            b = (uint32_t)__builtin_amdgcn_readfirstlane((int)b);
            if (b >= nBinsHere)
                break;
            const uint32_t s = (uint32_t)__builtin_amdgcn_readfirstlane((int)L.binStart[b]);
            const uint32_t n = (uint32_t)__builtin_amdgcn_readfirstlane((int)L.binStart[b + 1]) - s;
            popMine = max(popMine, n);
            if (n < 2u)
                continue;
            const uint32_t total = n * (n - 1u) / 2u;
            // lane / n and 64 / n through the reciprocal: (x + 0.5) / n is at least 0.5 / n away from an integer, far outside the
            // rounding of the product for x <= 64, n < 2 SW_T
            const float rn = __builtin_amdgcn_rcpf((float)n);  // (1 ulp)
            uint32_t m = (uint32_t)(((float)lane + 0.5f) * rn);
            uint32_t k = lane - m * n;
            m += 1u;
            const uint32_t c2 = (uint32_t)(64.5f * rn), c1 = 64u - c2 * n;
            for (uint32_t f = lane, f0 = 0; f0 < total; f0 += 64u, f += 64u) {
                const bool act = f < total;
                uint32_t j = k + m;
                j = min(j, j - n);  // (j - n wraps to a huge value when j < n)
                const uint32_t ia = s + k, ib = act ? s + j : s + k;  // an idle lane pairs an entry with itself: same owner, never near
                const float4 fa = L.f[ia], fb = L.f[ib];
                const uint32_t oa = L.owner[ia], ob = L.owner[ib];
                enqueue(act & pair_near_f(fa, oa, fb, ob), ia | (ib << 16));  // fp32, conservative; the exact fp64 test follows in phase 2
                k += c1, m += c2;
                if (k >= n)
                    k -= n, m += 1u;
            }
        }
        if (qn) {
            sweep_confirm(p, L, geo, wq, qn, lane, ar);
            qn = 0;
        }
        if (lane == 0 && popMine)
            atomicMax(&L.pop, popMine);
        uint32_t giantPop = 0;
        // ---- giant bin: SW_T x SW_T tiles, A tile in LDS, each thread holds one B entry
        if (giantStart < SW_W) {
            const uint32_t gs = base + giantStart;
            const uint32_t gbin = keys[gs];
            uint32_t lo = gs + 1, hi = P;  // end of the giant bin: first index with key > gbin
            while (lo < hi) {
                const uint32_t mid = lo + ((hi - lo) >> 1);
                if (keys[mid] <= gbin)
                    lo = mid + 1;
                else
                    hi = mid;
            }
            const uint32_t ge = lo;
            giantPop = ge - gs;
            for (uint32_t ta = gs; ta < ge; ta += SW_T) {
                __syncthreads();
                const uint32_t na = min((uint32_t)SW_T, ge - ta);
                if (t < na) {
                    const uint32_t sph = sphIds[ta + t];
                    const GeoRec g = geo[sph];
                    L.owner[t] = g.owner, L.sph[t] = sph;
                    L.fam[t] = (uint16_t)((p.familyTrivial && !p.hasGhosts) ? 0u : sphFam[sph]);
                }
                __syncthreads();
                for (uint32_t tb = ta; tb < ge; tb += SW_T) {
                    const uint32_t jb = tb + t;
                    const bool valid = jb < ge;
                    double mx = 0, my = 0, mz = 0;
                    float mr = 0;
                    uint32_t mo = 0, mf = 0, ms = 0;
                    if (valid) {
                        ms = sphIds[jb];
                        const GeoRec g = geo[ms];
                        mx = g.x, my = g.y, mz = g.z, mr = g.r, mo = g.owner;
                        mf = (p.familyTrivial && !p.hasGhosts) ? 0u : sphFam[ms];
                    }
                    for (uint32_t i = 0; i < na; i++) {  // uniform trip count
                        const bool act = valid && (ta + i < jb);  // each unordered pair once
                        bool hit = false;
                        uint64_t key = 0;
                        if (act) {
                            const GeoRec ga = geo[L.sph[i]];  // same address for every lane
                            // A is the sphere of the owner with the smaller number in the caller's numbering (the staged numbers): the
                            // entry, unless an engine-side order put the lane's sphere first -- ONE test with the arguments in place
                            // (two calls cost the whole kernel registers: its hot loop spilled)
                            const bool entryFirst = L.owner[i] <= mo;
                            const double ax = entryFirst ? ga.x : mx, ay = entryFirst ? ga.y : my, az = entryFirst ? ga.z : mz;
                            const double bx = entryFirst ? mx : ga.x, by = entryFirst ? my : ga.y, bz = entryFirst ? mz : ga.z;
                            const float ar_ = entryFirst ? ga.r : mr, br_ = entryFirst ? mr : ga.r;
                            const uint32_t ao = entryFirst ? L.owner[i] : mo, bo = entryFirst ? mo : L.owner[i];
                            const uint32_t af = entryFirst ? (uint32_t)L.fam[i] : mf, bf = entryFirst ? mf : (uint32_t)L.fam[i];
                            hit = pair_test(p, ax, ay, az, ar_, ao, af, bx, by, bz, br_, bo, bf, gbin);
                            if (hit)
                                key = entryFirst ? make_key(DEME_KEY_CLASS_SS, L.sph[i], ms) : make_key(DEME_KEY_CLASS_SS, ms, L.sph[i]);
                        }
                        sweep_emit(L, hit, key, lane, ar);
                    }
                    __syncthreads();
                    if (L.nOut >= SW_FLUSH)
                        sweep_flush(L, t, ar);
                }
            }
        }
        __syncthreads();
        if (t == 0) {  // active bins and the largest population among the bins this window owns (k_bin_stats_final adds them up)
            uint32_t heads = (giantStart < SW_W) ? 1u : 0u;
            for (uint32_t ch = 0; ch < SW_W / 64; ch++)
                heads += (uint32_t)__popcll(L.headMask[ch]);
            winStats[win] = make_uint2(heads, max(L.pop, giantPop));
        }
        if (L.nOut >= SW_FLUSH)  // uniform: read after the barrier
            sweep_flush(L, t, ar);
    }
    __syncthreads();
    sweep_flush(L, t, ar);
}

// Active-bin count and the largest bin population (numSpheresBinTouches statistics of DEMCubContactDetection.cu:195-230; the
// population check feeds errOutBinSphNum): k_sweep leaves one (bins, largest population) record per window, added up here.
__global__ __launch_bounds__(256) void k_bin_stats_final(const uint2* __restrict__ perBlock, uint32_t nBlocks, DetectCounters* ctr) {
    __shared__ uint32_t sHeads[4], sPop[4];
    uint32_t heads = 0, pop = 0;
    // a workgroup per 2048 records, eight independent loads in flight per thread; one pair of atomics per workgroup (a few dozen
    // in all: same-address atomics serialise at ~12 ns each)
    {
        const uint32_t i0 = blockIdx.x * 2048u;
        uint2 v[8];
#pragma unroll
        for (uint32_t k = 0; k < 8; k++) {
            const uint32_t i = i0 + k * 256u + threadIdx.x;
            v[k] = (i < nBlocks) ? perBlock[i] : make_uint2(0u, 0u);
        }
#pragma unroll
        for (uint32_t k = 0; k < 8; k++) {
            heads += v[k].x;
            pop = max(pop, v[k].y);
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        heads += (uint32_t)__shfl_xor((int)heads, off);
        pop = max(pop, (uint32_t)__shfl_xor((int)pop, off));
    }
    if ((threadIdx.x & 63u) == 0) {
        sHeads[threadIdx.x >> 6] = heads;
        sPop[threadIdx.x >> 6] = pop;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(&ctr->nActiveBins, sHeads[0] + sHeads[1] + sHeads[2] + sHeads[3]);
        const uint32_t m = max(max(sPop[0], sPop[1]), max(sPop[2], sPop[3]));
        if (m > 1)
            atomicMax(&ctr->maxInBin, m);
    }
}

// Second half of the contact-key sort.  The radix sort orders the keys by their upper part only (sphere A, type class: the
// occupied bits above bit 31); a segment of equal (A, class) holds a sphere's handful of partners in arbitrary order.  Every
// key finds its segment by walking to its ends and takes the number of keys that go before it as its rank (ties by position) -- a
// few compares per key instead of three more radix passes over the list.
__global__ __launch_bounds__(256) void k_segment_rank_sort(uint32_t n, const uint64_t* __restrict__ in, uint64_t* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const uint64_t k = in[i], seg = k >> 31;
    uint32_t s = i, rank = 0;
    while (s > 0) {
        const uint64_t q = in[s - 1];
        if ((q >> 31) != seg)
            break;
        rank += (q <= k) ? 1u : 0u;  // an equal key further left goes first (persistent marks repeat keys the sweep found too)
        s--;
    }
    for (uint32_t j = i + 1; j < n; j++) {
        const uint64_t q = in[j];
        if ((q >> 31) != seg)
            break;
        rank += (q < k) ? 1u : 0u;
    }
    out[s + rank] = k;
}

// ---------------------------------------------------------------------------
// History map + wildcard migration.  Semantics of kernel/DEMHistoryMappingKernels.cu:17-61
// ("same (A, B, type) as in the previous list") and kernel/DEMPrepForceKernels.cu:46-68
// (rearrangeContactWildcards); the per-sphere linear search becomes a binary search because both
// lists are key-sorted.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_history(uint32_t nNew, const uint64_t* __restrict__ newKeys, uint32_t nPrev,
                                                 const uint64_t* __restrict__ prevKeys, uint32_t* __restrict__ mapping) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nNew)
        return;
    const uint64_t k = newKeys[c];
    // both lists are sorted and mostly the same pairs: contact c of the new list sits near c * nPrev / nNew in the previous one.
    // Gallop outwards from that guess to bracket the key, then bisect the bracket (a handful of probes, neighbouring lanes on
    // neighbouring addresses, instead of log2(nPrev) dependent probes across the whole list)
    uint32_t lo = 0, hi = nPrev;  // invariant: the first index with prevKeys >= k lies in [lo, hi]
    if (nPrev) {
        uint32_t g = (uint32_t)(((uint64_t)c * nPrev) / nNew);
        g = min(g, nPrev - 1);
        if (prevKeys[g] < k) {
            lo = g + 1;
            uint32_t step = 8;
            while (lo < nPrev) {
                const uint32_t j = (lo + step < nPrev) ? lo + step : nPrev - 1;
                if (prevKeys[j] < k) {
                    lo = j + 1;
                    step <<= 2;
                } else {
                    hi = j;
                    break;
                }
            }
        } else {
            hi = g;
            uint32_t step = 8;
            while (hi > 0) {
                const uint32_t j = (hi > step) ? hi - step : 0;
                if (prevKeys[j] < k) {
                    lo = j + 1;
                    break;
                }
                hi = j;
                step <<= 2;
            }
        }
    }
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (prevKeys[mid] < k)
            lo = mid + 1;
        else
            hi = mid;
    }
    mapping[c] = (lo < nPrev && prevKeys[lo] == k) ? lo : 0xFFFFFFFFu;
}

__global__ __launch_bounds__(256) void k_migrate(uint32_t nNew, uint32_t nW, const uint32_t* __restrict__ mapping,
                                                 uint32_t nPrevStored, const float* __restrict__ oldW,
                                                 float* __restrict__ newW) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nNew)
        return;
    const uint32_t m = mapping[c];
    const bool have = (m != 0xFFFFFFFFu) && (m < nPrevStored);
    if (nW == 4) {
        float4 v = make_float4(0, 0, 0, 0);
        if (have)
            v = reinterpret_cast<const float4*>(oldW)[m];
        reinterpret_cast<float4*>(newW)[c] = v;
    } else {
        for (uint32_t w = 0; w < nW; w++)
            newW[(size_t)c * nW + w] = have ? oldW[(size_t)m * nW + w] : 0.f;
    }
}

// ---------------------------------------------------------------------------
// Integration.  kernel/DEMIntegrationKernels.cu:100-264 (integrateVelPos / integrateOwners) with the
// pass-on of IntegrationVelPassOn*.cu; fixed families follow SetFamilyFixed (APIPublic.cpp:980-1011).
// FUSED variant: the owner's a/alpha are gathered from the per-contact contributions inside the same
// kernel (no clear pass, no atomics, no accumulator round trip: replaces prepareAccArrays +
// forceToAcc + integrateOwners, DEMPrepForceKernels.cu:32, DEMCollectForceKernels_Compact.cu:13).
// ---------------------------------------------------------------------------
// Per-owner gather of the per-contact contributions written by k_calc_forces (atomics-free, and in
// a fixed order -- A-side run, then B-side list, both by ascending contact index -- so the fp32 sum is
// reproducible run to run; it differs from a pure list-order sum only in rounding).
// An owner's A-side contacts are the contiguous run [aStart[o], aStart[o+1]); its B-side contacts are
// bIdx[bStart[o] .. bStart[o+1]) (ascending).
struct GatherArgs {
    const uint32_t* aStart;  // nOwners+1
    const uint32_t* bStart;  // nOwners+1
    const uint32_t* bIdx;    // contact indices sorted by B's owner (stable)
    const uint8_t* heavy;    // 1: too many contacts for one thread; summed by k_reduce_heavy into acc
    const float4* conA4;     // per-contact A-side records: only valid for runs that straddle a force-kernel block
    const float2* conA2;
    const float4* conB4;
    const float2* conB2;
    const float4* aSum;      // in-order sum of an A run that lies inside one force-kernel block (deme_force.h)
    const AccRec* nextAcc;   // null, or per owner: acceleration the script added for this step (deme_add_owner_acc)
    // fast arithmetic mode: the contributions are world-frame forces and torques (ForceArgs::world); their per-owner sums
    // become a and alpha through acc_from_world
    uint32_t world;
    // owner-tile form of the force pass (deme_tile.h): aSum holds, for EVERY owner, the sum over the contacts its tile evaluated
    // (A and B sides); bStart / bIdx list only the contacts whose B owner lives in another tile than A's, with records in conB
    uint32_t tile;
    const float4* rec32;     // tile form: the crossing contacts' B-side records, 32 bytes each, dense; bIdx holds record numbers
    // one evaluation per cross-cut contact (deme_halo_group_set_cross_contacts): the neighbour that evaluated an own clump's
    // contacts with ITS clumps sends a / alpha of their sum; revSlot[o] = the clump's place in that message, 0xFFFFFFFF = none
    const uint32_t* revSlot;
    const float4* revAcc;    // two float4 per place
    uint32_t revPhase;  // k_integrate: 1 = leave the owners that wait for a reverse share (revSlot set) to k_integrate_list
    uint32_t revInIntegrator;  // k_reduce_heavy: 1 = launched by the stepping loop (integrate_owner adds the share afterwards)
};

// one B-side record for the per-owner gather: by contact index from the two per-contact arrays, or (tile form) by record number
__device__ inline void gather_b_load(const GatherArgs& g, uint32_t idx, float4& c4, float2& c2) {
    if (g.tile) {
#if DEME_REC24
        const float2* r24 = reinterpret_cast<const float2*>(g.rec32) + 3 * (size_t)idx;  // deme_tile.h: 24-byte records
        const float2 r0 = r24[0], r1 = r24[1];
        c4 = make_float4(r0.x, r0.y, r1.x, r1.y), c2 = r24[2];
#else
        const float4 r0 = g.rec32[2 * (size_t)idx], r1 = g.rec32[2 * (size_t)idx + 1];
        c4 = r0, c2 = make_float2(r1.x, r1.y);
#endif
    } else {
        conb_load(g.conB4, g.conB2, idx, c4, c2);
    }
}

// per-owner conversion of the fast mode: a = F / m, alpha = R^T tau / I (body frame, like the reference's alpha)
__device__ inline void acc_from_world(const DevParams& p, const OwnerRec& r, float4& a, float4& al) {
    const float4 mp = p.massProps[r.inertiaOff];
    const RotM R = rot_coeffs(r.qw, r.qx, r.qy, r.qz);
    const f3 tl = rot_apply(rot_transpose(R), mk3(al.x, al.y, al.z));
    a = make_float4(a.x / mp.x, a.y / mp.x, a.z / mp.x, 0.f);
    al = make_float4(tl.x / mp.y, tl.y / mp.z, tl.z / mp.w, 0.f);
}


// A-side sum of one owner: the force kernel's in-workgroup result when the run lay inside one block, else the
// same in-order sum over the per-contact records (loads issued four at a time).
__device__ inline void a_side_sum(const GatherArgs& g, uint32_t o, uint32_t s, uint32_t e, float& ax, float& ay, float& az,
                                  float& lx, float& ly, float& lz) {
#if DEME_REC24
    if (g.tile) {  // the tile pass leaves six floats per owner (deme_tile.h)
        const float2* t24 = reinterpret_cast<const float2*>(g.aSum) + 3 * (size_t)o;
        const float2 u = t24[0], v = t24[1], w = t24[2];
        ax = u.x, ay = u.y, az = v.x, lx = v.y, ly = w.x, lz = w.y;
        return;
    }
#endif
    if (g.tile || a_run_in_one_block(s, e)) {
        const float4 v = g.aSum[2 * (size_t)o], w = g.aSum[2 * (size_t)o + 1];
        ax = v.x, ay = v.y, az = v.z, lx = w.x, ly = w.y, lz = w.z;
        return;
    }
    for (uint32_t ia = s; ia < e; ia += 4) {
        float4 c4[4];
        float2 c2[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const bool ok = ia + k < e;
            c4[k] = ok ? g.conA4[ia + k] : make_float4(0, 0, 0, 0);
            c2[k] = ok ? g.conA2[ia + k] : make_float2(0, 0);
        }
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (ia + k < e) {
                ax += c4[k].x, ay += c4[k].y, az += c4[k].z;
                lx += c4[k].w, ly += c2[k].x, lz += c2[k].y;
            }
    }
}

__device__ inline void gather_owner(const GatherArgs& g, uint32_t o, float4& a, float4& al) {
    // fixed order: the A-side run (ascending contact index), then the B-side list (ascending contact index).
    // Loads are issued four at a time so that their latencies overlap; the fp32 sum is reproducible run to run.
    float ax = 0.f, ay = 0.f, az = 0.f, lx = 0.f, ly = 0.f, lz = 0.f;
    const uint32_t ib0 = g.bStart[o], eb = g.bStart[o + 1];
    a_side_sum(g, o, g.aStart[o], g.aStart[o + 1], ax, ay, az, lx, ly, lz);
    for (uint32_t ib = ib0; ib < eb; ib += 4) {
        uint32_t idx[4];
        float4 c4[4];
        float2 c2[4];
#pragma unroll
        for (int k = 0; k < 4; k++)
            idx[k] = (ib + k < eb) ? g.bIdx[ib + k] : 0u;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const bool ok = ib + k < eb;
            c4[k] = make_float4(0, 0, 0, 0);
            c2[k] = make_float2(0, 0);
            if (ok)
                gather_b_load(g, idx[k], c4[k], c2[k]);
        }
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (ib + k < eb) {
                ax += c4[k].x, ay += c4[k].y, az += c4[k].z;
                lx += c4[k].w, ly += c2[k].x, lz += c2[k].y;
            }
    }
    a = make_float4(ax, ay, az, 0.f);
    al = make_float4(lx, ly, lz, 0.f);
}

// Workgroup-cooperative form of gather_owner for 256 consecutive owners (the integrator's fused path).
// The A-side runs of consecutive owners are one contiguous range of the contribution arrays, and so are
// their B-side index lists: the range is streamed through LDS in tiles with fully coalesced loads (A side)
// or with one independent gather per lane (B side), and every owner then sums its own slice out of LDS in
// the same order as gather_owner (A run ascending, then B list ascending) -- bit-identical results, but no
// lane walks a chain of dependent global loads any more.  `want` is false for lanes that do not need a sum
// (beyond the end, fixed, ghost, heavy); all lanes of the workgroup must call.
#ifndef DEME_GATHER_TILE
#define DEME_GATHER_TILE 768  // 18 KB: with the 20 KB of the record transposes the integrator keeps 8 workgroups per CU (1024: 6; -4 %)
#endif
struct GatherLds {
    float4 c4[DEME_GATHER_TILE];
    float2 c2[DEME_GATHER_TILE];
};
// What gather_block needs that hangs on nothing but the owner's number: asked for at the very start of the integrator, before its
// owner records are loaded and transposed -- the kernel is a chain of dependent loads per workgroup (bounds -> indices -> records),
// and these were a fourth link behind the records' barrier (42.5 -> 41.7 us; fetching the first tile of records into registers as
// well costs 17 VGPRs and a workgroup per CU: 43.2-44.6 us, not kept)
struct GatherPre {
    uint32_t loB, hiB, sA, eA, sB, eB;
};
__device__ inline GatherPre gather_prefetch(const GatherArgs& g, uint32_t nOwners, uint32_t o, bool valid) {
    GatherPre q;
    const uint32_t oFirst = blockIdx.x * blockDim.x;
    const uint32_t oEnd = (oFirst + blockDim.x < nOwners) ? oFirst + blockDim.x : nOwners;
    q.loB = g.bStart[oFirst], q.hiB = g.bStart[oEnd];
    q.sA = q.eA = q.sB = q.eB = 0;
    if (valid) {
        q.sA = g.aStart[o], q.eA = g.aStart[o + 1];
        q.sB = g.bStart[o], q.eB = g.bStart[o + 1];
    }
    return q;
}
__device__ inline void gather_block(const GatherArgs& g, uint32_t nOwners, uint32_t o, bool want, GatherLds& L, float4& a,
                                    float4& al, const GatherPre& q) {
    const uint32_t t = threadIdx.x;
    const uint32_t loB = q.loB, hiB = q.hiB;
    if (hiB - loB > 8u * DEME_GATHER_TILE) {  // workgroup-uniform: a giant run inside
        if (want)
            gather_owner(g, o, a, al);
        return;
    }
    float ax = 0.f, ay = 0.f, az = 0.f, lx = 0.f, ly = 0.f, lz = 0.f;
    uint32_t sA = 0, eA = 0, sB = 0, eB = 0;
    if (want) {
        sA = q.sA, eA = q.eA;
        sB = q.sB, eB = q.eB;
    }
    if (want)
        a_side_sum(g, o, sA, eA, ax, ay, az, lx, ly, lz);
    for (uint32_t base = loB; base < hiB; base += DEME_GATHER_TILE) {
        uint32_t idx[DEME_GATHER_TILE / 256];
#pragma unroll
        for (int k = 0; k < DEME_GATHER_TILE / 256; k++) {
            const uint32_t j = base + k * 256 + t;
            idx[k] = (j < hiB) ? g.bIdx[j] : 0xFFFFFFFFu;
        }
#pragma unroll
        for (int k = 0; k < DEME_GATHER_TILE / 256; k++)
            if (idx[k] != 0xFFFFFFFFu) {
                gather_b_load(g, idx[k], L.c4[k * 256 + t], L.c2[k * 256 + t]);
            }
        __syncthreads();
        const uint32_t s = (sB > base) ? sB : base;
        const uint32_t e = (eB < base + DEME_GATHER_TILE) ? eB : base + DEME_GATHER_TILE;
        for (uint32_t i = s; i < e; i++) {
            const float4 c4 = L.c4[i - base];
            const float2 c2 = L.c2[i - base];
            ax += c4.x, ay += c4.y, az += c4.z;
            lx += c4.w, ly += c2.x, lz += c2.y;
        }
        __syncthreads();
    }
    a = make_float4(ax, ay, az, 0.f);
    al = make_float4(lx, ly, lz, 0.f);
}

// (slab group, one evaluation per cross-cut contact) the a / alpha of the contacts the LEFT neighbour evaluated for this clump: added
// by the integrator and by every stand-alone reduction alike -- downloads, trackers and family rules that read accelerations see
// the same sums the integrator uses
__device__ inline void add_reverse_share(const GatherArgs& g, uint32_t o, float4& a, float4& al) {
    if (!g.revSlot)
        return;
    const uint32_t slot = g.revSlot[o];
    if (slot != 0xFFFFFFFFu) {
        const float4 ea = g.revAcc[2 * (size_t)slot], el = g.revAcc[2 * (size_t)slot + 1];
        a.x += ea.x, a.y += ea.y, a.z += ea.z;
        al.x += el.x, al.y += el.y, al.z += el.z;
    }
}

// stand-alone reduction (deme_calc_forces): a/alpha of every non-heavy owner
__global__ __launch_bounds__(256) void k_gather_acc(const DevParams p, const GatherArgs g, const OwnerRec* __restrict__ owners,
                                                    AccRec* __restrict__ acc) {
    const uint32_t o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= p.nOwners || g.heavy[o])
        return;
    float4 a, al;
    gather_owner(g, o, a, al);
    if (g.world)
        acc_from_world(p, load_owner(owners, o), a, al);
    add_reverse_share(g, o, a, al);
    float4* ap = reinterpret_cast<float4*>(acc + o);
    ap[0] = a;
    ap[1] = al;
}

// Owners with very many contacts (walls, large meshes): one workgroup each, fixed-shape tree
// reduction (deterministic; summation order differs from list order).
__global__ __launch_bounds__(256) void k_reduce_heavy(const DevParams p, const GatherArgs g, const OwnerRec* __restrict__ owners,
                                                      const uint32_t* __restrict__ heavyList,
                                                      const uint32_t* __restrict__ nHeavy,
                                                      const uint8_t* __restrict__ skip, AccRec* __restrict__ acc) {
    __shared__ float red[6][256];
    const uint32_t n = *nHeavy;
    for (uint32_t h = blockIdx.x; h < n; h += gridDim.x) {
        const uint32_t o = heavyList[h];
        if (skip && skip[o])
            continue;
        float s[6] = {0, 0, 0, 0, 0, 0};
        const uint32_t a0 = g.aStart[o], a1 = g.aStart[o + 1];
        if (g.tile || a_run_in_one_block(a0, a1)) {
            if (threadIdx.x == 0) {
#if DEME_REC24
                if (g.tile) {
                    const float2* t24 = reinterpret_cast<const float2*>(g.aSum) + 3 * (size_t)o;
                    s[0] = t24[0].x, s[1] = t24[0].y, s[2] = t24[1].x, s[3] = t24[1].y, s[4] = t24[2].x, s[5] = t24[2].y;
                } else
#endif
                {
                    const float4 v = g.aSum[2 * (size_t)o], w = g.aSum[2 * (size_t)o + 1];
                    s[0] = v.x, s[1] = v.y, s[2] = v.z, s[3] = w.x, s[4] = w.y, s[5] = w.z;
                }
            }
        } else {
            for (uint32_t c = a0 + threadIdx.x; c < a1; c += 256) {
                const float4 c4 = g.conA4[c];
                const float2 c2 = g.conA2[c];
                s[0] += c4.x, s[1] += c4.y, s[2] += c4.z, s[3] += c4.w, s[4] += c2.x, s[5] += c2.y;
            }
        }
        for (uint32_t i = g.bStart[o] + threadIdx.x; i < g.bStart[o + 1]; i += 256) {
            const uint32_t c = g.bIdx[i];
            float4 c4;
            float2 c2;
            gather_b_load(g, c, c4, c2);
            s[0] += c4.x, s[1] += c4.y, s[2] += c4.z, s[3] += c4.w, s[4] += c2.x, s[5] += c2.y;
        }
        for (int k = 0; k < 6; k++)
            red[k][threadIdx.x] = s[k];
        __syncthreads();
        for (int off = 128; off > 0; off >>= 1) {
            if (threadIdx.x < off)
                for (int k = 0; k < 6; k++)
                    red[k][threadIdx.x] += red[k][threadIdx.x + off];
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            float4 a = make_float4(red[0][0], red[1][0], red[2][0], 0.f), al = make_float4(red[3][0], red[4][0], red[5][0], 0.f);
            if (g.world)
                acc_from_world(p, load_owner(owners, o), a, al);
            if (!g.revInIntegrator)  // (the stepping loop's integrator adds the share itself: integrate_owner)
                add_reverse_share(g, o, a, al);
            float4* ap = reinterpret_cast<float4*>(acc + o);
            ap[0] = a;
            ap[1] = al;
        }
        __syncthreads();
    }
}

// Once per detection: each contact's gather record for the force kernel (see ForceArgs::info), A's owner
// (ascending: the list is sorted by sphere A and spheres are clump-major) and B's owner (sort key of the
// B-side lists).
__global__ __launch_bounds__(256) void k_contact_owners(const DevParams p, uint32_t nC, const uint64_t* __restrict__ keys,
                                                        const SphereRec* __restrict__ spheres,
                                                        uint32_t* __restrict__ ownerA, uint32_t* __restrict__ ownerB,
                                                        uint32_t* __restrict__ idx, uint4* __restrict__ info,
                                                        uint8_t* __restrict__ smFlag, uint32_t* __restrict__ tileRem,
                                                        uint32_t tileNB) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t tA = 0xFFFFFFFFu;
    bool remote = false;
    if (c < nC) {
        const uint64_t k = keys[c];
        const SphereRec sa = load_sphere(spheres, key_a(k));
        const uint32_t cls = key_class(k);
        uint32_t ob = 0, w = key_b(k);
        if (cls == DEME_KEY_CLASS_SS) {
            const SphereRec sb = load_sphere(spheres, key_b(k));
            ob = sb.owner;
            w = (uint32_t)sb.comp | ((uint32_t)sb.mat << 16);
        } else if (cls == DEME_KEY_CLASS_SA) {
            ob = p.anal[key_b(k)].owner;
        } else {  // sphere-mesh: TriRec is 48 bytes with the owner id at byte 36 (deme_mesh.h)
            ob = reinterpret_cast<const uint32_t*>(p.tris)[12 * (size_t)key_b(k) + 9];
        }
        ownerA[c] = sa.owner;
        ownerB[c] = ob;
        idx[c] = c;
        info[c] = make_uint4(sa.owner | (cls << 30), ob, (uint32_t)sa.comp | ((uint32_t)sa.mat << 16), w);
        if (smFlag)
            smFlag[c] = (cls == DEME_KEY_CLASS_SM) ? 1 : 0;
        if (tileRem) {
            tA = sa.owner / tileNB;
            remote = tA != ob / tileNB;
        }
    }
    if (tileRem) {  // per tile of tileNB owners: how many of its contacts (by A) have their B owner in another tile (deme_tile.h)
        // one atomic per wavefront and tile: the lanes of a wavefront sit in one tile, or in two neighbouring ones
        unsigned long long todo = __ballot(remote);
        while (todo) {
            const int first = __ffsll((long long)todo) - 1;
            const uint32_t t0 = (uint32_t)__shfl((int)tA, first);
            const unsigned long long m = __ballot(remote && tA == t0);
            if ((int)(threadIdx.x & 63u) == first)
                atomicAdd(&tileRem[t0], (uint32_t)__popcll(m));
            todo &= ~m;
        }
    }
}

// Persistent-contact qualification of every contact of the current list (DEM/APIPrivate.cpp:33-117, done there in a host
// loop): mode 0 all, 1 either owner's family == N1, 2 both == N1, 3 the pair (N1, N2) in either order.
__global__ __launch_bounds__(256) void k_persist_flags(const DevParams p, uint32_t nC, const uint64_t* __restrict__ keys,
                                                       const SphereRec* __restrict__ spheres,
                                                       const OwnerRec* __restrict__ owners, int mode, uint32_t N1, uint32_t N2,
                                                       uint8_t* __restrict__ flag) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nC)
        return;
    const uint64_t k = keys[c];
    const uint32_t oa = load_sphere(spheres, key_a(k)).owner, cls = key_class(k);
    uint32_t ob;
    if (cls == DEME_KEY_CLASS_SS)
        ob = load_sphere(spheres, key_b(k)).owner;
    else if (cls == DEME_KEY_CLASS_SA)
        ob = p.anal[key_b(k)].owner;
    else
        ob = reinterpret_cast<const uint32_t*>(p.tris)[12 * (size_t)key_b(k) + 9];
    const uint32_t fA = fam_of(owners[oa].family), fB = fam_of(owners[ob].family);
    const bool q = mode == 0 || (mode == 1 && (fA == N1 || fB == N1)) || (mode == 2 && fA == N1 && fB == N1) ||
                   (mode == 3 && ((fA == N1 && fB == N2) || (fA == N2 && fB == N1)));
    flag[c] = q ? 1 : 0;
}

__device__ inline uint32_t lower_bound_u32(const uint32_t* a, uint32_t n, uint32_t v) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (a[mid] < v)
            lo = mid + 1;
        else
            hi = mid;
    }
    return lo;
}

#define DEME_HEAVY_THRESHOLD 256u
struct RangeCounters {
    unsigned int nHeavy;
    unsigned int nHeavyFree;  // heavy owners that are not fixed (they must be reduced every step)
    unsigned int nSA, nSM;    // sphere-analytical / sphere-mesh contacts in the list
    unsigned int tileOverflow;  // (unused since the per-tile fallback: an overflowing tile is evaluated by k_tile_forces_big)
    unsigned int tileMaxHalo, tileMaxList;  // the largest tile's foreign owners / local-B list entries: they size the kernel's LDS
    unsigned int nBig;    // tiles whose halo, contact range or local lists do not fit the LDS area of k_tile_forces (deme_tile.h)
    unsigned int nExtra;  // records of those tiles' contacts that hold a B owner of the same tile (every contact of such a tile writes a record)
    unsigned int tileHaloSum;  // foreign owners staged by all fitting tiles together (the engine watches its mean: deme_order.inc)
    unsigned int nBigIn;         // closed tiles (deme_tile_step.h) whose halo, extended by their incoming contacts' owners, does not fit
    unsigned int tileMaxHaloIn;  // the largest closed tile's foreign owners
    unsigned int pad[4];
};

// start[o] = first index i with owner[i] >= o, for o = 0 .. nOwners (owner[] ascending): every element fills the owners that
// begin at it -- one store per element where four binary searches per owner used to walk the arrays
__global__ __launch_bounds__(256) void k_run_starts(uint32_t n, const uint32_t* __restrict__ owner, uint32_t nOwners,
                                                    uint32_t* __restrict__ start) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const uint32_t cur = owner[i];
    const uint32_t first = (i == 0) ? 0u : owner[i - 1] + 1u;
    for (uint32_t o = first; o <= cur && o <= nOwners; o++)
        start[o] = i;
    if (i == n - 1)
        for (uint32_t o = cur + 1; o <= nOwners; o++)
            start[o] = n;
}

// aStart / bStart (k_run_starts) -> per-owner flags, the heavy-owner list, the halo split (once per detection).
__global__ __launch_bounds__(256) void k_owner_ranges(const DevParams p, uint32_t nC, const uint32_t* __restrict__ ownerA,
                                                      const uint32_t* __restrict__ ownerBSorted,
                                                      const OwnerRec* __restrict__ owners, uint32_t* __restrict__ aStart,
                                                      uint32_t* __restrict__ bStart, uint8_t* __restrict__ heavy,
                                                      uint8_t* __restrict__ fixedFlag, uint32_t* __restrict__ heavyList,
                                                      uint32_t heavyCap, RangeCounters* rc,
                                                      const uint4* __restrict__ info, uint8_t* __restrict__ cDefer,
                                                      uint32_t* __restrict__ blockMode) {
    const uint32_t o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o > p.nOwners)
        return;
    if (o == p.nOwners)
        return;
    const uint32_t a0 = aStart[o], a1 = aStart[o + 1];
    const uint32_t b0 = bStart[o], b1 = bStart[o + 1];
    const uint32_t fw = owners[o].family;
    const bool isFixed = (p.familyFlags[fam_of(fw)] & 1u) != 0 || ghost_of(fw);  // fixed or ghost: a/alpha never integrated here
    fixedFlag[o] = isFixed ? 1 : 0;
    if (cDefer) {  // halo overlap: an owner run that reads any ghost owner is evaluated after the ghost records arrive
        bool d = ghost_of(fw);
        for (uint32_t c = a0; c < a1 && !d; c++)
            d = ghost_of(owners[info[c].y].family);  // info.y: B's owner (k_contact_owners)
        for (uint32_t c = a0; c < a1; c++) {
            cDefer[c] = d ? 1 : 0;
            if (c == a0 || (c % DEME_FORCE_BLOCK) == 0)
                atomicOr(&blockMode[c / DEME_FORCE_BLOCK], d ? 2u : 1u);
        }
    }
    // a replicated free owner always takes the separate reduction: its sum has to exist in memory for the cross-slab addition
    const bool hv = (a1 - a0) + (b1 - b0) > DEME_HEAVY_THRESHOLD || shared_of(fw);
    heavy[o] = hv ? 1 : 0;
    if (hv) {
        const unsigned int slot = atomicAdd(&rc->nHeavy, 1u);
        if (slot < heavyCap)
            heavyList[slot] = o;
        if (!isFixed)
            atomicAdd(&rc->nHeavyFree, 1u);
    }
}

struct PrescArgs {
    const PrescRec* rec;    // null: no family carries a prescription
    const uint32_t* slot;   // per owner: index into rec (only read for owners of a prescribed family)
};

// The 64 consecutive owner records of a wavefront, read and written as 4 KB of contiguous memory (four fully coalesced 16-byte
// accesses per lane) and transposed to one record per lane through LDS -- a lane that loads or stores its own 64-byte record
// piece by piece makes every instruction touch 64 different cache lines (the vector L1 serves one per cycle).
#ifndef DEME_INT_COOP
#define DEME_INT_COOP 1
#endif
#define DEME_INT_STAGE (64 * 5)  // uint4 per wavefront: records at an 80-byte stride (bank-conflict-free ds_read_b128 per lane)
__device__ inline void lds_wave_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // LDS operations of one wavefront complete in order: only the
    __builtin_amdgcn_wave_barrier();                         // compiler has to be kept from reordering
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ inline OwnerRec coop_load_owner(const OwnerRec* owners, uint32_t nOwners, uint32_t waveBase, uint4* stage) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint4* src = reinterpret_cast<const uint4*>(owners) + (size_t)waveBase * 4;
    const uint32_t lim = (nOwners > waveBase ? min(64u, nOwners - waveBase) : 0u) * 4u;
    uint4 v[4];
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) {
        const uint32_t e = k * 64u + lane;
        v[k] = (e < lim) ? src[e] : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) {
        const uint32_t e = k * 64u + lane;
        stage[(e >> 2) * 5u + (e & 3u)] = v[k];
    }
    lds_wave_fence();
    OwnerRec r;
    uint4* q = reinterpret_cast<uint4*>(&r);
    const uint4* mine = stage + lane * 5u;
    q[0] = mine[0], q[1] = mine[1], q[2] = mine[2], q[3] = mine[3];
    return r;
}
__device__ inline void coop_store_owner(OwnerRec* owners, uint32_t nOwners, uint32_t waveBase, uint4* stage, const OwnerRec& r) {
    const uint32_t lane = threadIdx.x & 63u;
    uint4* dst = reinterpret_cast<uint4*>(owners) + (size_t)waveBase * 4;
    const uint32_t lim = (nOwners > waveBase ? min(64u, nOwners - waveBase) : 0u) * 4u;
    const uint4* q = reinterpret_cast<const uint4*>(&r);
    uint4* mine = stage + lane * 5u;
    lds_wave_fence();
    mine[0] = q[0], mine[1] = q[1], mine[2] = q[2], mine[3] = q[3];
    lds_wave_fence();
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) {
        const uint32_t e = k * 64u + lane;
        if (e < lim)
            dst[e] = stage[(e >> 2) * 5u + (e & 3u)];
    }
}

__device__ inline void integrate_owner(const DevParams& p, OwnerRec& r, float4 a, float4 al, uint32_t o, uint32_t fflags, bool fixed,
                                       const GatherArgs& g, const PrescArgs& pa);

template <bool FUSED>
__global__ __launch_bounds__(256) void k_integrate(const DevParams p, OwnerRec* __restrict__ owners,
                                                   AccRec* __restrict__ acc, const GatherArgs g, const PrescArgs pa) {
    const uint32_t o = blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = o < p.nOwners;
    // one LDS area: the record transposes at both ends of the kernel, the gather tiles of the fused path in between
    constexpr uint32_t kStage16 = DEME_INT_COOP ? 4u * DEME_INT_STAGE : 1u, kGather16 = FUSED ? (uint32_t)(sizeof(GatherLds) / 16) : 1u;
    __shared__ uint4 smem[kStage16 > kGather16 ? kStage16 : kGather16];
    GatherPre pre{};
    if (FUSED)
        pre = gather_prefetch(g, p.nOwners, o, valid);
#if DEME_INT_COOP
    uint4* stage = smem + (threadIdx.x >> 6) * DEME_INT_STAGE;
    const uint32_t waveBase = o - (threadIdx.x & 63u);
    OwnerRec r = coop_load_owner(owners, p.nOwners, waveBase, stage);
    if (FUSED)
        __syncthreads();  // every wavefront has its records in registers before the gather tiles overwrite the area
#else
    OwnerRec r = load_owner(owners, valid ? o : 0u);
#endif
    const uint32_t fflags = p.familyFlags[fam_of(r.family)];
    const bool ghost = ghost_of(r.family);  // its owner rank integrates it; refreshed by deme_halo_unpack
    const bool fixed = (fflags & 1u) != 0;
    // (slab group, one evaluation per cross-cut contact) an owner whose share of the left neighbour's contacts is still on its way is
    // left as it is: k_integrate_list takes it when the share has arrived, the rest of the slab integrates beside the exchange
    const bool later = g.revPhase == 1u && g.revSlot && valid && g.revSlot[o] != 0xFFFFFFFFu;
    float4 a = make_float4(0, 0, 0, 0), al = a;
    if (FUSED) {
        // a fixed owner's a/alpha never feed the integrator (they are reduced on demand); a/alpha are not stored in
        // the stepping loop either (32 B/owner of HBM writes per step saved): a state download re-derives them from
        // the per-contact contributions (launch_full_reduction)
        GatherLds& lds = *reinterpret_cast<GatherLds*>(smem);
        const bool hv = valid && g.heavy[o];
        gather_block(g, p.nOwners, o, valid && !ghost && !fixed && !hv && !later, lds, a, al, pre);
        if (valid && !ghost && !later) {
            if (g.world && !fixed && !hv)
                acc_from_world(p, r, a, al);
            if (hv) {
                const float4* ap = reinterpret_cast<const float4*>(acc + o);
                a = ap[0];
                al = ap[1];
            }
        }
    } else if (valid && !ghost) {
        const float4* ap = reinterpret_cast<const float4*>(acc + o);
        a = ap[0];
        al = ap[1];
    }
    if (valid && !ghost && !later)
        integrate_owner(p, r, a, al, o, fflags, fixed, g, pa);
#if DEME_INT_COOP
    coop_store_owner(owners, p.nOwners, waveBase, stage, r);  // a record nobody integrated goes back as it came
#else
    if (valid && !ghost && !later)
        store_owner(owners, o, r);
#endif
}

// the owners k_integrate left for later (revPhase 1): the same update, one thread per listed owner, sums gathered per thread in the
// same fixed order (gather_owner == gather_block bit for bit)
__global__ __launch_bounds__(256) void k_integrate_list(const DevParams p, OwnerRec* __restrict__ owners, const AccRec* __restrict__ acc,
                                                        const GatherArgs g, const PrescArgs pa, const uint32_t* __restrict__ ids,
                                                        uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const uint32_t o = ids[i];
    OwnerRec r = load_owner(owners, o);
    if (ghost_of(r.family))
        return;
    const uint32_t fflags = p.familyFlags[fam_of(r.family)];
    const bool fixed = (fflags & 1u) != 0;
    float4 a = make_float4(0, 0, 0, 0), al = a;
    if (g.heavy[o]) {
        const float4* ap = reinterpret_cast<const float4*>(acc + o);
        a = ap[0], al = ap[1];
    } else if (!fixed) {
        gather_owner(g, o, a, al);
        if (g.world)
            acc_from_world(p, r, a, al);
    }
    integrate_owner(p, r, a, al, o, fflags, fixed, g, pa);
    store_owner(owners, o, r);
}

// the explicit update of one owner (DEMIntegrationKernels.cu:100-236); r is updated in place
__device__ inline void integrate_owner(const DevParams& p, OwnerRec& r, float4 a, float4 al, uint32_t o, uint32_t fflags, bool fixed,
                                       const GatherArgs& g, const PrescArgs& pa) {

    add_reverse_share(g, o, a, al);  // the share of the contacts a neighbouring rank evaluated for this clump
    if (g.nextAcc) {  // DEMTracker::AddAcc / AddAngAcc: on top of the contact sums (added last: the sums keep their order)
        const float4* ep = reinterpret_cast<const float4*>(g.nextAcc + o);
        const float4 ea = ep[0], el = ep[1];
        a.x += ea.x, a.y += ea.y, a.z += ea.z;
        al.x += el.x, al.y += el.y, al.z += el.z;
    }
    const float h = p.h;
    f3 old_v = mk3(r.vx, r.vy, r.vz), old_w = mk3(r.wx, r.wy, r.wz);
    d3 X = decode_pos(r.voxelID, r.locX, r.locY, r.locZ, p);
    X.x += (double)p.LBFX;
    X.y += (double)p.LBFY;
    X.z += (double)p.LBFZ;
    // prescribed motion (applyPrescribedVel / applyPrescribedPos / applyAddedAcceleration,
    // DEMIntegrationKernels.cu:8-98): evaluated by the run-time compiled k_prescribe just before this kernel
    uint32_t pf = 0u;
    f3 extra_a = mk3(0, 0, 0), extra_al = mk3(0, 0, 0);
    if (pa.rec && (fflags & 4u)) {
        const PrescRec pr = pa.rec[pa.slot[o]];
        pf = pr.flags;
        r.vx = pr.vx, r.vy = pr.vy, r.vz = pr.vz;
        r.wx = pr.wx, r.wy = pr.wy, r.wz = pr.wz;
        r.qw = pr.qw, r.qx = pr.qx, r.qy = pr.qy, r.qz = pr.qz;
        X = {pr.X, pr.Y, pr.Z};
        extra_a = mk3(pr.ax, pr.ay, pr.az);
        extra_al = mk3(pr.lx, pr.ly, pr.lz);
    }
    f3 v_upd = mk3(0, 0, 0), w_upd = mk3(0, 0, 0);
    if (fixed) {
        r.vx = r.vy = r.vz = 0.f;
        r.wx = r.wy = r.wz = 0.f;
        old_v = mk3(0, 0, 0);
        old_w = mk3(0, 0, 0);
    } else {
        if (!(pf & 1u)) {
            v_upd.x = (a.x + extra_a.x + p.Gx) * h;
            r.vx += v_upd.x;
        } else {
            old_v.x = r.vx;
        }
        if (!(pf & 2u)) {
            v_upd.y = (a.y + extra_a.y + p.Gy) * h;
            r.vy += v_upd.y;
        } else {
            old_v.y = r.vy;
        }
        if (!(pf & 4u)) {
            v_upd.z = (a.z + extra_a.z + p.Gz) * h;
            r.vz += v_upd.z;
        } else {
            old_v.z = r.vz;
        }
        if (!(pf & 8u)) {
            w_upd.x = (al.x + extra_al.x) * h;
            r.wx += w_upd.x;
        } else {
            old_w.x = r.wx;
        }
        if (!(pf & 16u)) {
            w_upd.y = (al.y + extra_al.y) * h;
            r.wy += w_upd.y;
        } else {
            old_w.y = r.wy;
        }
        if (!(pf & 32u)) {
            w_upd.z = (al.z + extra_al.z) * h;
            r.wz += w_upd.z;
        } else {
            old_w.z = r.wz;
        }
    }
    f3 v, w;
    if (p.integrator == 0) {
        v = old_v;
        w = old_w;
    } else if (p.integrator == 1) {
        v = old_v + v_upd;
        w = old_w + w_upd;
    } else {
        v = old_v + v_upd * 0.5f;
        w = old_w + w_upd * 0.5f;
    }
    if (!fixed) {
        if (!(pf & 64u))
            X.x += (double)v.x * h;
        if (!(pf & 128u))
            X.y += (double)v.y * h;
        if (!(pf & 256u))
            X.z += (double)v.z * h;
    }
    X.x -= (double)p.LBFX;
    X.y -= (double)p.LBFY;
    X.z -= (double)p.LBFZ;
    encode_pos(X, p, r.voxelID, r.locX, r.locY, r.locZ);
    if (!fixed && !(pf & 512u)) {
        const float hh = (float)(0.5 * h);
        const f3 ha = hh * w;
        // HamiltonProduct(q, (1, ha)) : DEMHelperKernels.cuh:228-245
        const float a1 = r.qw, b1 = r.qx, c1 = r.qy, d1 = r.qz;
        const float a2 = 1.0f, b2 = ha.x, c2 = ha.y, d2 = ha.z;
        const float qw = a1 * a2 - b1 * b2 - c1 * c2 - d1 * d2;
        const float qx = a1 * b2 + b1 * a2 + c1 * d2 - d1 * c2;
        const float qy = a1 * c2 - b1 * d2 + c1 * a2 + d1 * b2;
        const float qz = a1 * d2 + b1 * c2 - c1 * b2 + d1 * a2;
        const float len = sqrtf(qx * qx + qy * qy + qz * qz + qw * qw);
        r.qw = qw / len;
        r.qx = qx / len;
        r.qy = qy / len;
        r.qz = qz / len;
    }
}

// ---- packing between the C-ABI's SoA view and the device records -------------------------------
struct OwnerSoA {
    uint64_t* voxelID;
    uint16_t *locX, *locY, *locZ;
    float *oriQw, *oriQx, *oriQy, *oriQz, *vX, *vY, *vZ, *omgBarX, *omgBarY, *omgBarZ;
    float *aX, *aY, *aZ, *alphaX, *alphaY, *alphaZ;
    uint8_t* familyID;
    uint16_t* inertiaPropOffsets;
};

// dir 0: SoA -> records (null members keep the record's value); dir 1: records -> SoA
__global__ __launch_bounds__(256) void k_pack_owners(uint32_t n, OwnerRec* owners, AccRec* acc, OwnerSoA s, int dir, const uint32_t* __restrict__ o2e) {
    const uint32_t o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= n)
        return;
    OwnerRec r = owners[o];
    AccRec a = acc[o];
    const uint32_t e = o2e ? o2e[o] : o;  // the caller's number of this owner (engine-side spatial order, deme_order.inc)
    if (dir == 0) {
        if (s.voxelID) r.voxelID = s.voxelID[e];
        if (s.locX) r.locX = s.locX[e];
        if (s.locY) r.locY = s.locY[e];
        if (s.locZ) r.locZ = s.locZ[e];
        if (s.oriQw) r.qw = s.oriQw[e];
        if (s.oriQx) r.qx = s.oriQx[e];
        if (s.oriQy) r.qy = s.oriQy[e];
        if (s.oriQz) r.qz = s.oriQz[e];
        if (s.vX) r.vx = s.vX[e];
        if (s.vY) r.vy = s.vY[e];
        if (s.vZ) r.vz = s.vZ[e];
        if (s.omgBarX) r.wx = s.omgBarX[e];
        if (s.omgBarY) r.wy = s.omgBarY[e];
        if (s.omgBarZ) r.wz = s.omgBarZ[e];
        if (s.familyID) r.family = (r.family & OWNER_FLAG_BITS) | s.familyID[e];
        if (s.inertiaPropOffsets) r.inertiaOff = s.inertiaPropOffsets[e];
        if (s.aX) a.ax = s.aX[e];
        if (s.aY) a.ay = s.aY[e];
        if (s.aZ) a.az = s.aZ[e];
        if (s.alphaX) a.lx = s.alphaX[e];
        if (s.alphaY) a.ly = s.alphaY[e];
        if (s.alphaZ) a.lz = s.alphaZ[e];
        owners[o] = r;
        acc[o] = a;
    } else {
        if (s.voxelID) s.voxelID[e] = r.voxelID;
        if (s.locX) s.locX[e] = r.locX;
        if (s.locY) s.locY[e] = r.locY;
        if (s.locZ) s.locZ[e] = r.locZ;
        if (s.oriQw) s.oriQw[e] = r.qw;
        if (s.oriQx) s.oriQx[e] = r.qx;
        if (s.oriQy) s.oriQy[e] = r.qy;
        if (s.oriQz) s.oriQz[e] = r.qz;
        if (s.vX) s.vX[e] = r.vx;
        if (s.vY) s.vY[e] = r.vy;
        if (s.vZ) s.vZ[e] = r.vz;
        if (s.omgBarX) s.omgBarX[e] = r.wx;
        if (s.omgBarY) s.omgBarY[e] = r.wy;
        if (s.omgBarZ) s.omgBarZ[e] = r.wz;
        if (s.familyID) s.familyID[e] = (uint8_t)r.family;
        if (s.aX) s.aX[e] = a.ax;
        if (s.aY) s.aY[e] = a.ay;
        if (s.aZ) s.aZ[e] = a.az;
        if (s.alphaX) s.alphaX[e] = a.lx;
        if (s.alphaY) s.alphaY[e] = a.ly;
        if (s.alphaZ) s.alphaZ[e] = a.lz;
    }
}

__global__ __launch_bounds__(256) void k_change_family(OwnerRec* __restrict__ owners, uint32_t n, uint32_t from, uint32_t to) {
    const uint32_t o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o < n && fam_of(owners[o].family) == from)
        owners[o].family = (owners[o].family & OWNER_FLAG_BITS) | to;
}

__global__ __launch_bounds__(256) void k_set_ghost_bits(uint32_t n, OwnerRec* __restrict__ owners, const uint8_t* __restrict__ flag) {
    const uint32_t o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o < n)
        owners[o].family = fam_of(owners[o].family) | ((flag[o] & 1u) ? OWNER_GHOST_BIT : 0u) | ((flag[o] & 2u) ? OWNER_SHARED_BIT : 0u);
}

// cross-slab addition of the replicated free owners' a / alpha: rows of the owners' AccRec to and from a contiguous buffer, and
// the sum of the buffers of the slabs one process holds (slab order: the same on every rank)
__global__ __launch_bounds__(256) void k_shared_pack(uint32_t n, const uint32_t* __restrict__ ids, const AccRec* __restrict__ acc, float4* __restrict__ buf) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2 * n)
        return;
    buf[i] = reinterpret_cast<const float4*>(acc + ids[i >> 1])[i & 1u];
}
__global__ __launch_bounds__(256) void k_shared_unpack(uint32_t n, const uint32_t* __restrict__ ids, AccRec* __restrict__ acc, const float4* __restrict__ buf) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2 * n)
        return;
    reinterpret_cast<float4*>(acc + ids[i >> 1])[i & 1u] = buf[i];
}
__global__ __launch_bounds__(256) void k_shared_add(uint32_t n4, float4* __restrict__ into, const float4* __restrict__ other) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4)
        return;
    const float4 a = into[i], b = other[i];
    into[i] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
}

// SetFamilyClumpMaterial / SetFamilyMeshMaterial: geometries whose owner is of `family` take `material`
__global__ __launch_bounds__(256) void k_family_material_spheres(uint32_t n, SphereRec* __restrict__ spheres,
                                                                 const OwnerRec* __restrict__ owners, uint32_t family, uint32_t material) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < n && (owners[spheres[s].owner].family & 0xFFu) == family)
        spheres[s].mat = (uint16_t)material;
}

// ---- inspectors: DEMSphereQueryKernels.cu:13-54 / DEMOwnerQueryKernels.cu:11-63 with the quantity fragments of
// AuxClasses.cpp:19-92.  Elements that do not take part get the reduction's identity.
__global__ __launch_bounds__(256) void k_inspect_sphere(const DevParams p, const OwnerRec* __restrict__ owners,
                                                       const SphereRec* __restrict__ spheres, uint32_t quantity, float identity,
                                                       float* __restrict__ out) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= p.nSpheres)
        return;
    const SphereRec sr = spheres[s];
    const OwnerRec o = load_owner(owners, sr.owner);
    if (ghost_of(o.family)) {  // ghost copy: its owner rank reports it
        out[s] = identity;
        return;
    }
    const float4 c = p.comp[sr.comp];
    const RotM R = rot_coeffs(o.qw, o.qx, o.qy, o.qz);
    const f3 rel = rot_apply(R, mk3(c.x, c.y, c.z));  // myRelPos, rotated in place
    const d3 X = decode_pos(o.voxelID, o.locX, o.locY, o.locZ, p);
    float q;
    if (quantity == 0u) {
        const float Z = (float)(X.z + (double)rel.z + (double)p.LBFZ);
        q = Z + c.w;
    } else if (quantity == 1u) {
        const float Z = (float)(X.z + (double)rel.z + (double)p.LBFZ);
        q = Z - c.w;
    } else {  // INSP_CODE_SPHERE_HIGH_ABSV: cross(omgBar, rotated relPos), rotated once more, + v
        const f3 pr = rot_apply(R, cross3(mk3(o.wx, o.wy, o.wz), rel));
        q = len3(pr + mk3(o.vx, o.vy, o.vz));
    }
    out[s] = q;
}

__global__ __launch_bounds__(256) void k_inspect_owner(const DevParams p, const OwnerRec* __restrict__ owners, uint32_t nClumps,
                                                      uint32_t quantity, float identity, const float* __restrict__ volumes,
                                                      float* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.nOwners)
        return;
    const OwnerRec o = load_owner(owners, i);
    const bool clumpOnly = quantity == 3u || quantity == 5u || quantity == 7u;  // OWNER_T_CLUMP quantities
    if (ghost_of(o.family) || (clumpOnly && i >= nClumps)) {
        out[i] = identity;
        return;
    }
    const float4 mp = p.massProps[o.inertiaOff];
    float q;
    if (quantity == 3u) {
        q = mp.x;
    } else if (quantity == 7u) {  // INSP_CODE_CLUMP_APPROX_VOL: the template's volume as the user declared it
        q = volumes[o.inertiaOff];
    } else if (quantity == 5u) {
        double vx = o.vx, vy = o.vy, vz = o.vz;
        double ke = 0.5 * mp.x * (vx * vx + vy * vy + vz * vz);
        vx = o.wx, vy = o.wy, vz = o.wz;
        ke += 0.5 * ((double)mp.y * vx * vx + (double)mp.z * vy * vy + (double)mp.w * vz * vz);
        q = (float)ke;
    } else {
        const double vx = o.vx, vy = o.vy, vz = o.vz;
        q = (float)sqrt(vx * vx + vy * vy + vz * vz);
    }
    out[i] = q;
}

// ghost-owner exchange records for the slab decomposition: pose + velocities + family
struct __attribute__((aligned(8))) GhostRec {
    uint64_t voxelID;
    uint16_t locX, locY, locZ, family;
    float qw, qx, qy, qz, vx, vy, vz, wx, wy, wz;
};
static_assert(sizeof(GhostRec) == 56, "ghost record is 56 bytes");

__global__ __launch_bounds__(256) void k_halo_pack(uint32_t n, const uint32_t* ids, const OwnerRec* owners, GhostRec* buf) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const OwnerRec r = load_owner(owners, ids[i]);
    GhostRec g;
    g.voxelID = r.voxelID, g.locX = r.locX, g.locY = r.locY, g.locZ = r.locZ, g.family = (uint16_t)fam_of(r.family);
    g.qw = r.qw, g.qx = r.qx, g.qy = r.qy, g.qz = r.qz;
    g.vx = r.vx, g.vy = r.vy, g.vz = r.vz, g.wx = r.wx, g.wy = r.wy, g.wz = r.wz;
    buf[i] = g;
}
// One evaluation per cross-cut contact: the evaluating rank turns each ghost's contact sum into the a / alpha its owner rank adds
// to the clump's own before integrating (the ghost's record carries what the conversion needs: orientation, mass properties)
__global__ __launch_bounds__(256) void k_ghost_acc_pack(const DevParams p, uint32_t n, const uint32_t* __restrict__ ids,
                                                        const GatherArgs g, const OwnerRec* __restrict__ owners,
                                                        float4* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const uint32_t o = ids[i];
    float4 a, al;
    gather_owner(g, o, a, al);
    if (g.world)
        acc_from_world(p, load_owner(owners, o), a, al);
    out[2 * (size_t)i] = a, out[2 * (size_t)i + 1] = al;
}
__global__ __launch_bounds__(256) void k_owner_set_bits(uint32_t n, const uint32_t* __restrict__ ids, OwnerRec* owners, uint32_t clear,
                                                        uint32_t set) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        owners[ids[i]].family = (owners[ids[i]].family & ~clear) | set;
}
__global__ __launch_bounds__(256) void k_rev_slots(uint32_t n, const uint32_t* __restrict__ ids, uint32_t* __restrict__ slot) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        slot[ids[i]] = i;
}
__global__ __launch_bounds__(256) void k_halo_unpack(uint32_t n, const uint32_t* ids, OwnerRec* owners, const GhostRec* buf) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const GhostRec g = buf[i];
    OwnerRec* r = owners + ids[i];
    // the copy follows its owner rank's record, family included (on-the-fly family changes travel with the state)
    r->family = (r->family & OWNER_FLAG_BITS) | g.family;
    r->voxelID = g.voxelID, r->locX = g.locX, r->locY = g.locY, r->locZ = g.locZ;
    r->qw = g.qw, r->qx = g.qx, r->qy = g.qy, r->qz = g.qz;
    r->vx = g.vx, r->vy = g.vy, r->vz = g.vz, r->wx = g.wx, r->wy = g.wy, r->wz = g.wz;
}

}  // namespace deme_dev
