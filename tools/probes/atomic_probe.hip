// Throughput of global u32 atomics on gfx950 for the access patterns a counting sort of the detection lists would have:
// n operations over m counters, addresses either local (operation i goes near counter i*m/n, +- a jitter) or random.
// hipcc --offload-arch=gfx950 -O3 -o atomic_probe atomic_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
static inline uint32_t hash(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__global__ void k_noret(uint32_t n, const uint32_t* idx, uint32_t* cnt) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) atomicAdd(&cnt[idx[i]], 1u);
}
__global__ void k_ret(uint32_t n, const uint32_t* idx, uint32_t* cnt, uint32_t* out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = atomicAdd(&cnt[idx[i]], 1u);
}
__global__ void k_scatter(uint32_t n, const uint32_t* idx, const uint32_t* off, uint32_t* cnt, uint2* out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const uint32_t b = idx[i]; const uint32_t s = off[b] + atomicAdd(&cnt[b], 1u); out[s] = make_uint2(b, i); }
}
__global__ void k_plain(uint32_t n, const uint32_t* idx, uint32_t* out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = idx[i] + 1;
}
// one returning 64-bit atomic on ONE address per workgroup (the reservation k_sweep makes per window), after `work` dependent FMAs
__global__ void k_reserve(unsigned long long* ctr, uint32_t* out, int work, int spread) {
    float a = threadIdx.x;
    for (int i = 0; i < work; i++) a = a * 1.0001f + 0.5f;
    __shared__ unsigned long long base;
    if (threadIdx.x == 0) base = atomicAdd(&ctr[(blockIdx.x % spread) * 16], 64ull);
    __syncthreads();
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)base + (uint32_t)a;
}
// lanes of a wave hitting one address individually (k_sphere_prep's sphere-wall contacts): fraction 1/every of the threads
__global__ void k_lanes(unsigned long long* ctr, uint32_t* out, uint32_t n, uint32_t every, int aggregate) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const bool hit = (i / 64) % every == 0;  // whole wavefronts hit together (a wall layer is contiguous in a space-filling order)
    if (!aggregate) {
        if (hit) out[i] = (uint32_t)atomicAdd(ctr, 1ull);
    } else {
        const unsigned long long m = __ballot(hit);
        if (m) {
            const int leader = __ffsll((long long)m) - 1;
            unsigned long long b = 0;
            if ((int)(threadIdx.x & 63) == leader) b = atomicAdd(ctr, (unsigned long long)__popcll(m));
            b = __shfl(b, leader);
            if (hit) out[i] = (uint32_t)b + __popcll(m & ((1ull << (threadIdx.x & 63)) - 1ull));
        }
    }
}
int main() {
    {
        unsigned long long* ctr; uint32_t* out;
        hipMalloc(&ctr, 64 * 16 * 8); hipMalloc(&out, 32768u * 256 * 4);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int spread : {1, 8, 64})
            for (int work : {0, 2000, 20000}) {
                float tot = 0;
                for (int rep = 0; rep < 4; rep++) {
                    float ms; hipMemset(ctr, 0, 64 * 16 * 8);
                    hipEventRecord(e0); k_reserve<<<32768, 256>>>(ctr, out, work, spread); hipEventRecord(e1); hipEventSynchronize(e1);
                    hipEventElapsedTime(&ms, e0, e1); if (rep) tot += ms;
                }
                printf("32768 workgroups, one returning atomic each over %2d address(es), %5d FMAs of work: %7.1f us\n", spread, work, tot / 3 * 1e3);
            }
        for (uint32_t every : {1000u, 100u, 20u})
            for (int agg = 0; agg < 2; agg++) {
                float tot = 0;
                for (int rep = 0; rep < 4; rep++) {
                    float ms; hipMemset(ctr, 0, 8);
                    hipEventRecord(e0); k_lanes<<<(3000000 + 255) / 256, 256>>>(ctr, out, 3000000, every, agg); hipEventRecord(e1); hipEventSynchronize(e1);
                    hipEventElapsedTime(&ms, e0, e1); if (rep) tot += ms;
                }
                printf("3e6 threads, every %4u-th wavefront's lanes append to one counter (%u lanes), %s: %7.1f us\n", every, 3000000 / every, agg ? "one atomic per wavefront" : "one atomic per lane", tot / 3 * 1e3);
            }
        hipFree(ctr); hipFree(out);
    }
    const uint32_t sizes[][2] = {{4300000, 1000000}, {4300000, 3000000}, {8200000, 600000}, {8200000, 2000000}};
    for (auto& sz : sizes) {
        const uint32_t n = sz[0], m = sz[1];
        for (int pat = 0; pat < 3; pat++) {
            std::vector<uint32_t> h(n);
            for (uint32_t i = 0; i < n; i++) {
                const uint32_t c = (uint32_t)((uint64_t)i * m / n);
                if (pat == 0) h[i] = (c + hash(i) % 64) % m;          // local, jitter 64
                else if (pat == 1) h[i] = (c + hash(i) % 4096) % m;   // local, jitter 4096
                else h[i] = hash(i) % m;                              // random
            }
            uint32_t *idx, *cnt, *out, *off; uint2* o2;
            hipMalloc(&idx, n * 4); hipMalloc(&cnt, (m + 1) * 4); hipMalloc(&out, n * 4); hipMalloc(&off, (m + 1) * 4); hipMalloc(&o2, n * 8);
            hipMemcpy(idx, h.data(), n * 4, hipMemcpyHostToDevice);
            std::vector<uint32_t> ho(m + 1, 0), hc(m, 0);
            for (uint32_t i = 0; i < n; i++) hc[h[i]]++;
            for (uint32_t b = 0; b < m; b++) ho[b + 1] = ho[b] + hc[b];
            hipMemcpy(off, ho.data(), (m + 1) * 4, hipMemcpyHostToDevice);
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            float t[4] = {0, 0, 0, 0};
            for (int rep = 0; rep < 6; rep++) {
                float ms;
                hipMemset(cnt, 0, m * 4);
                hipEventRecord(e0); k_noret<<<(n + 255) / 256, 256>>>(n, idx, cnt); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); if (rep) t[0] += ms;
                hipMemset(cnt, 0, m * 4);
                hipEventRecord(e0); k_ret<<<(n + 255) / 256, 256>>>(n, idx, cnt, out); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); if (rep) t[1] += ms;
                hipMemset(cnt, 0, m * 4);
                hipEventRecord(e0); k_scatter<<<(n + 255) / 256, 256>>>(n, idx, off, cnt, o2); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); if (rep) t[2] += ms;
                hipEventRecord(e0); k_plain<<<(n + 255) / 256, 256>>>(n, idx, out); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); if (rep) t[3] += ms;
            }
            printf("n %8u m %8u pattern %d: no-return %7.1f us  with-return %7.1f us  count+scatter 8B %7.1f us  (plain copy %6.1f us)\n", n, m, pat,
                   t[0] / 5 * 1e3, t[1] / 5 * 1e3, t[2] / 5 * 1e3, t[3] / 5 * 1e3);
            hipFree(idx); hipFree(cnt); hipFree(out); hipFree(off); hipFree(o2);
        }
    }
    return 0;
}
