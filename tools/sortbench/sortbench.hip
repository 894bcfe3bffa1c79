// micro-benchmark: rocprim radix_sort_pairs on (u32 bin, u32 sphere) incidences with different onesweep digit widths
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <cstdio>
#include <cstring>
#include <vector>
#include <random>
#ifndef SB_RB
#define SB_RB 8
#endif
#ifndef SB_BS
#define SB_BS 512
#endif
#ifndef SB_IPT
#define SB_IPT 12
#endif
using Cfg = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                       rocprim::radix_sort_onesweep_config<rocprim::kernel_config<SB_BS, SB_IPT>, rocprim::kernel_config<SB_BS, SB_IPT>, SB_RB,
                                                                           rocprim::block_radix_rank_algorithm::match>>;
int main(int argc, char** argv) {
    const size_t n = argc > 1 ? atol(argv[1]) : 10237635;
    const unsigned bits = argc > 2 ? atoi(argv[2]) : 23;
    std::vector<uint32_t> hk(n), hv(n);
    std::mt19937 g(1);
    for (size_t i = 0; i < n; i++) hk[i] = g() & ((1u << bits) - 1), hv[i] = (uint32_t)i;
    uint32_t *k0, *k1, *v0, *v1;
    hipMalloc(&k0, n * 4), hipMalloc(&k1, n * 4), hipMalloc(&v0, n * 4), hipMalloc(&v1, n * 4);
    hipMemcpy(k0, hk.data(), n * 4, hipMemcpyHostToDevice), hipMemcpy(v0, hv.data(), n * 4, hipMemcpyHostToDevice);
    size_t need = 0;
    rocprim::radix_sort_pairs<Cfg>(nullptr, need, k0, k1, v0, v1, n, 0, bits, 0);
    void* tmp;
    hipMalloc(&tmp, need);
    hipEvent_t a, b;
    hipEventCreate(&a), hipEventCreate(&b);
    float best = 1e9;
    for (int it = 0; it < 6; it++) {
        hipEventRecord(a, 0);
        rocprim::radix_sort_pairs<Cfg>(tmp, need, k0, k1, v0, v1, n, 0, bits, 0);
        hipEventRecord(b, 0);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    std::vector<uint32_t> ok(n);
    hipMemcpy(ok.data(), k1, n * 4, hipMemcpyDeviceToHost);
    bool sorted = true;
    for (size_t i = 1; i < n; i++) if (ok[i - 1] > ok[i]) { sorted = false; break; }
    printf("SB_RB %d SB_BS %d SB_IPT %d n %zu bits %u: %.1f us sorted %d tmp %zu\n", SB_RB, SB_BS, SB_IPT, n, bits, best * 1e3f, (int)sorted, need);
    return 0;
}
