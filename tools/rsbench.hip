// tools/rsbench.hip -- rocprim radix-sort configurations at the sizes of one detection (hipcc --offload-arch=gfx950 -O2 tools/rsbench.hip -o /tmp/rsbench; run on the GPU box).
// Results kept in profiles/r04/r04j_radix_configs.txt; the chosen ones are DemeRadixCfg in csrc/deme_hip.hip.
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <cstdio>
#include <vector>
#include <random>
template <class Cfg>
float run(const char* name, uint32_t* k0, uint32_t* k1, uint32_t* v0, uint32_t* v1, size_t n, unsigned bits) {
  size_t need = 0;
  rocprim::radix_sort_pairs<Cfg>(nullptr, need, k0, k1, v0, v1, n, 0, bits, (hipStream_t)0);
  void* tmp; hipMalloc(&tmp, need);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; i++) rocprim::radix_sort_pairs<Cfg>(tmp, need, k0, k1, v0, v1, n, 0, bits, (hipStream_t)0);
  hipEventRecord(a);
  for (int i = 0; i < 20; i++) rocprim::radix_sort_pairs<Cfg>(tmp, need, k0, k1, v0, v1, n, 0, bits, (hipStream_t)0);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  printf("%-28s n=%zu bits=%u: %.1f us\n", name, n, bits, ms * 1000 / 20);
  hipFree(tmp);
  return ms;
}
template <unsigned B, unsigned I, unsigned RB>
using C = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
      rocprim::radix_sort_onesweep_config<rocprim::kernel_config<B, I>, rocprim::kernel_config<B, I>, RB, rocprim::block_radix_rank_algorithm::match>, 1024*1024>;
int main() {
  for (int coherent = 0; coherent < 1; coherent++)
  for (auto cfg : {std::pair<size_t, unsigned>{4400000, 21}, {1400000, 20}}) {
    size_t n = cfg.first; unsigned bits = cfg.second;
    std::vector<uint32_t> h(n); std::mt19937 g(1);
    for (size_t i = 0; i < n; i++) h[i] = coherent ? (uint32_t)(((i * (1ull << bits)) / n + (g() & 1023)) & ((1u << bits) - 1)) : (g() & ((1u << bits) - 1));
    uint32_t *k0, *k1, *v0, *v1;
    hipMalloc(&k0, n * 4); hipMalloc(&k1, n * 4); hipMalloc(&v0, n * 4); hipMalloc(&v1, n * 4);
    hipMemcpy(k0, h.data(), n * 4, hipMemcpyHostToDevice);
    printf("coherent=%d\n", coherent);
    run<rocprim::default_config>("default", k0, k1, v0, v1, n, bits);
    run<C<1024, 8, 8>>("1024x8 rb8", k0, k1, v0, v1, n, bits);
    run<C<1024, 6, 8>>("1024x6 rb8", k0, k1, v0, v1, n, bits);
    run<C<1024, 4, 8>>("1024x4 rb8", k0, k1, v0, v1, n, bits);
    run<C<512, 8, 8>>("512x8 rb8", k0, k1, v0, v1, n, bits);
    run<C<512, 6, 8>>("512x6 rb8", k0, k1, v0, v1, n, bits);
    run<C<512, 4, 8>>("512x4 rb8", k0, k1, v0, v1, n, bits);
    run<C<256, 8, 8>>("256x8 rb8", k0, k1, v0, v1, n, bits);
    run<C<256, 6, 8>>("256x6 rb8", k0, k1, v0, v1, n, bits);
    run<C<256, 4, 8>>("256x4 rb8", k0, k1, v0, v1, n, bits);
    run<C<1024, 6, 10>>("1024x6 rb10", k0, k1, v0, v1, n, bits);
    run<C<1024, 8, 10>>("1024x8 rb10", k0, k1, v0, v1, n, bits);
    run<C<1024, 10, 10>>("1024x10 rb10", k0, k1, v0, v1, n, bits);
    run<C<1024, 6, 7>>("1024x6 rb7", k0, k1, v0, v1, n, bits);
    run<C<1024, 8, 7>>("1024x8 rb7", k0, k1, v0, v1, n, bits);
    run<C<1024, 8, 6>>("1024x8 rb6", k0, k1, v0, v1, n, bits);
    run<C<512, 8, 7>>("512x8 rb7", k0, k1, v0, v1, n, bits);
    hipFree(k0); hipFree(k1); hipFree(v0); hipFree(v1);
  }
}
