// valu_rate.hip -- issue cost of the VALU / LDS instructions the force pass is made of, on gfx950, per wave-instruction and SIMD,
// at 1 / 2 / 4 / 8 wavefronts per SIMD.  Independent streams of 8 registers (no dependent chains inside a stream of 8).
// build: hipcc -O3 --offload-arch=gfx950 valu_rate.hip -o valu_rate ; run: ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int OP>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    float a[8], b = seed, c = seed * 0.5f;
    double d[8], e = (double)seed;
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f p[8], q = {seed, seed};
    __shared__ float4 lds[1024];
    lds[threadIdx.x] = make_float4(seed, seed, seed, seed);
    __syncthreads();
    uint32_t addr = (threadIdx.x * 16u) & 0x3FFFu;
    float4 l4 = make_float4(0, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 8; i++)
        a[i] = seed + i, d[i] = seed + i, p[i] = v2f{seed + i, seed - i};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (OP == 0) {
#define X(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
                REP8(X)
#undef X
            } else if (OP == 1) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(q), "v"(q));
                REP8(X)
#undef X
            } else if (OP == 2) {
#define X(i) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d[i]) : "v"(e), "v"(e));
                REP8(X)
#undef X
            } else if (OP == 3) {
#define X(i) asm volatile("v_add_f64 %0, %1, %0" : "+v"(d[i]) : "v"(e));
                REP8(X)
#undef X
            } else if (OP == 4) {
#define X(i) asm volatile("v_mul_f64 %0, %1, %0" : "+v"(d[i]) : "v"(e));
                REP8(X)
#undef X
            } else if (OP == 5) {
#define X(i) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(a[i]));
                REP8(X)
#undef X
            } else if (OP == 6) {
#define X(i) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a[i]) : "v"(d[i]));
                REP8(X)
#undef X
            } else if (OP == 7) {
#define X(i) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(b));
                REP8(X)
#undef X
            } else if (OP == 8) {
#define X(i) asm volatile("v_rcp_f32 %0, %1" : "=v"(a[i]) : "v"(b));
                REP8(X)
#undef X
            } else if (OP == 9) {
#define X(i) asm volatile("v_sqrt_f32 %0, %1" : "=v"(a[i]) : "v"(b));
                REP8(X)
#undef X
            } else if (OP == 10) {
#define X(i) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
                REP8(X)
#undef X
            } else if (OP == 11) {
#define X(i) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(p[i]) : "v"(q));
                REP8(X)
#undef X
            } else if (OP == 12) {
#define X(i) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(p[i]) : "v"(q));
                REP8(X)
#undef X
            } else if (OP == 13) {  // dependent chain of v_fma_f32
#define X(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[0]) : "v"(b), "v"(c));
                REP8(X)
#undef X
            } else if (OP == 14) {  // ds_read_b128, conflict-free (consecutive 16-byte cells)
#define X(i) asm volatile("ds_read_b128 %0, %1 offset:" #i "*16\n s_waitcnt lgkmcnt(0)" : "=v"(l4) : "v"(addr));
                REP8(X)
#undef X
            } else if (OP == 15) {  // ds_read_b128, eight in flight
#define X(i) asm volatile("ds_read_b128 %0, %1 offset:" #i "*16" : "=v"(l4) : "v"(addr));
                REP8(X)
#undef X
                asm volatile("s_waitcnt lgkmcnt(0)");
            } else if (OP == 16) {  // ds_read_b64 x8 in flight
                float2 l2;
#define X(i) asm volatile("ds_read_b64 %0, %1 offset:" #i "*8" : "=v"(l2) : "v"(addr));
                REP8(X)
#undef X
                asm volatile("s_waitcnt lgkmcnt(0)");
                l4.x += l2.x;
            } else if (OP == 17) {  // v_cndmask
#define X(i) asm volatile("v_cndmask_b32 %0, %1, %0, vcc" : "+v"(a[i]) : "v"(b));
                REP8(X)
#undef X
            } else if (OP == 18) {  // 4 fma + 4 f64 fma interleaved
#define X(i) asm volatile("v_fma_f32 %0, %2, %3, %0\n v_fma_f64 %1, %4, %4, %1" : "+v"(a[i]), "+v"(d[i]) : "v"(b), "v"(c), "v"(e));
                REP8(X)
#undef X
            }
        }
    }
    float s = l4.x + l4.y + l4.z + l4.w;
#pragma unroll
    for (int i = 0; i < 8; i++)
        s += a[i] + (float)d[i] + p[i].x + p[i].y;
    if (s == 123.456f)
        out[threadIdx.x] = s;
}

template <int OP>
void run(const char* name, int per_iter_mult = 1) {
    float* out;
    hipMalloc(&out, 4096);
    const int iters = 20000;
    for (int wps : {1, 2, 4, 8}) {
        dim3 grid(256 * wps), block(256);  // 256 threads = one wavefront on each SIMD; wps blocks per CU
        hipEvent_t e0, e1;
        hipEventCreate(&e0), hipEventCreate(&e1);
        hipLaunchKernelGGL(k<OP>, grid, block, 0, 0, out, 100, 1.0f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<OP>, grid, block, 0, 0, out, iters, 1.0f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double insts_per_simd = (double)iters * 32.0 * per_iter_mult * wps;  // wave-instructions issued on one SIMD
        const double ns_per = ms * 1e6 / insts_per_simd;
        printf("%-28s waves/SIMD %d: %8.3f ms  %.3f ns per wave-instruction per SIMD  (= %.2f cycles at 2.4 GHz)\n", name, wps, ms, ns_per,
               ns_per * 2.4);
    }
    hipFree(out);
}

int main() {
    run<0>("v_fma_f32");
    run<10>("v_mul_f32");
    run<1>("v_pk_fma_f32");
    run<11>("v_pk_mul_f32");
    run<12>("v_pk_add_f32");
    run<2>("v_fma_f64");
    run<3>("v_add_f64");
    run<4>("v_mul_f64");
    run<5>("v_cvt_f64_f32");
    run<6>("v_cvt_f32_f64");
    run<7>("v_mov_b32");
    run<17>("v_cndmask_b32");
    run<8>("v_rcp_f32");
    run<9>("v_sqrt_f32");
    run<13>("v_fma_f32 dependent");
    run<18>("v_fma_f32+v_fma_f64 pair", 2);
    run<14>("ds_read_b128 serial");
    run<15>("ds_read_b128 x8");
    run<16>("ds_read_b64 x8");
    return 0;
}
