// valu_rate2.hip -- second sheet of instruction issue costs on gfx950: selects, integer address arithmetic, compares, LDS forms.
// Same method as valu_rate.hip (8 independent registers per stream; 1 / 2 / 4 / 8 wavefronts per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int OP>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed, unsigned long long mask) {
    float a[8], b = seed, c = seed * 0.5f;
    uint32_t u[8], w = (uint32_t)seed + 3u;
    unsigned long long q[8];
    double d[8];
    __shared__ float4 lds[1024];
    lds[threadIdx.x] = make_float4(seed, seed, seed, seed);
    __syncthreads();
    uint32_t addr = (threadIdx.x * 16u) & 0x3FFFu;
    typedef float v4f __attribute__((ext_vector_type(4)));
    v4f l4 = {0.f, 0.f, 0.f, 0.f};
    typedef float v3f __attribute__((ext_vector_type(3)));
    v3f l3 = {0.f, 0.f, 0.f};
    unsigned long long sm = mask;  // an SGPR pair
#pragma unroll
    for (int i = 0; i < 8; i++)
        a[i] = seed + i, u[i] = (uint32_t)seed + i, q[i] = (unsigned long long)seed + i, d[i] = seed + i;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            if (OP == 0) {
#define X(i) asm volatile("v_cndmask_b32_e32 %0, %1, %0, vcc" : "+v"(a[i]) : "v"(b) : );
                REP8(X)
#undef X
            } else if (OP == 1) {
#define X(i) asm volatile("v_cndmask_b32_e64 %0, %1, %0, %2" : "+v"(a[i]) : "v"(b), "s"(sm));
                REP8(X)
#undef X
            } else if (OP == 2) {  // the usual pair: compare into vcc, select
#define X(i) asm volatile("v_cmp_lt_f32_e32 vcc, %1, %0\n v_cndmask_b32_e32 %0, %1, %0, vcc" : "+v"(a[i]) : "v"(b) : "vcc");
                REP8(X)
#undef X
            } else if (OP == 3) {
#define X(i) asm volatile("v_cmp_lt_f32_e32 vcc, %1, %0" : : "v"(a[i]), "v"(b) : "vcc");
                REP8(X)
#undef X
            } else if (OP == 4) {
#define X(i) asm volatile("v_max_f32_e32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
                REP8(X)
#undef X
            } else if (OP == 5) {
#define X(i) asm volatile("v_lshl_add_u32 %0, %0, 4, %1" : "+v"(u[i]) : "v"(w));
                REP8(X)
#undef X
            } else if (OP == 6) {
#define X(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[i]) : "v"(w));
                REP8(X)
#undef X
            } else if (OP == 7) {
#define X(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(u[i]) : "v"(w));
                REP8(X)
#undef X
            } else if (OP == 8) {
#define X(i) asm volatile("v_lshl_add_u64 %0, %0, 3, %1" : "+v"(q[i]) : "v"(q[(i + 1) & 7]));
                REP8(X)
#undef X
            } else if (OP == 9) {
#define X(i) asm volatile("v_bfe_u32 %0, %0, 3, 7" : "+v"(u[i]));
                REP8(X)
#undef X
            } else if (OP == 10) {
#define X(i) asm volatile("v_add_u32_e32 %0, %1, %0" : "+v"(u[i]) : "v"(w));
                REP8(X)
#undef X
            } else if (OP == 11) {
#define X(i) asm volatile("v_cvt_f64_i32_e32 %0, %1" : "=v"(d[i]) : "v"(u[i]));
                REP8(X)
#undef X
            } else if (OP == 12) {  // select on an SGPR mask written by a compare just before (e64 compare -> s[...])
#define X(i) asm volatile("v_cmp_lt_f32_e64 %1, %2, %0\n v_cndmask_b32_e64 %0, %2, %0, %1" : "+v"(a[i]), "+s"(sm) : "v"(b) : );
                REP8(X)
#undef X
            } else if (OP == 13) {
#define X(i) asm volatile("ds_read_b96 %0, %1 offset:" #i "*16" : "=v"(l3) : "v"(addr));
                REP8(X)
#undef X
                asm volatile("s_waitcnt lgkmcnt(0)");
            } else if (OP == 14) {
#define X(i) asm volatile("ds_read_b32 %0, %1 offset:" #i "*4" : "=v"(l4.x) : "v"(addr));
                REP8(X)
#undef X
                asm volatile("s_waitcnt lgkmcnt(0)");
            } else if (OP == 15) {
#define X(i) asm volatile("ds_write_b128 %1, %0 offset:" #i "*16" : : "v"(l4), "v"(addr));
                REP8(X)
#undef X
                asm volatile("s_waitcnt lgkmcnt(0)");
            } else if (OP == 16) {
#define X(i) asm volatile("ds_read_u16 %0, %1 offset:" #i "*2" : "=v"(u[i]) : "v"(addr));
                REP8(X)
#undef X
                asm volatile("s_waitcnt lgkmcnt(0)");
            } else if (OP == 17) {  // ds_read_b128 with a 112-byte lane stride (28 words: the staged-record stride)
                uint32_t ad2 = ((threadIdx.x & 63u) * 112u + (threadIdx.x >> 6) * 16u) & 0x3FFFu;
#define X(i) asm volatile("ds_read_b128 %0, %1" : "=v"(l4) : "v"(ad2));
                REP8(X)
#undef X
                asm volatile("s_waitcnt lgkmcnt(0)");
            } else if (OP == 18) {  // s_and_saveexec + s_or exec around one VALU (the cost of a tiny predicated block)
#define X(i) asm volatile("s_and_saveexec_b64 %1, %2\n v_add_f32_e32 %0, %0, %0\n s_or_b64 exec, exec, %1" : "+v"(a[i]), "=&s"(q[0]) : "s"(sm) : );
                REP8(X)
#undef X
            } else if (OP == 19) {
#define X(i) asm volatile("v_mov_b64_e32 %0, %1" : "=v"(q[i]) : "v"(q[(i + 1) & 7]));
                REP8(X)
#undef X
            } else if (OP == 20) {
#define X(i) asm volatile("v_sub_f32_e32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
                REP8(X)
#undef X
            } else if (OP == 21) {
#define X(i) asm volatile("v_rsq_f32_e32 %0, %1" : "=v"(a[i]) : "v"(b));
                REP8(X)
#undef X
            }
        }
    }
    float s = l4.x + l4.y + l4.z + l4.w + c + l3.x + l3.y + l3.z;
#pragma unroll
    for (int i = 0; i < 8; i++)
        s += a[i] + (float)u[i] + (float)q[i] + (float)d[i];
    if (s == 123.456f)
        out[threadIdx.x] = s + (float)sm;
}

template <int OP>
void run(const char* name, int per_iter_mult = 1) {
    float* out;
    (void)hipMalloc(&out, 4096);
    const int iters = 10000;
    for (int wps : {1, 2, 4, 8}) {
        dim3 grid(256 * wps), block(256);
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(k<OP>, grid, block, 0, 0, out, 100, 1.0f, 0x5555555555555555ull);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<OP>, grid, block, 0, 0, out, iters, 1.0f, 0x5555555555555555ull);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double insts_per_simd = (double)iters * 32.0 * per_iter_mult * wps;
        const double ns_per = ms * 1e6 / insts_per_simd;
        printf("%-34s waves/SIMD %d: %8.3f ms  %.3f ns per wave-instruction per SIMD  (= %.2f cycles at 2.4 GHz)\n", name, wps, ms, ns_per, ns_per * 2.4);
    }
    (void)hipFree(out);
}

int main() {
    run<0>("v_cndmask_b32_e32 (vcc)");
    run<1>("v_cndmask_b32_e64 (sgpr pair)");
    run<2>("v_cmp_lt_f32 vcc + v_cndmask", 2);
    run<3>("v_cmp_lt_f32_e32 vcc");
    run<12>("v_cmp_e64 sgpr + v_cndmask_e64", 2);
    run<4>("v_max_f32");
    run<20>("v_sub_f32");
    run<21>("v_rsq_f32");
    run<5>("v_lshl_add_u32");
    run<6>("v_mul_lo_u32");
    run<7>("v_mad_u32_u24");
    run<8>("v_lshl_add_u64");
    run<9>("v_bfe_u32");
    run<10>("v_add_u32");
    run<11>("v_cvt_f64_i32");
    run<19>("v_mov_b64");
    run<18>("saveexec+v_add+restore", 1);
    run<13>("ds_read_b96 x8");
    run<14>("ds_read_b32 x8");
    run<15>("ds_write_b128 x8");
    run<16>("ds_read_u16 x8");
    run<17>("ds_read_b128 stride 112 B");
    return 0;
}
