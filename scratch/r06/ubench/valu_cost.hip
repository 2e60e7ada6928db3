// Cycles per VALU instruction for ONE wave per SIMD on gfx950 (the row team of k_occ_mlp_bwd_ws is exactly that): a stream of
// 16 independent instructions of one kind, repeated; s_memtime around it.  Build: hipcc -O3 --offload-arch=gfx950 valu_cost.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>
#include <cstdlib>

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(512) void k(long long* out, int iters, float seed) {
    float a[16], b[16];
    f32x2 p[16], q[16];
    unsigned u[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        a[i] = seed + i + threadIdx.x;
        b[i] = seed * 0.5f + i;
        p[i] = f32x2{a[i], b[i]};
        q[i] = f32x2{b[i], a[i]};
        u[i] = (unsigned)(threadIdx.x * 17 + i);
    }
    const float m = seed * 1.0001f;
    const f32x2 m2 = {m, m * 0.5f};
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if constexpr (KIND == 0) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(b[i]));
            REP16(X)
#undef X
        } else if constexpr (KIND == 1) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(m2), "v"(q[i]));
            REP16(X)
#undef X
        } else if constexpr (KIND == 2) {
#define X(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(m2));
            REP16(X)
#undef X
        } else if constexpr (KIND == 3) {
#define X(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(m2));
            REP16(X)
#undef X
        } else if constexpr (KIND == 4) {
#define X(i) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[i]) : "v"(a[i]), "v"(b[i]));
            REP16(X)
#undef X
        } else if constexpr (KIND == 5) {
#define X(i) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(u[i]));
            REP16(X)
#undef X
        } else if constexpr (KIND == 6) {
#define X(i) asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(u[i]) : "v"(u[(i + 1) & 15]));
            REP16(X)
#undef X
        } else if constexpr (KIND == 7) {
#define X(i) asm volatile("v_cmp_lt_f32 vcc, 0, %1\n\tv_cndmask_b32 %0, 0, %2, vcc" : "=v"(a[i]) : "v"(b[i]), "v"(b[(i + 1) & 15]) : "vcc");
            REP16(X)
#undef X
        } else if constexpr (KIND == 8) {
#define X(i) asm volatile("s_nop 0\n\tv_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
            REP16(X)
#undef X
        } else if constexpr (KIND == 9) {
#define X(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            REP16(X)
#undef X
        } else if constexpr (KIND == 10) {
#define X(i) asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(u[i]) : "v"(m), "v"(b[i]));
            REP16(X)
#undef X
        } else if constexpr (KIND == 11) {
#define X(i) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(u[i]) : "v"(m));
            REP16(X)
#undef X
        } else if constexpr (KIND == 12) {      // dependent chain of v_fma_f32
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(m), "v"(b[i]));
            REP16(X)
#undef X
        } else if constexpr (KIND == 13) {      // dependent chain of v_pk_fma_f32
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[0]) : "v"(m2), "v"(q[i]));
            REP16(X)
#undef X
        } else if constexpr (KIND == 14) {      // v_max_f32 with constant (relu)
#define X(i) asm volatile("v_max_f32 %0, 0, %0" : "+v"(a[i]));
            REP16(X)
#undef X
        } else if constexpr (KIND == 15) {      // v_fma_mix_f32: f32 = f16/f32 mix (bf16 is NOT a mix type on gfx950)
#define X(i) asm volatile("v_fma_mix_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(u[i]), "v"(m));
            REP16(X)
#undef X
        } else if constexpr (KIND == 16) {      // v_perm_b32
#define X(i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(u[(i + 1) & 15]), "v"(m));
            REP16(X)
#undef X
        } else if constexpr (KIND == 17) {      // scalar-float pair: 2 x v_fma_f32 replacing one v_pk_fma_f32
#define X(i) asm volatile("v_fma_f32 %0, %0, %2, %3\n\tv_fma_f32 %1, %1, %2, %3" : "+v"(a[i]), "+v"(b[i]) : "v"(m), "v"(m));
            REP16(X)
#undef X
        } else if constexpr (KIND == 18) {      // v_mul_f32 with DPP row broadcast? (quad_perm)
#define X(i) asm volatile("s_nop 0\n\tv_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
            REP16(X)
#undef X
        } else if constexpr (KIND == 19) {      // v_dot2c_f32_bf16: acc += a.lo*b.lo + a.hi*b.hi (bf16 pairs)
#define X(i) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(a[i]) : "v"(u[i]), "v"(u[(i + 1) & 15]));
            REP16(X)
#undef X
        } else if constexpr (KIND == 20) {      // a mix the row team issues: mul, and, fma, cvt_pk, cndmask in turn
#define X(i) asm volatile("v_mul_f32 %0, %0, %3\n\tv_and_b32 %2, 0xffff0000, %2\n\tv_fma_f32 %1, %1, %3, %3\n\tv_cvt_pk_bf16_f32 %2, %0, %1" : "+v"(a[i]), "+v"(b[i]), "+v"(u[i]) : "v"(m));
            REP16(X)
#undef X
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    unsigned us = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) { s += a[i] + b[i] + p[i].x + p[i].y + q[i].x; us ^= u[i]; }
    if (s == 12345.678f && us == 77) out[4095] = 1;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

static int g_threads = 256, g_blocks = 256;
template <int KIND>
void run(const char* name, int per_iter, long long* d) {
    const int iters = 100000;
    std::vector<long long> h(4096);
    for (int r = 0; r < 3; ++r) {
        hipLaunchKernelGGL(k<KIND>, dim3(g_blocks), dim3(g_threads), 0, 0, d, iters, 1.0f);
        hipDeviceSynchronize();
    }
    hipMemcpy(h.data(), d, 4096 * sizeof(long long), hipMemcpyDeviceToHost);
    double s = 0;
    const int nw = g_blocks * (g_threads / 64);
    for (int i = 0; i < nw; ++i) s += (double)h[i];
    printf("%-42s %6.2f cycles per instruction (%d per iteration)\n", name, s / nw / iters / per_iter, per_iter);
}

int main(int argc, char** argv) {
    if (argc > 2) { g_threads = atoi(argv[1]); g_blocks = atoi(argv[2]); }
    printf("%d threads per workgroup, %d workgroups\n", g_threads, g_blocks);
    long long* d;
    hipMalloc(&d, 4096 * sizeof(long long));
    run<0>("v_fma_f32 (independent)", 16, d);
    run<12>("v_fma_f32 (dependent chain)", 16, d);
    run<9>("v_mul_f32", 16, d);
    run<14>("v_max_f32", 16, d);
    run<17>("2 x v_fma_f32 (pair)", 32, d);
    run<1>("v_pk_fma_f32 (independent)", 16, d);
    run<13>("v_pk_fma_f32 (dependent chain)", 16, d);
    run<2>("v_pk_mul_f32", 16, d);
    run<3>("v_pk_add_f32", 16, d);
    run<4>("v_cvt_pk_bf16_f32", 16, d);
    run<5>("v_and_b32 (literal)", 16, d);
    run<6>("v_lshlrev_b32", 16, d);
    run<7>("v_cmp_lt_f32 + v_cndmask_b32", 32, d);
    run<8>("s_nop + v_add_f32_dpp row_shr", 32, d);
    run<18>("s_nop + v_mov_b32_dpp quad_perm", 32, d);
    run<10>("v_pk_fma_f16", 16, d);
    run<11>("v_pk_max_i16", 16, d);
    run<15>("v_fma_mix_f32", 16, d);
    run<16>("v_perm_b32", 16, d);
    run<19>("v_dot2c_f32_bf16", 16, d);
    run<20>("mix: mul, and, fma, cvt_pk", 64, d);
    return 0;
}
