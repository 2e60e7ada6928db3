// Do MFMA and VALU instructions overlap on a gfx950 SIMD?  (a) from ONE wave, interleaved; (b) from two waves of a SIMD,
// one issuing only MFMAs, the other only VALU.  hipcc --offload-arch=gfx950 -O3 ubench_mfma.hip -o ubench_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int ITER = 2000;

// MODE 0: 8 MFMA / iter; 1: 32 VALU / iter; 2: both interleaved (1 MFMA : 4 VALU); 3: waves 0-3 MFMA only, waves 4-7 VALU only
template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, int nw) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(i); }
    f32x4 acc[4] = {{0,0,0,0},{0,0,0,0},{0,0,0,0},{0,0,0,0}};
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = (float)(threadIdx.x + i);
    float w = 1.0001f;
    const int wave = threadIdx.x >> 6;
    const bool do_m = MODE == 0 || MODE == 2 || (MODE == 3 && wave < 4);
    const bool do_v = MODE == 1 || MODE == 2 || (MODE == 3 && wave >= 4);
    if (MODE == 3 ? (wave < 4) : do_m && !do_v) {
        for (int it = 0; it < ITER; ++it) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
        }
    } else if (MODE == 3 ? (wave >= 4) : do_v && !do_m) {
        for (int it = 0; it < ITER; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[i]) : "v"(w), "v"(v[(i + 1) & 7]));
        }
    } else {
        for (int it = 0; it < ITER; ++it) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < 4; ++j) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[(4 * i + j) & 7]) : "v"(w), "v"(v[(4 * i + j + 1) & 7]));
                }
        }
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; ++i) s += v[i];
    if (s == 123.456f) out[threadIdx.x] = s;
}

template <typename F>
double time_ms(F f) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    f(); CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0)); for (int i = 0; i < 5; ++i) f(); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); return ms / 5;
}

int main() {
    float* out; CHECK(hipMalloc(&out, 4096));
    const int blocks = 256;   // one workgroup per CU
    const char* names[4] = {"8 MFMA/iter (1 wave/SIMD)", "32 VALU/iter (1 wave/SIMD)", "8 MFMA + 32 VALU interleaved, one wave/SIMD", "2 waves/SIMD: one MFMA-only (8/iter), one VALU-only (32/iter)"};
    double ms;
    ms = time_ms([&] { hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, 4); }); printf("%-70s %8.3f ms  %.1f cycles/iter @2.4GHz\n", names[0], ms, ms * 1e-3 * 2.4e9 / ITER);
    ms = time_ms([&] { hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, 4); }); printf("%-70s %8.3f ms  %.1f cycles/iter\n", names[1], ms, ms * 1e-3 * 2.4e9 / ITER);
    ms = time_ms([&] { hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, out, 4); }); printf("%-70s %8.3f ms  %.1f cycles/iter\n", names[2], ms, ms * 1e-3 * 2.4e9 / ITER);
    ms = time_ms([&] { hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(512), 0, 0, out, 8); }); printf("%-70s %8.3f ms  %.1f cycles/iter\n", names[3], ms, ms * 1e-3 * 2.4e9 / ITER);
    ms = time_ms([&] { hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(512), 0, 0, out, 8); }); printf("%-70s %8.3f ms  %.1f cycles/iter\n", "8 MFMA/iter, 2 waves/SIMD", ms, ms * 1e-3 * 2.4e9 / ITER);
    ms = time_ms([&] { hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(512), 0, 0, out, 8); }); printf("%-70s %8.3f ms  %.1f cycles/iter\n", "32 VALU/iter, 2 waves/SIMD", ms, ms * 1e-3 * 2.4e9 / ITER);
    ms = time_ms([&] { hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(512), 0, 0, out, 8); }); printf("%-70s %8.3f ms  %.1f cycles/iter\n", "8 MFMA + 32 VALU interleaved, 2 waves/SIMD", ms, ms * 1e-3 * 2.4e9 / ITER);
    return 0;
}
