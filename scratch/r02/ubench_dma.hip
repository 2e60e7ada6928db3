// How fast can nload waves of a CU stream a 75-KB tile into LDS with global_load_lds_dwordx4?
// pattern 0: contiguous 1 KiB per instruction; pattern 1: 8 x 128-B segments 3072 B apart (planar fp32 tile rows)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int PATTERN>
__global__ __launch_bounds__(1024) void k_dma(const float* src, long long* cycles, int nload, int tiles, size_t tile_stride) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave >= nload) return;
    const int total = 3 * 196 * 8;       // 16-byte chunks of a 75264-byte tile
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int t = 0; t < tiles; ++t) {
        const float* base = src + ((size_t)blockIdx.x * tiles + t) * tile_stride;
        float* dst = reinterpret_cast<float*>(smem + (t & 1) * 75264);
        int pi = 0, pk = (wave * 64 + lane) / 8;
        const int jc = lane & 7;
        for (int q0 = wave * 64; q0 < total; q0 += 64 * nload) {
            if (q0 + lane < total) {
                const float* g = PATTERN == 0 ? base + (size_t)(q0 + lane) * 4 : base + (size_t)pk * 768 + pi * 32 + jc * 4;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                                 (__attribute__((address_space(3))) void*)(dst + (size_t)q0 * 4), 16, 0, 0);
            }
            pk += 8 * nload;
            if (pk >= 196) { pk -= 196; pi += 1; }
        }
        __builtin_amdgcn_s_waitcnt(0);
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0 && wave == 0) cycles[blockIdx.x] = t1 - t0;
}

int main() {
    const int tiles = 8, blocks = 256;
    const size_t tile_stride = 196 * 768;         // floats: one (viewpoint, camera) slab holds 8 heads' tiles
    float* src; CHECK(hipMalloc(&src, (size_t)blocks * tiles * tile_stride * 4));
    CHECK(hipMemset(src, 0, (size_t)blocks * tiles * tile_stride * 4));
    long long* cyc; CHECK(hipMalloc(&cyc, blocks * 8));
    CHECK(hipFuncSetAttribute((const void*)k_dma<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 150528));
    CHECK(hipFuncSetAttribute((const void*)k_dma<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 150528));
    for (int pat = 0; pat < 2; ++pat)
        for (int nl : {1, 2, 4, 8, 16})
            for (int nb : {1, 256}) {
                for (int rep = 0; rep < 2; ++rep) {
                    if (pat == 0) hipLaunchKernelGGL(k_dma<0>, dim3(nb), dim3(1024), 150528, 0, src, cyc, nl, tiles, tile_stride);
                    else hipLaunchKernelGGL(k_dma<1>, dim3(nb), dim3(1024), 150528, 0, src, cyc, nl, tiles, tile_stride);
                    CHECK(hipDeviceSynchronize());
                }
                long long h[256]; CHECK(hipMemcpy(h, cyc, nb * 8, hipMemcpyDeviceToHost));
                double s = 0; for (int i = 0; i < nb; ++i) s += h[i];
                printf("pattern %d  loaders %2d  workgroups %3d : %.0f cycles per 75-KB tile (%.1f B/cycle/CU)\n", pat, nl, nb,
                       s / nb / tiles, 75264.0 / (s / nb / tiles));
            }
    return 0;
}
