// Which lanes does ds_read_b64 serve together on gfx950?  Per-lane address tables, 16 waves / CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>
#include <functional>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
constexpr int ITER = 4000;

template <int WIDTH>
__global__ __launch_bounds__(256) void k_probe(const unsigned* table, float* out) {
    extern __shared__ unsigned char smem[];
    for (int i = threadIdx.x; i < 40000 / 4; i += 256) ((unsigned*)smem)[i] = i;
    __syncthreads();
    const unsigned a = table[threadIdx.x & 63];
    unsigned acc = 0;
    for (int it = 0; it < ITER; ++it) {
        if constexpr (WIDTH == 8) {
            unsigned long long v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("ds_read_b64 %0, %1" : "=v"(v[j]) : "v"(a));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int j = 0; j < 8; ++j) acc ^= (unsigned)v[j];
        } else {
            typedef unsigned u4 __attribute__((ext_vector_type(4)));
            u4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("ds_read_b128 %0, %1" : "=v"(v[j]) : "v"(a));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int j = 0; j < 8; ++j) acc ^= v[j].x;
        }
    }
    if (acc == 0x1234567u) out[0] = 1.0f;
}

int main() {
    float* out; CHECK(hipMalloc(&out, 1024));
    unsigned* dtab; CHECK(hipMalloc(&dtab, 256));
    CHECK(hipFuncSetAttribute((const void*)k_probe<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 40000));
    CHECK(hipFuncSetAttribute((const void*)k_probe<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 40000));
    auto run = [&](const char* name, std::function<unsigned(int)> f, int width = 8) {
        unsigned tab[64];
        for (int l = 0; l < 64; ++l) tab[l] = f(l);
        CHECK(hipMemcpy(dtab, tab, 256, hipMemcpyHostToDevice));
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        auto launch = [&] {
            if (width == 8) hipLaunchKernelGGL(k_probe<8>, dim3(1024), dim3(256), 40000, 0, dtab, out);
            else hipLaunchKernelGGL(k_probe<16>, dim3(1024), dim3(256), 40000, 0, dtab, out);
        };
        launch(); launch(); CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        for (int i = 0; i < 3; ++i) launch();
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
        printf("%-64s %7.3f ms  %.2f (x bcast)\n", name, ms, 0.0);
        fflush(stdout);
        return ms;
    };
    run("bcast: all lanes addr 0", [](int l) { return 0u; });
    run("linear: lane*8", [](int l) { return (unsigned)l * 8; });
    run("linear within {0-15,32-47},{16-31,48-63}", [](int l) { int g = (l >> 4) & 1; int k = (l & 15) + ((l >> 5) << 4); return (unsigned)(g * 256 + k * 8); });
    run("linear within 8-lane interleave {0-7,16-23,32-39,48-55},..", [](int l) { int g = (l >> 3) & 1; int k = (l & 7) + ((l >> 4) << 3); return (unsigned)(g * 256 + k * 8); });
    run("linear within even/odd lanes", [](int l) { int g = l & 1; int k = l >> 1; return (unsigned)(g * 256 + k * 8); });
    // slot patterns: slot = (l>>3)&3, l8 = l&7, half = l>>5; rows of 192 B
    auto slotrow = [](int l, int base0, int base1, const int* dr) { int slot = (l >> 3) & 3, half = l >> 5, l8 = l & 7; int r = (half ? base1 : base0) + dr[slot]; return (unsigned)(r * 192 + l8 * 8); };
    static const int corner[4] = {0, 1, 14, 15};
    static const int same[4] = {0, 4, 8, 12};      // 4 rows in the SAME quarter: 4-way conflict expected
    static const int two[4] = {0, 4, 1, 5};        // 2-way
    run("slots rows (0,1,14,15), both halves same base", [&](int l) { return slotrow(l, 0, 0, corner); });
    run("slots rows (0,1,14,15), half1 base +4 (same quarters)", [&](int l) { return slotrow(l, 0, 4, corner); });
    run("slots rows (0,1,14,15), half1 base +1", [&](int l) { return slotrow(l, 0, 1, corner); });
    run("slots rows (0,1,14,15), half1 base +2", [&](int l) { return slotrow(l, 0, 2, corner); });
    run("slots rows (0,1,14,15), half1 base +3", [&](int l) { return slotrow(l, 0, 3, corner); });
    run("slots rows (0,1,14,15), half1 base +50", [&](int l) { return slotrow(l, 0, 50, corner); });
    run("slots rows (0,4,8,12) 4-way, halves same", [&](int l) { return slotrow(l, 0, 0, same); });
    run("slots rows (0,4,1,5) 2-way, halves same", [&](int l) { return slotrow(l, 0, 0, two); });
    run("slots rows (0,4,1,5) 2-way, half1 base +16", [&](int l) { return slotrow(l, 0, 16, two); });
    // +offset 64 / 128 (the other two vectors of the row)
    run("slots rows (0,1,14,15) +64 B, half1 base +7", [&](int l) { return slotrow(l, 0, 7, corner) + 64; });
    // b128 probes: 8 lanes x 16 B = 128 B per slot
    run("b128 bcast", [](int l) { return 0u; }, 16);
    run("b128 linear lane*16", [](int l) { return (unsigned)l * 16; }, 16);
    auto slot128 = [](int l, int base0, int base1, const int* dr, int stride) { int slot = (l >> 3) & 3, half = l >> 5, l8 = l & 7; int r = (half ? base1 : base0) + dr[slot]; return (unsigned)(r * stride + l8 * 16); };
    run("b128 slots rows (0,1,14,15) stride 384, half1 +5", [&](int l) { return slot128(l, 0, 5, corner, 384); }, 16);
    run("b128 slots rows (0,1,14,15) stride 192(+l8*16), half1 +5", [&](int l) { return slot128(l, 0, 5, corner, 192); }, 16);
    return 0;
}
