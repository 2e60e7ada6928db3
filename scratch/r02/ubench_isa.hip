// Instruction-rate probes for the gather kernel design (gfx950).  hipcc --offload-arch=gfx950 -O3 ubench_isa.hip -o ubench_isa
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int ITER = 2000;
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int KIND>
__global__ __launch_bounds__(256) void k_valu(float* out, unsigned seed) {
    float a[8]; unsigned u[8];
    float c2[16];
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (float)(threadIdx.x + i); u[i] = seed + threadIdx.x * 7 + i; }
#pragma unroll
    for (int i = 0; i < 16; ++i) c2[i] = (float)i;
    float w = 1.0001f; unsigned wu = 0x3f803f80u;
    for (int it = 0; it < ITER; ++it) {
        if constexpr (KIND == 0) {
#define X(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(w), "v"(a[(i+1)&7]));
            REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == 1) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(*(double*)&c2[2*i]) : "v"(*(double*)&c2[(2*i+2)&15]), "v"(*(double*)&c2[(2*i+4)&15]));
            REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == 2) {
#define X(i) asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(u[i]) : "v"(u[(i+1)&7]));
            REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == 3) {
#define X(i) asm volatile("v_dot2_f32_bf16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(wu), "v"(u[i]));
            REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == 4) {
#define X(i) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(a[i]) : "v"(wu), "v"(u[i]));
            REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == 5) {
#define X(i) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(a[i]) : "v"(u[i]), "v"(w));
            REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == 6) {
#define X(i) asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(u[i]) : "v"(u[(i+1)&7]));
            REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == 7) {
#define X(i) asm volatile("v_mov_b32_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "=v"(u[i]) : "v"(u[(i+1)&7]));
            REP8(X) REP8(X)
#undef X
        }
    }
    float s = 0; unsigned t = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) { s += a[i]; t += u[i]; }
#pragma unroll
    for (int i = 0; i < 16; ++i) s += c2[i];
    if (s == 1.2345f || t == 0x12345) out[0] = s;
}

// LDS probes: 16 DS ops per loop iteration, lgkmcnt(0) once per iteration
// PAT 0: ds_read_b64, 8 lanes / slot contiguous 64 B, 4 slots of a half wave at rows r, r+1, r+14, r+15 (stride 192 B)
// PAT 1: same but the 4 slots at pseudo-random rows
// PAT 2: ds_bpermute_b32
// PAT 3: ds_read_b64 broadcast records (8 lanes same address, slots 8 B apart)
// PAT 4: ds_read_b64, all 8 slots of the wave random rows (the round-1 kernel's pattern)
template <int PAT>
__global__ __launch_bounds__(256) void k_lds(float* out, unsigned seed) {
    extern __shared__ unsigned char smem[];
    for (int i = threadIdx.x; i < 38000 / 4; i += 256) ((unsigned*)smem)[i] = i;
    __syncthreads();
    const int lane = threadIdx.x & 63, slot = (lane >> 3) & 3, half = lane >> 5, l8 = lane & 7;
    unsigned rng = seed * 2654435761u + (threadIdx.x >> 3) * 40503u;
    unsigned long long acc = 0; unsigned accu = 0;
    for (int it = 0; it < ITER; ++it) {
        unsigned addr[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            rng = rng * 1664525u + 1013904223u;
            unsigned r;
            if (PAT == 0) {
                unsigned base = __shfl((int)(rng >> 8), (lane & 32), 64) % 180u;   // one base row per half wave
                r = base + (slot & 1) + (slot >> 1) * 14;
            } else {
                r = (rng >> 8) % 196u;                                              // per-slot random row (same for the 8 lanes)
                r = __shfl((int)r, lane & ~7, 64);
            }
            addr[j] = r * 192u + l8 * 8u;
        }
        if constexpr (PAT == 0 || PAT == 1 || PAT == 4) {
            unsigned long long v[12];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                asm volatile("ds_read_b64 %0, %1" : "=v"(v[3*j]) : "v"(addr[j]));
                asm volatile("ds_read_b64 %0, %1 offset:64" : "=v"(v[3*j+1]) : "v"(addr[j]));
                asm volatile("ds_read_b64 %0, %1 offset:128" : "=v"(v[3*j+2]) : "v"(addr[j]));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int j = 0; j < 12; ++j) acc ^= v[j];
        } else if constexpr (PAT == 2) {
            unsigned v[12];
#pragma unroll
            for (int j = 0; j < 12; ++j) asm volatile("ds_bpermute_b32 %0, %1, %2" : "=v"(v[j]) : "v"((addr[j & 3] & 0xfc)), "v"(rng + j));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int j = 0; j < 12; ++j) accu ^= v[j];
        } else if constexpr (PAT == 3) {
            unsigned long long v[12];
            const unsigned ra = 37632u + half * 256u + slot * 8u;
#pragma unroll
            for (int j = 0; j < 12; ++j) asm volatile("ds_read_b64 %0, %1" : "=v"(v[j]) : "v"(ra + (j & 7) * 32u));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int j = 0; j < 12; ++j) acc ^= v[j];
        }
    }
    if (acc == 0x1234567ull || accu == 0x7654321u) out[0] = 1.0f;
}

template <typename F>
double time_ms(F&& launch) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    launch(); CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < 5; ++i) launch();
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms / 5;
}

int main() {
    float* out; CHECK(hipMalloc(&out, 1024));
    const int blocks = 256 * 4;      // 4 workgroups of 4 waves per CU = 4 waves / SIMD
    const char* vn[] = {"v_fma_f32", "v_pk_fma_f32", "v_and_b32", "v_dot2_f32_bf16", "v_dot2c_f32_bf16", "v_fma_mix_f32", "v_lshlrev_b32", "v_mov_dpp row_newbcast"};
    auto report = [&](const char* name, double ms, double insts_per_wave, double waves_per_unit, const char* unit) {
        // cycles per wave-instruction per unit at 2.4 GHz
        double cyc = ms * 1e-3 * 2.4e9 / (insts_per_wave * waves_per_unit);
        printf("%-28s %8.3f ms   %.2f cyc / wave-instr / %s (at 2.4 GHz)\n", name, ms, cyc, unit);
    };
#define RUNV(K) { double ms = time_ms([&] { hipLaunchKernelGGL(k_valu<K>, dim3(blocks), dim3(256), 0, 0, out, 1u); }); report(vn[K], ms, ITER * 16.0, 4.0, "SIMD"); }
    RUNV(0) RUNV(1) RUNV(2) RUNV(3) RUNV(4) RUNV(5) RUNV(6) RUNV(7)
    const char* ln[] = {"ds_read_b64 4-corner rows", "ds_read_b64 4 random rows/half", "ds_bpermute_b32", "ds_read_b64 bcast records", "(same as 1)"};
#define RUNL(P) { CHECK(hipFuncSetAttribute((const void*)k_lds<P>, hipFuncAttributeMaxDynamicSharedMemorySize, 39000)); \
      double ms = time_ms([&] { hipLaunchKernelGGL(k_lds<P>, dim3(blocks), dim3(256), 39000, 0, out, 1u); }); report(ln[P], ms, ITER * 12.0, 16.0, "CU"); }
    RUNL(0) RUNL(1) RUNL(2) RUNL(3)
    return 0;
}
