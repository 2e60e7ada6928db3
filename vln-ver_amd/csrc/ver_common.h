// Shared device helpers for the gfx950 kernels of the VER lifting path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/ver_ops.h"

#define VER_WAVE 64

int ver_fail(int code, const char* fmt, ...);           // ver_abi.hip
int ver_check_launch(const char* what);                 // ver_abi.hip
// Zero `bytes` bytes (a multiple of 4, 4-byte aligned) at `ptr` with a KERNEL on `stream` (ver_abi.hip).  Used instead of
// hipMemsetAsync wherever a launcher may run under stream capture: in a replayed hipGraph (ROCm 7.2) the memset NODE of
// ver_sca_backward did not reliably precede the kernel that accumulates into the buffer with atomics -- from the second
// replay on the atomics met the previous replay's bytes (scratch/r06/graph_nan.py: non-finite d(offsets) of voxels seen by
// two cameras).  A kernel node keeps the stream order.
int ver_zero_async(void* ptr, size_t bytes, void* stream);

#define VER_REQUIRE(cond, code, ...)                     \
    do {                                                 \
        if (!(cond)) return ver_fail(code, __VA_ARGS__); \
    } while (0)

// Bilinear footprint of one sample on an H x W map, mmcv / Deformable-DETR convention
// (SURVEY.md A.3): pixel coords x = loc_x*W - 0.5, y = loc_y*H - 0.5; the sample contributes
// only if -1 < x < W and -1 < y < H; corners outside [0,W-1]x[0,H-1] contribute zero.
struct Bilinear {
    float w[4];   // weights of (y0,x0) (y0,x1) (y1,x0) (y1,x1); 0 for corners outside the map
    int key[4];   // y*W + x of each corner, clamped into the map (safe to load from)
    float gx[4];  // d w[k] / d x   (0 for invalid corners)
    float gy[4];  // d w[k] / d y
    bool any;
};

// Branch-free on purpose: with an early return for samples outside the map the struct's arrays were
// kept in scratch memory by the compiler (k_sca_bwd_off ran 72 B of private segment per lane).
template <bool GRAD>
__device__ __forceinline__ void bilinear_setup(float loc_x, float loc_y, int H, int W, Bilinear& s) {
    const float x = loc_x * (float)W - 0.5f;
    const float y = loc_y * (float)H - 0.5f;
    const bool any = (y > -1.0f) && (x > -1.0f) && (y < (float)H) && (x < (float)W);
    s.any = any;
    const float xf = floorf(x), yf = floorf(y);
    // clamp before the int conversion: far-away samples (|x| ~ 1e9) must not overflow it
    const int x0 = (int)fminf(fmaxf(xf, -2.0f), (float)W), y0 = (int)fminf(fmaxf(yf, -2.0f), (float)H);
    const int x1 = x0 + 1, y1 = y0 + 1;
    const float lx = x - xf, ly = y - yf;
    const float hx = 1.0f - lx, hy = 1.0f - ly;
    const bool vx0 = x0 >= 0 && x0 <= W - 1, vx1 = x1 >= 0 && x1 <= W - 1;
    const bool vy0 = y0 >= 0 && y0 <= H - 1, vy1 = y1 >= 0 && y1 <= H - 1;
    const int cx0 = min(max(x0, 0), W - 1), cx1 = min(max(x1, 0), W - 1);
    const int cy0 = min(max(y0, 0), H - 1), cy1 = min(max(y1, 0), H - 1);
    const bool v00 = any && vy0 && vx0, v01 = any && vy0 && vx1, v10 = any && vy1 && vx0, v11 = any && vy1 && vx1;
    s.w[0] = v00 ? hy * hx : 0.0f;
    s.w[1] = v01 ? hy * lx : 0.0f;
    s.w[2] = v10 ? ly * hx : 0.0f;
    s.w[3] = v11 ? ly * lx : 0.0f;
    s.key[0] = any ? cy0 * W + cx0 : 0;
    s.key[1] = any ? cy0 * W + cx1 : 0;
    s.key[2] = any ? cy1 * W + cx0 : 0;
    s.key[3] = any ? cy1 * W + cx1 : 0;
    if (GRAD) {
        s.gx[0] = v00 ? -hy : 0.0f;
        s.gx[1] = v01 ? hy : 0.0f;
        s.gx[2] = v10 ? -ly : 0.0f;
        s.gx[3] = v11 ? ly : 0.0f;
        s.gy[0] = v00 ? -hx : 0.0f;
        s.gy[1] = v01 ? -lx : 0.0f;
        s.gy[2] = v10 ? hx : 0.0f;
        s.gy[3] = v11 ? lx : 0.0f;
    }
}

// Trilinear footprint of one sample on a D x H x W volume (5-D grid_sample semantics of the
// detection decoder's op, voxel_temporal_self_attention.py:301-323: pixel = loc*size - 0.5,
// zero padding).  Corner k = dz*4 + dy*2 + dx.
struct Trilinear {
    float w[8];
    int key[8];
    float gx[8], gy[8], gz[8];
    bool any;
};

template <bool GRAD>
__device__ __forceinline__ void trilinear_setup(float loc_x, float loc_y, float loc_z, int D, int H, int W,
                                                Trilinear& s) {
    const float x = loc_x * (float)W - 0.5f, y = loc_y * (float)H - 0.5f, z = loc_z * (float)D - 0.5f;
    s.any = (x > -1.0f) && (y > -1.0f) && (z > -1.0f) && (x < (float)W) && (y < (float)H) && (z < (float)D);
    if (!s.any) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            s.w[k] = 0.0f;
            s.key[k] = 0;
            if (GRAD) s.gx[k] = s.gy[k] = s.gz[k] = 0.0f;
        }
        return;
    }
    const float xf = floorf(x), yf = floorf(y), zf = floorf(z);
    const int x0 = (int)xf, y0 = (int)yf, z0 = (int)zf;
    const float lx = x - xf, ly = y - yf, lz = z - zf;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int dx = k & 1, dy = (k >> 1) & 1, dz = k >> 2;
        const int xi = x0 + dx, yi = y0 + dy, zi = z0 + dz;
        const bool ok = xi >= 0 && xi <= W - 1 && yi >= 0 && yi <= H - 1 && zi >= 0 && zi <= D - 1;
        const float wx = dx ? lx : 1.0f - lx, wy = dy ? ly : 1.0f - ly, wz = dz ? lz : 1.0f - lz;
        s.w[k] = ok ? wx * wy * wz : 0.0f;
        const int cx = min(max(xi, 0), W - 1), cy = min(max(yi, 0), H - 1), cz = min(max(zi, 0), D - 1);
        s.key[k] = (cz * H + cy) * W + cx;
        if (GRAD) {
            s.gx[k] = ok ? (dx ? wy * wz : -wy * wz) : 0.0f;
            s.gy[k] = ok ? (dy ? wx * wz : -wx * wz) : 0.0f;
            s.gz[k] = ok ? (dz ? wx * wy : -wx * wy) : 0.0f;
        }
    }
}

// Cross-lane butterflies inside aligned groups of G lanes.  For G <= 16 everything stays in
// the VALU through DPP (a row = 16 lanes): xor-1 and xor-2 are quad permutes, then
// row_half_mirror joins the two quads of each 8, row_mirror the two halves of the row -- after
// log2(G) steps every lane of the group holds the result.  No LDS traffic (ds_bpermute) at all.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
constexpr int kDppXor1 = 0xB1;         // quad_perm:[1,0,3,2]
constexpr int kDppXor2 = 0x4E;         // quad_perm:[2,3,0,1]
constexpr int kDppHalfMirror = 0x141;  // row_half_mirror
constexpr int kDppMirror = 0x140;      // row_mirror

template <int G>
__device__ __forceinline__ float group_sum(float v) {
    static_assert(G == 1 || G == 2 || G == 4 || G == 8 || G == 16 || G == 32 || G == 64, "group size");
    if constexpr (G >= 2) v += dpp_mov<kDppXor1>(v);
    if constexpr (G >= 4) v += dpp_mov<kDppXor2>(v);
    if constexpr (G >= 8) v += dpp_mov<kDppHalfMirror>(v);
    if constexpr (G >= 16) v += dpp_mov<kDppMirror>(v);
    if constexpr (G >= 32) v += __shfl_xor(v, 16, VER_WAVE);
    if constexpr (G >= 64) v += __shfl_xor(v, 32, VER_WAVE);
    return v;
}

template <int G>
__device__ __forceinline__ float group_max(float v) {
    static_assert(G == 1 || G == 2 || G == 4 || G == 8 || G == 16, "group size");
    if constexpr (G >= 2) v = fmaxf(v, dpp_mov<kDppXor1>(v));
    if constexpr (G >= 4) v = fmaxf(v, dpp_mov<kDppXor2>(v));
    if constexpr (G >= 8) v = fmaxf(v, dpp_mov<kDppHalfMirror>(v));
    if constexpr (G >= 16) v = fmaxf(v, dpp_mov<kDppMirror>(v));
    return v;
}

__device__ __forceinline__ float bf16_to_f32(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
