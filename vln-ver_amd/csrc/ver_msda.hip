// mmcv op boundary on gfx950: ver_msda_forward / ver_msda_backward (include/ver_ops.h).
//
// Replaces `_ext.ms_deform_attn_forward/backward` as called by the reference at
// bevformer/modules/multi_scale_deformable_attn_function.py:118-124,150-160 for arbitrary
// (levels, points, head_dim <= 256).  One aligned group of G lanes owns one (b, q, head)
// output row; lanes stride the head_dim channels so every corner read is one contiguous
// head_dim*4-byte segment of the value tensor.  The fused, LDS-tiled kernels the encoder
// actually runs on live in ver_sca.hip; this file is the general-shape drop-in.
#include <cstdlib>
#include "ver_common.h"

template <int G, int NC>
__global__ __launch_bounds__(256) void k_msda_fwd(const float* __restrict__ value,
                                                  const int64_t* __restrict__ shapes,
                                                  const int64_t* __restrict__ lstart,
                                                  const float* __restrict__ loc,
                                                  const float* __restrict__ aw, float* __restrict__ out,
                                                  int B, int Nk, int heads, int hd, int L, int P, int Nq) {
    const int gpb = 256 / G;
    const long gid = (long)blockIdx.x * gpb + threadIdx.x / G;
    const int lane = threadIdx.x % G;
    const long total = (long)B * Nq * heads;
    if (gid >= total) return;
    const int h = (int)(gid % heads);
    const int b = (int)(gid / heads / Nq);
    const float* lp = loc + gid * L * P * 2;
    const float* wp = aw + gid * L * P;
    const float* vb = value + (long)b * Nk * heads * hd + (long)h * hd;
    const long vstride = (long)heads * hd;
    float acc[NC];
#pragma unroll
    for (int i = 0; i < NC; ++i) acc[i] = 0.0f;
    for (int l = 0; l < L; ++l) {
        const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
        const float* vl = vb + (long)lstart[l] * vstride;
        for (int p = 0; p < P; ++p) {
            Bilinear s;
            bilinear_setup<false>(lp[(l * P + p) * 2], lp[(l * P + p) * 2 + 1], H, W, s);
            if (!s.any) continue;
            const float a = wp[l * P + p];
#pragma unroll
            for (int i = 0; i < NC; ++i) {
                const int ch = lane + i * G;
                if (ch < hd) {
                    float v = 0.0f;
#pragma unroll
                    for (int k = 0; k < 4; ++k) v += s.w[k] * vl[(long)s.key[k] * vstride + ch];
                    acc[i] += a * v;
                }
            }
        }
    }
    float* op = out + gid * hd;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int ch = lane + i * G;
        if (ch < hd) op[ch] = acc[i];
    }
}

template <int G, int NC>
__global__ __launch_bounds__(256) void k_msda_bwd(const float* __restrict__ value,
                                                  const int64_t* __restrict__ shapes,
                                                  const int64_t* __restrict__ lstart,
                                                  const float* __restrict__ loc,
                                                  const float* __restrict__ aw,
                                                  const float* __restrict__ gout, float* gvalue,
                                                  float* __restrict__ gloc, float* __restrict__ gaw,
                                                  int B, int Nk, int heads, int hd, int L, int P, int Nq) {
    const int gpb = 256 / G;
    const long gid = (long)blockIdx.x * gpb + threadIdx.x / G;
    const int lane = threadIdx.x % G;
    const long total = (long)B * Nq * heads;
    if (gid >= total) return;   // whole groups leave together, so the group shuffles stay valid
    const int h = (int)(gid % heads);
    const int b = (int)(gid / heads / Nq);
    const float* lp = loc + gid * L * P * 2;
    const float* wp = aw + gid * L * P;
    const long vstride = (long)heads * hd;
    const long voff = (long)b * Nk * vstride + (long)h * hd;
    float g[NC];
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int ch = lane + i * G;
        g[i] = ch < hd ? gout[gid * hd + ch] : 0.0f;
    }
    for (int l = 0; l < L; ++l) {
        const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
        const long lbase = voff + (long)lstart[l] * vstride;
        for (int p = 0; p < P; ++p) {
            Bilinear s;
            bilinear_setup<true>(lp[(l * P + p) * 2], lp[(l * P + p) * 2 + 1], H, W, s);
            const float a = wp[l * P + p];
            float sa = 0.0f, sx = 0.0f, sy = 0.0f;
            if (s.any) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (s.w[k] == 0.0f && s.gx[k] == 0.0f && s.gy[k] == 0.0f) continue;
                    float d = 0.0f;
#pragma unroll
                    for (int i = 0; i < NC; ++i) {
                        const int ch = lane + i * G;
                        if (ch < hd) {
                            const long idx = lbase + (long)s.key[k] * vstride + ch;
                            d += g[i] * value[idx];
                            atomicAdd(gvalue + idx, s.w[k] * a * g[i]);
                        }
                    }
                    sa += s.w[k] * d;
                    sx += s.gx[k] * d;
                    sy += s.gy[k] * d;
                }
            }
            sa = group_sum<G>(sa);
            sx = group_sum<G>(sx);
            sy = group_sum<G>(sy);
            if (lane == 0) {
                gaw[gid * L * P + l * P + p] = sa;
                gloc[(gid * L * P + l * P + p) * 2] = (float)W * a * sx;
                gloc[(gid * L * P + l * P + p) * 2 + 1] = (float)H * a * sy;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// 3-D (trilinear) variant for the detection decoder: the reference's in-tree
// voxel_multi_scale_deformable_attn_pytorch (voxel_temporal_self_attention.py:275-335) called
// from VoxelCustomMSDeformableAttention.forward (voxel_decoder.py:312-313).  shapes [L,3] =
// (D,H,W), loc [...,3] = (x,y,z).  Same lane organisation as the 2-D kernels above.
template <int G, int NC>
__global__ __launch_bounds__(256) void k_msda3d_fwd(const float* __restrict__ value,
                                                    const int64_t* __restrict__ shapes,
                                                    const int64_t* __restrict__ lstart,
                                                    const float* __restrict__ loc,
                                                    const float* __restrict__ aw, float* __restrict__ out,
                                                    int B, int Nk, int heads, int hd, int L, int P, int Nq) {
    const int gpb = 256 / G;
    const long gid = (long)blockIdx.x * gpb + threadIdx.x / G;
    const int lane = threadIdx.x % G;
    if (gid >= (long)B * Nq * heads) return;
    const int h = (int)(gid % heads);
    const int b = (int)(gid / heads / Nq);
    const float* lp = loc + gid * L * P * 3;
    const float* wp = aw + gid * L * P;
    const long vstride = (long)heads * hd;
    const float* vb = value + (long)b * Nk * vstride + (long)h * hd;
    float acc[NC];
#pragma unroll
    for (int i = 0; i < NC; ++i) acc[i] = 0.0f;
    for (int l = 0; l < L; ++l) {
        const int D = (int)shapes[3 * l], H = (int)shapes[3 * l + 1], W = (int)shapes[3 * l + 2];
        const float* vl = vb + (long)lstart[l] * vstride;
        for (int p = 0; p < P; ++p) {
            Trilinear s;
            const float* q = lp + (l * P + p) * 3;
            trilinear_setup<false>(q[0], q[1], q[2], D, H, W, s);
            if (!s.any) continue;
            const float a = wp[l * P + p];
#pragma unroll
            for (int i = 0; i < NC; ++i) {
                const int ch = lane + i * G;
                if (ch < hd) {
                    float v = 0.0f;
#pragma unroll
                    for (int k = 0; k < 8; ++k) v += s.w[k] * vl[(long)s.key[k] * vstride + ch];
                    acc[i] += a * v;
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int ch = lane + i * G;
        if (ch < hd) out[gid * hd + ch] = acc[i];
    }
}

template <int G, int NC>
__global__ __launch_bounds__(256) void k_msda3d_bwd(const float* __restrict__ value,
                                                    const int64_t* __restrict__ shapes,
                                                    const int64_t* __restrict__ lstart,
                                                    const float* __restrict__ loc,
                                                    const float* __restrict__ aw,
                                                    const float* __restrict__ gout, float* gvalue,
                                                    float* __restrict__ gloc, float* __restrict__ gaw, int B,
                                                    int Nk, int heads, int hd, int L, int P, int Nq) {
    const int gpb = 256 / G;
    const long gid = (long)blockIdx.x * gpb + threadIdx.x / G;
    const int lane = threadIdx.x % G;
    if (gid >= (long)B * Nq * heads) return;
    const int h = (int)(gid % heads);
    const int b = (int)(gid / heads / Nq);
    const float* lp = loc + gid * L * P * 3;
    const float* wp = aw + gid * L * P;
    const long vstride = (long)heads * hd;
    const long voff = (long)b * Nk * vstride + (long)h * hd;
    float g[NC];
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int ch = lane + i * G;
        g[i] = ch < hd ? gout[gid * hd + ch] : 0.0f;
    }
    for (int l = 0; l < L; ++l) {
        const int D = (int)shapes[3 * l], H = (int)shapes[3 * l + 1], W = (int)shapes[3 * l + 2];
        const long lbase = voff + (long)lstart[l] * vstride;
        for (int p = 0; p < P; ++p) {
            Trilinear s;
            const float* q = lp + (l * P + p) * 3;
            trilinear_setup<true>(q[0], q[1], q[2], D, H, W, s);
            const float a = wp[l * P + p];
            float sa = 0.0f, sx = 0.0f, sy = 0.0f, sz = 0.0f;
            if (s.any) {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    if (s.w[k] == 0.0f && s.gx[k] == 0.0f && s.gy[k] == 0.0f && s.gz[k] == 0.0f) continue;
                    float d = 0.0f;
#pragma unroll
                    for (int i = 0; i < NC; ++i) {
                        const int ch = lane + i * G;
                        if (ch < hd) {
                            const long idx = lbase + (long)s.key[k] * vstride + ch;
                            d += g[i] * value[idx];
                            atomicAdd(gvalue + idx, s.w[k] * a * g[i]);
                        }
                    }
                    sa += s.w[k] * d;
                    sx += s.gx[k] * d;
                    sy += s.gy[k] * d;
                    sz += s.gz[k] * d;
                }
            }
            sa = group_sum<G>(sa);
            sx = group_sum<G>(sx);
            sy = group_sum<G>(sy);
            sz = group_sum<G>(sz);
            if (lane == 0) {
                const long o = gid * L * P + l * P + p;
                gaw[o] = sa;
                gloc[o * 3] = (float)W * a * sx;
                gloc[o * 3 + 1] = (float)H * a * sy;
                gloc[o * 3 + 2] = (float)D * a * sz;
            }
        }
    }
}

namespace {

template <typename F>
int dispatch_group(int hd, F&& f) {
    if (hd <= 8) return f(std::integral_constant<int, 8>(), std::integral_constant<int, 1>());
    if (hd <= 16) return f(std::integral_constant<int, 16>(), std::integral_constant<int, 1>());
    if (hd <= 32) return f(std::integral_constant<int, 32>(), std::integral_constant<int, 1>());
    if (hd <= 64) return f(std::integral_constant<int, 64>(), std::integral_constant<int, 1>());
    if (hd <= 96) return f(std::integral_constant<int, 32>(), std::integral_constant<int, 3>());
    if (hd <= 128) return f(std::integral_constant<int, 64>(), std::integral_constant<int, 2>());
    if (hd <= 192) return f(std::integral_constant<int, 64>(), std::integral_constant<int, 3>());
    return f(std::integral_constant<int, 64>(), std::integral_constant<int, 4>());
}

int check_common(const void* value, const void* shapes, const void* lstart, const void* loc,
                 const void* aw, int B, int Nk, int heads, int hd, int L, int P, int Nq) {
    VER_REQUIRE(B >= 0 && Nq >= 0, VER_EINVAL, "ver_msda: negative batch/query count");
    VER_REQUIRE(Nk > 0 && heads > 0 && hd > 0 && L > 0 && P > 0, VER_EINVAL,
                "ver_msda: num_keys/heads/head_dim/levels/points must be positive");
    if (B == 0 || Nq == 0) return VER_OK;   // empty query set: buffers may legitimately be null
    VER_REQUIRE(value && shapes && lstart && loc && aw, VER_EINVAL, "ver_msda: null pointer argument");
    VER_REQUIRE(hd <= 256, VER_EUNSUPPORTED, "ver_msda: head_dim %d > 256", hd);
    return VER_OK;
}

}  // namespace

extern "C" int ver_msda_forward(const float* value, const int64_t* shapes_hw, const int64_t* level_start,
                                const float* loc, const float* attn_w, float* out, int B, int num_keys,
                                int heads, int head_dim, int levels, int points, int Nq, int im2col_step,
                                void* stream) {
    (void)im2col_step;
    int rc = check_common(value, shapes_hw, level_start, loc, attn_w, B, num_keys, heads, head_dim, levels,
                          points, Nq);
    if (rc) return rc;
    const long total = (long)B * Nq * heads;
    if (total == 0) return VER_OK;   // empty input: nothing to write (reference returns an empty tensor)
    VER_REQUIRE(out, VER_EINVAL, "ver_msda_forward: out is null");
    hipStream_t st = (hipStream_t)stream;
    return dispatch_group(head_dim, [&](auto g, auto nc) {
        constexpr int G = decltype(g)::value, NC = decltype(nc)::value;
        const int gpb = 256 / G;
        const unsigned blocks = (unsigned)((total + gpb - 1) / gpb);
        hipLaunchKernelGGL((k_msda_fwd<G, NC>), dim3(blocks), dim3(256), 0, st, value, shapes_hw, level_start,
                           loc, attn_w, out, B, num_keys, heads, head_dim, levels, points, Nq);
        return ver_check_launch("ver_msda_forward");
    });
}

extern "C" int ver_msda_backward(const float* value, const int64_t* shapes_hw, const int64_t* level_start,
                                 const float* loc, const float* attn_w, const float* grad_out,
                                 float* grad_value, float* grad_loc, float* grad_attn_w, int B, int num_keys,
                                 int heads, int head_dim, int levels, int points, int Nq, int im2col_step,
                                 void* stream) {
    (void)im2col_step;
    int rc = check_common(value, shapes_hw, level_start, loc, attn_w, B, num_keys, heads, head_dim, levels,
                          points, Nq);
    if (rc) return rc;
    const long total = (long)B * Nq * heads;
    if (total == 0) return VER_OK;
    VER_REQUIRE(grad_out && grad_value && grad_loc && grad_attn_w, VER_EINVAL,
                "ver_msda_backward: null gradient pointer");
    hipStream_t st = (hipStream_t)stream;
    return dispatch_group(head_dim, [&](auto g, auto nc) {
        constexpr int G = decltype(g)::value, NC = decltype(nc)::value;
        const int gpb = 256 / G;
        const unsigned blocks = (unsigned)((total + gpb - 1) / gpb);
        hipLaunchKernelGGL((k_msda_bwd<G, NC>), dim3(blocks), dim3(256), 0, st, value, shapes_hw, level_start,
                           loc, attn_w, grad_out, grad_value, grad_loc, grad_attn_w, B, num_keys, heads,
                           head_dim, levels, points, Nq);
        return ver_check_launch("ver_msda_backward");
    });
}

extern "C" int ver_msda3d_forward(const float* value, const int64_t* shapes_dhw, const int64_t* level_start,
                                  const float* loc, const float* attn_w, float* out, int B, int num_keys,
                                  int heads, int head_dim, int levels, int points, int Nq, void* stream) {
    int rc = check_common(value, shapes_dhw, level_start, loc, attn_w, B, num_keys, heads, head_dim, levels, points,
                          Nq);
    if (rc) return rc;
    const long total = (long)B * Nq * heads;
    if (total == 0) return VER_OK;
    VER_REQUIRE(out, VER_EINVAL, "ver_msda3d_forward: out is null");
    hipStream_t st = (hipStream_t)stream;
    return dispatch_group(head_dim, [&](auto g, auto nc) {
        constexpr int G = decltype(g)::value, NC = decltype(nc)::value;
        const int gpb = 256 / G;
        hipLaunchKernelGGL((k_msda3d_fwd<G, NC>), dim3((unsigned)((total + gpb - 1) / gpb)), dim3(256), 0, st, value,
                           shapes_dhw, level_start, loc, attn_w, out, B, num_keys, heads, head_dim, levels, points, Nq);
        return ver_check_launch("ver_msda3d_forward");
    });
}

extern "C" int ver_msda3d_backward(const float* value, const int64_t* shapes_dhw, const int64_t* level_start,
                                   const float* loc, const float* attn_w, const float* grad_out, float* grad_value,
                                   float* grad_loc, float* grad_attn_w, int B, int num_keys, int heads, int head_dim,
                                   int levels, int points, int Nq, void* stream) {
    int rc = check_common(value, shapes_dhw, level_start, loc, attn_w, B, num_keys, heads, head_dim, levels, points,
                          Nq);
    if (rc) return rc;
    const long total = (long)B * Nq * heads;
    if (total == 0) return VER_OK;
    VER_REQUIRE(grad_out && grad_value && grad_loc && grad_attn_w, VER_EINVAL,
                "ver_msda3d_backward: null gradient pointer");
    hipStream_t st = (hipStream_t)stream;
    return dispatch_group(head_dim, [&](auto g, auto nc) {
        constexpr int G = decltype(g)::value, NC = decltype(nc)::value;
        const int gpb = 256 / G;
        hipLaunchKernelGGL((k_msda3d_bwd<G, NC>), dim3((unsigned)((total + gpb - 1) / gpb)), dim3(256), 0, st, value,
                           shapes_dhw, level_start, loc, attn_w, grad_out, grad_value, grad_loc, grad_attn_w, B,
                           num_keys, heads, head_dim, levels, points, Nq);
        return ver_check_launch("ver_msda3d_backward");
    });
}
