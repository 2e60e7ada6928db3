// Fused  y = LayerNorm(residual + dropout(a))  of the encoder layers, forward and backward (include/ver_ops.h).
//
// VoxelFormerLayer (voxel_encoder.py:344-464, operation_order cross_attn - norm - ffn - norm) ends both of its branches
// the same way: SpatialCrossAttention returns `dropout(output_proj(slots)) + residual` (spatial_cross_attention.py:173-176),
// the FFN `identity + dropout(layers(x))` (mmcv FFN), and a LayerNorm(768) follows.  As torch kernels under bf16 autocast
// that is dropout (bf16 -> bf16 + mask), add (bf16 + fp32 -> fp32), LayerNorm (fp32 -> fp32) and the next Linear's cast
// (fp32 -> bf16): four passes over [N, 768] forward (3.8 GB for N = 172 800 rows) and as many backward.  Here: one pass
// each way (1.7 GB), the bf16 copy for the next Linear written alongside, and no mask tensor -- the keep decision of
// element i is a hash of (seed, i) that the backward pass recomputes.
//
// One wave per row: lane l owns elements 4 (l + 64 k) .. + 3 of the row, k < C / 256 (16-byte loads of fp32, 8-byte of
// bf16, coalesced across the wave); row statistics are wave reductions; d(gamma), d(beta) are accumulated per lane over
// the grid-stride loop and reduced once per workgroup.
#include "ver_common.h"

namespace {
constexpr int kMaxK = 4;                     // C <= 1024, a multiple of 256

__device__ __forceinline__ uint32_t f2bf(float f) {   // round to nearest even
    const uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}

// keep decision of element `idx` (murmur3 finaliser of idx ^ seed, 24 bits against the threshold)
__device__ __forceinline__ bool keep_elem(uint64_t idx, uint64_t seed, uint32_t thresh) {
    uint64_t h = idx * 0x9E3779B97F4A7C15ull + seed;
    h ^= h >> 33;
    h *= 0xff51afd7ed558ccdull;
    h ^= h >> 33;
    h *= 0xc4ceb9fe1a85ec53ull;
    h ^= h >> 33;
    return (uint32_t)(h & 0xffffffu) < thresh;
}

template <bool ABF16>
__device__ __forceinline__ void load_a4(const void* a, long off, float (&v)[4]) {
    if (ABF16) {
        const uint2 t = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(a) + off);
        v[0] = __uint_as_float(t.x << 16);
        v[1] = __uint_as_float(t.x & 0xffff0000u);
        v[2] = __uint_as_float(t.y << 16);
        v[3] = __uint_as_float(t.y & 0xffff0000u);
    } else {
        const float4 t = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(a) + off);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
}

__device__ __forceinline__ void store_bf4(uint16_t* p, long off, const float (&v)[4]) {
    uint2 t;
    t.x = f2bf(v[0]) | (f2bf(v[1]) << 16);
    t.y = f2bf(v[2]) | (f2bf(v[3]) << 16);
    *reinterpret_cast<uint2*>(p + off) = t;
}

// x = residual + dropout(a) of this lane's 4 K elements of row `row`
template <bool ABF16, int K>
__device__ __forceinline__ void load_x(const void* a, const float* res, long row, int C, int lane, bool drop,
                                       uint64_t seed, uint32_t thresh, float scale, float (&x)[K][4],
                                       bool (&keep)[K][4]) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const long off = row * C + 4 * (lane + 64 * k);
        float av[4];
        load_a4<ABF16>(a, off, av);
        const float4 r = *reinterpret_cast<const float4*>(res + off);
        const float rv[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            keep[k][i] = !drop || keep_elem((uint64_t)(off + i), seed, thresh);
            x[k][i] = rv[i] + (keep[k][i] ? av[i] * scale : 0.0f);
        }
    }
}
}  // namespace

template <bool ABF16, int K>
__global__ __launch_bounds__(256) void k_add_ln_fwd(const void* __restrict__ a, const float* __restrict__ res,
                                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                                    const int64_t* __restrict__ seed_p, float p_drop, float eps,
                                                    float* __restrict__ y, uint16_t* __restrict__ y16,
                                                    float* __restrict__ mean, float* __restrict__ rstd, long N) {
    constexpr int C = 256 * K;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool drop = p_drop > 0.0f;
    const uint64_t seed = drop ? (uint64_t)seed_p[0] : 0ull;
    const uint32_t thresh = (uint32_t)((1.0f - p_drop) * 16777216.0f);
    const float scale = drop ? 1.0f / (1.0f - p_drop) : 1.0f;
    float g[K][4], bt[K][4];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const float4 gv = *reinterpret_cast<const float4*>(gamma + 4 * (lane + 64 * k));
        const float4 bv = *reinterpret_cast<const float4*>(beta + 4 * (lane + 64 * k));
        g[k][0] = gv.x; g[k][1] = gv.y; g[k][2] = gv.z; g[k][3] = gv.w;
        bt[k][0] = bv.x; bt[k][1] = bv.y; bt[k][2] = bv.z; bt[k][3] = bv.w;
    }
    for (long row = (long)blockIdx.x * 4 + wave; row < N; row += (long)gridDim.x * 4) {
        float x[K][4];
        bool keep[K][4];
        load_x<ABF16, K>(a, res, row, C, lane, drop, seed, thresh, scale, x, keep);
        float s = 0.0f;
#pragma unroll
        for (int k = 0; k < K; ++k) s += (x[k][0] + x[k][1]) + (x[k][2] + x[k][3]);
        const float mu = group_sum<64>(s) * (1.0f / C);
        float q = 0.0f;
#pragma unroll
        for (int k = 0; k < K; ++k)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                x[k][i] -= mu;
                q += x[k][i] * x[k][i];
            }
        const float rs = rsqrtf(group_sum<64>(q) * (1.0f / C) + eps);
        if (lane == 0) {
            mean[row] = mu;
            rstd[row] = rs;
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const long off = row * C + 4 * (lane + 64 * k);
            float o[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = x[k][i] * rs * g[k][i] + bt[k][i];
            *reinterpret_cast<float4*>(y + off) = make_float4(o[0], o[1], o[2], o[3]);
            if (y16) store_bf4(y16, off, o);
        }
    }
}

template <bool ABF16, int K>
__global__ __launch_bounds__(256) void k_add_ln_bwd(const float* __restrict__ dy, const uint16_t* __restrict__ dy16,
                                                    const void* __restrict__ a, const float* __restrict__ res,
                                                    const float* __restrict__ gamma, const float* __restrict__ mean,
                                                    const float* __restrict__ rstd, const int64_t* __restrict__ seed_p,
                                                    float p_drop, void* __restrict__ d_a, float* __restrict__ d_res,
                                                    float* __restrict__ dgamma, float* __restrict__ dbeta, long N) {
    constexpr int C = 256 * K;
    __shared__ float red[2][4][C];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool drop = p_drop > 0.0f;
    const uint64_t seed = drop ? (uint64_t)seed_p[0] : 0ull;
    const uint32_t thresh = (uint32_t)((1.0f - p_drop) * 16777216.0f);
    const float scale = drop ? 1.0f / (1.0f - p_drop) : 1.0f;
    float g[K][4], dg[K][4], db[K][4];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const float4 gv = *reinterpret_cast<const float4*>(gamma + 4 * (lane + 64 * k));
        g[k][0] = gv.x; g[k][1] = gv.y; g[k][2] = gv.z; g[k][3] = gv.w;
#pragma unroll
        for (int i = 0; i < 4; ++i) dg[k][i] = db[k][i] = 0.0f;
    }
    for (long row = (long)blockIdx.x * 4 + wave; row < N; row += (long)gridDim.x * 4) {
        float x[K][4], go[K][4];
        bool keep[K][4];
        load_x<ABF16, K>(a, res, row, C, lane, drop, seed, thresh, scale, x, keep);
        const float mu = mean[row], rs = rstd[row];
        float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const long off = row * C + 4 * (lane + 64 * k);
            const float4 t = *reinterpret_cast<const float4*>(dy + off);
            go[k][0] = t.x; go[k][1] = t.y; go[k][2] = t.z; go[k][3] = t.w;
            if (dy16) {
                float h[4];
                load_a4<true>(dy16, off, h);
#pragma unroll
                for (int i = 0; i < 4; ++i) go[k][i] += h[i];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                x[k][i] = (x[k][i] - mu) * rs;             // normalised value
                dg[k][i] += go[k][i] * x[k][i];
                db[k][i] += go[k][i];
                go[k][i] *= g[k][i];
                s1 += go[k][i];
                s2 += go[k][i] * x[k][i];
            }
        }
        const float m1 = group_sum<64>(s1) * (1.0f / C), m2 = group_sum<64>(s2) * (1.0f / C);
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const long off = row * C + 4 * (lane + 64 * k);
            float dx[4], da[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                dx[i] = (go[k][i] - m1 - x[k][i] * m2) * rs;
                da[i] = keep[k][i] ? dx[i] * scale : 0.0f;
            }
            *reinterpret_cast<float4*>(d_res + off) = make_float4(dx[0], dx[1], dx[2], dx[3]);
            if (ABF16)
                store_bf4(reinterpret_cast<uint16_t*>(d_a), off, da);
            else
                *reinterpret_cast<float4*>(reinterpret_cast<float*>(d_a) + off) = make_float4(da[0], da[1], da[2], da[3]);
        }
    }
    // d(gamma), d(beta): the four waves of the workgroup, then one atomic per element and workgroup
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            red[0][wave][4 * (lane + 64 * k) + i] = dg[k][i];
            red[1][wave][4 * (lane + 64 * k) + i] = db[k][i];
        }
    __syncthreads();
    for (int e = threadIdx.x; e < C; e += 256) {
        atomicAdd(dgamma + e, (red[0][0][e] + red[0][1][e]) + (red[0][2][e] + red[0][3][e]));
        atomicAdd(dbeta + e, (red[1][0][e] + red[1][1][e]) + (red[1][2][e] + red[1][3][e]));
    }
}

namespace {
int check_add_ln(const char* who, const void* a, const float* res, const float* gamma, long N, int C, int a_dtype,
                 float p_drop, const int64_t* seed) {
    VER_REQUIRE(N >= 0, VER_EINVAL, "%s: negative row count", who);
    VER_REQUIRE(C > 0 && C % 256 == 0 && C <= 256 * kMaxK, VER_EUNSUPPORTED,
                "%s: built for rows of 256, 512, 768 or 1024 channels (got %d)", who, C);
    VER_REQUIRE(a_dtype == VER_F32 || a_dtype == VER_BF16, VER_EINVAL, "%s: dtype %d is neither VER_F32 nor VER_BF16", who,
                a_dtype);
    VER_REQUIRE(p_drop >= 0.0f && p_drop < 1.0f, VER_EINVAL, "%s: dropout probability %g outside [0, 1)", who, p_drop);
    if (N == 0) return VER_OK;
    VER_REQUIRE(a && res && gamma && (seed || p_drop == 0.0f), VER_EINVAL, "%s: null pointer argument", who);
    VER_REQUIRE(((uintptr_t)a & 15) == 0 && ((uintptr_t)res & 15) == 0, VER_EINVAL, "%s: buffers must be 16-byte aligned", who);
    return VER_OK;
}
}  // namespace

extern "C" int ver_add_ln_forward(const void* a, int a_dtype, const float* residual, const float* gamma,
                                  const float* beta, const int64_t* seed, float p_drop, float eps, float* y,
                                  void* y_bf16, float* mean, float* rstd, long N, int C, void* stream) {
    int rc = check_add_ln("ver_add_ln_forward", a, residual, gamma, N, C, a_dtype, p_drop, seed);
    if (rc) return rc;
    if (N == 0) return VER_OK;
    VER_REQUIRE(beta && y && mean && rstd, VER_EINVAL, "ver_add_ln_forward: null pointer argument");
    const unsigned grid = (unsigned)(((N + 3) / 4) < 8192 ? ((N + 3) / 4) : 8192);
    hipStream_t st = (hipStream_t)stream;
#define VER_FWD(BF, K_)                                                                                          \
    hipLaunchKernelGGL((k_add_ln_fwd<BF, K_>), dim3(grid), dim3(256), 0, st, a, residual, gamma, beta, seed, p_drop, \
                       eps, y, (uint16_t*)y_bf16, mean, rstd, N)
#define VER_FWD_K(BF)                        \
    do {                                     \
        if (C == 256) VER_FWD(BF, 1);        \
        else if (C == 512) VER_FWD(BF, 2);   \
        else if (C == 768) VER_FWD(BF, 3);   \
        else VER_FWD(BF, 4);                 \
    } while (0)
    if (a_dtype == VER_BF16) VER_FWD_K(true); else VER_FWD_K(false);
#undef VER_FWD_K
#undef VER_FWD
    return ver_check_launch("ver_add_ln_forward");
}

extern "C" int ver_add_ln_backward(const float* grad_y, const void* grad_y_bf16, const void* a, int a_dtype,
                                   const float* residual, const float* gamma, const float* mean, const float* rstd,
                                   const int64_t* seed, float p_drop, void* grad_a, float* grad_residual,
                                   float* grad_gamma, float* grad_beta, long N, int C, void* stream) {
    int rc = check_add_ln("ver_add_ln_backward", a, residual, gamma, N, C, a_dtype, p_drop, seed);
    if (rc) return rc;
    VER_REQUIRE(grad_gamma && grad_beta, VER_EINVAL, "ver_add_ln_backward: null parameter-gradient pointer");
    hipStream_t st = (hipStream_t)stream;
    int zrc = ver_zero_async(grad_gamma, C * sizeof(float), st);           // (kernel zero fills: ver_zero_async)
    if (!zrc) zrc = ver_zero_async(grad_beta, C * sizeof(float), st);
    if (zrc) return zrc;
    hipError_t e = hipSuccess;
    if (e != hipSuccess) return ver_fail(VER_ELAUNCH, "ver_add_ln_backward: memset: %s", hipGetErrorString(e));
    if (N == 0) return VER_OK;
    VER_REQUIRE(grad_y && mean && rstd && grad_a && grad_residual, VER_EINVAL, "ver_add_ln_backward: null pointer argument");
    const unsigned grid = (unsigned)(((N + 3) / 4) < 2048 ? ((N + 3) / 4) : 2048);
#define VER_BWD(BF, K_)                                                                                           \
    hipLaunchKernelGGL((k_add_ln_bwd<BF, K_>), dim3(grid), dim3(256), 0, st, grad_y, (const uint16_t*)grad_y_bf16, a, \
                       residual, gamma, mean, rstd, seed, p_drop, grad_a, grad_residual, grad_gamma, grad_beta, N)
#define VER_BWD_K(BF)                        \
    do {                                     \
        if (C == 256) VER_BWD(BF, 1);        \
        else if (C == 512) VER_BWD(BF, 2);   \
        else if (C == 768) VER_BWD(BF, 3);   \
        else VER_BWD(BF, 4);                 \
    } while (0)
    if (a_dtype == VER_BF16) VER_BWD_K(true); else VER_BWD_K(false);
#undef VER_BWD_K
#undef VER_BWD
    return ver_check_launch("ver_add_ln_backward");
}

// ------------------------------------------------------------------------------------------
// y = dropout(relu(x)) of the FFN's hidden activation (mmcv FFN: Linear - ReLU - Dropout - Linear), one pass each way.
// The backward needs no mask and no x: y > 0 exactly where x > 0 and the element was kept, so
// d(x) = y > 0 ? d(y) / (1 - p) : 0.  Same hash as above for the keep decision.
template <bool BF16>
__global__ __launch_bounds__(256) void k_relu_dropout_fwd(const void* __restrict__ x, void* __restrict__ y,
                                                          const int64_t* __restrict__ seed_p, float p_drop, long n4) {
    const bool drop = p_drop > 0.0f;
    const uint64_t seed = drop ? (uint64_t)seed_p[0] : 0ull;
    const uint32_t thresh = (uint32_t)((1.0f - p_drop) * 16777216.0f);
    const float scale = drop ? 1.0f / (1.0f - p_drop) : 1.0f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        float v[4];
        load_a4<BF16>(x, 4 * i, v);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            v[j] = (v[j] > 0.0f && (!drop || keep_elem((uint64_t)(4 * i + j), seed, thresh))) ? v[j] * scale : 0.0f;
        if (BF16)
            store_bf4(reinterpret_cast<uint16_t*>(y), 4 * i, v);
        else
            *reinterpret_cast<float4*>(reinterpret_cast<float*>(y) + 4 * i) = make_float4(v[0], v[1], v[2], v[3]);
    }
}

template <bool BF16>
__global__ __launch_bounds__(256) void k_relu_dropout_bwd(const void* __restrict__ y, const void* __restrict__ dy,
                                                          void* __restrict__ dx, float scale, long n4) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        float v[4], g[4];
        load_a4<BF16>(y, 4 * i, v);
        load_a4<BF16>(dy, 4 * i, g);
#pragma unroll
        for (int j = 0; j < 4; ++j) g[j] = v[j] > 0.0f ? g[j] * scale : 0.0f;
        if (BF16)
            store_bf4(reinterpret_cast<uint16_t*>(dx), 4 * i, g);
        else
            *reinterpret_cast<float4*>(reinterpret_cast<float*>(dx) + 4 * i) = make_float4(g[0], g[1], g[2], g[3]);
    }
}

extern "C" int ver_relu_dropout_forward(const void* x, void* y, const int64_t* seed, float p_drop, long n, int dtype,
                                        void* stream) {
    VER_REQUIRE(n >= 0 && n % 4 == 0, VER_EINVAL, "ver_relu_dropout_forward: element count must be a multiple of 4");
    VER_REQUIRE(dtype == VER_F32 || dtype == VER_BF16, VER_EINVAL, "ver_relu_dropout_forward: dtype %d", dtype);
    VER_REQUIRE(p_drop >= 0.0f && p_drop < 1.0f, VER_EINVAL, "ver_relu_dropout_forward: dropout probability %g", p_drop);
    if (n == 0) return VER_OK;
    VER_REQUIRE(x && y && (seed || p_drop == 0.0f), VER_EINVAL, "ver_relu_dropout_forward: null pointer argument");
    const long n4 = n / 4;
    const unsigned grid = (unsigned)(((n4 + 255) / 256) < 16384 ? ((n4 + 255) / 256) : 16384);
    if (dtype == VER_BF16)
        hipLaunchKernelGGL(k_relu_dropout_fwd<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, y, seed, p_drop, n4);
    else
        hipLaunchKernelGGL(k_relu_dropout_fwd<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, y, seed, p_drop, n4);
    return ver_check_launch("ver_relu_dropout_forward");
}

extern "C" int ver_relu_dropout_backward(const void* y, const void* grad_y, void* grad_x, float p_drop, long n, int dtype,
                                         void* stream) {
    VER_REQUIRE(n >= 0 && n % 4 == 0, VER_EINVAL, "ver_relu_dropout_backward: element count must be a multiple of 4");
    VER_REQUIRE(dtype == VER_F32 || dtype == VER_BF16, VER_EINVAL, "ver_relu_dropout_backward: dtype %d", dtype);
    VER_REQUIRE(p_drop >= 0.0f && p_drop < 1.0f, VER_EINVAL, "ver_relu_dropout_backward: dropout probability %g", p_drop);
    if (n == 0) return VER_OK;
    VER_REQUIRE(y && grad_y && grad_x, VER_EINVAL, "ver_relu_dropout_backward: null pointer argument");
    const long n4 = n / 4;
    const unsigned grid = (unsigned)(((n4 + 255) / 256) < 16384 ? ((n4 + 255) / 256) : 16384);
    const float scale = 1.0f / (1.0f - p_drop);
    if (dtype == VER_BF16)
        hipLaunchKernelGGL(k_relu_dropout_bwd<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, y, grad_y, grad_x, scale, n4);
    else
        hipLaunchKernelGGL(k_relu_dropout_bwd<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, y, grad_y, grad_x, scale, n4);
    return ver_check_launch("ver_relu_dropout_backward");
}
