// Fused LayerNorm + ReLU over rows of 128 channels, forward and backward (include/ver_ops.h).
//
// The occupancy MLP of the reference head (`occ_branches` = [Linear(128,128), LayerNorm(128),
// ReLU] x2 + Linear(128,16), dense_heads/voxelformer_occupancy_head.py:241-248, applied to
// 504 000 voxels per viewpoint at :580) spends its time in the two LayerNorm+ReLU stages: HBM
// streaming over [N,128] with N = 504 000 x viewpoints.  One pass each way here instead of
// LayerNorm, ReLU (+ fp32 up/down casts under autocast) as separate kernels.
// 16 lanes (one DPP row) own a row: 8 channels per lane = one 16-byte (bf16) or two 16-byte (fp32)
// vectors; mean / variance / the two backward row sums are DPP butterflies; d(gamma), d(beta)
// are accumulated in registers over a grid-stride loop and reduced once per workgroup.
#include "ver_common.h"

namespace {
constexpr int kW = 128;

template <bool BF16>
__device__ __forceinline__ void load8(const void* base, long row, int gl, float (&v)[8]) {
    if (BF16) {
        const uint4 t = reinterpret_cast<const uint4*>(base)[row * 16 + gl];
        const uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[2 * j] = __uint_as_float(w[j] << 16);
            v[2 * j + 1] = __uint_as_float(w[j] & 0xffff0000u);
        }
    } else {
        const float4 a = reinterpret_cast<const float4*>(base)[row * 32 + gl * 2];
        const float4 b = reinterpret_cast<const float4*>(base)[row * 32 + gl * 2 + 1];
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
        v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
}

__device__ __forceinline__ uint32_t f2bf(float f) {   // round to nearest even
    const uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}

template <bool BF16>
__device__ __forceinline__ void store8(void* base, long row, int gl, const float (&v)[8]) {
    if (BF16) {
        uint4 t;
        t.x = f2bf(v[0]) | (f2bf(v[1]) << 16);
        t.y = f2bf(v[2]) | (f2bf(v[3]) << 16);
        t.z = f2bf(v[4]) | (f2bf(v[5]) << 16);
        t.w = f2bf(v[6]) | (f2bf(v[7]) << 16);
        reinterpret_cast<uint4*>(base)[row * 16 + gl] = t;
    } else {
        reinterpret_cast<float4*>(base)[row * 32 + gl * 2] = make_float4(v[0], v[1], v[2], v[3]);
        reinterpret_cast<float4*>(base)[row * 32 + gl * 2 + 1] = make_float4(v[4], v[5], v[6], v[7]);
    }
}
}  // namespace

template <bool BF16>
__global__ __launch_bounds__(256) void k_ln_relu_fwd(const void* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, void* __restrict__ y,
                                                     float* __restrict__ mean, float* __restrict__ rstd, long N,
                                                     float eps) {
    const int gl = threadIdx.x & 15;
    float g[8], bt[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        g[j] = gamma[gl * 8 + j];
        bt[j] = beta[gl * 8 + j];
    }
    for (long row = (long)blockIdx.x * 16 + (threadIdx.x >> 4); row < N; row += (long)gridDim.x * 16) {
        float v[8];
        load8<BF16>(x, row, gl, v);
        float s = 0.0f;
#pragma unroll
        for (int j = 0; j < 8; ++j) s += v[j];
        const float mu = group_sum<16>(s) * (1.0f / kW);
        float q = 0.0f;
#pragma unroll
        for (int j = 0; j < 8; ++j) q += (v[j] - mu) * (v[j] - mu);
        const float rs = rsqrtf(group_sum<16>(q) * (1.0f / kW) + eps);
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = fmaxf((v[j] - mu) * rs * g[j] + bt[j], 0.0f);
        store8<BF16>(y, row, gl, o);
        if (gl == 0) {
            mean[row] = mu;
            rstd[row] = rs;
        }
    }
}

template <bool BF16>
__global__ __launch_bounds__(256) void k_ln_relu_bwd(const void* __restrict__ x, const void* __restrict__ gy,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     void* __restrict__ gx, float* ggamma, float* gbeta, long N) {
    __shared__ float red[2][16][kW];     // [gamma|beta][row slot of the block][channel]
    const int gl = threadIdx.x & 15, slot = threadIdx.x >> 4;
    float g[8], bt[8], ag[8], ab[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        g[j] = gamma[gl * 8 + j];
        bt[j] = beta[gl * 8 + j];
        ag[j] = 0.0f;
        ab[j] = 0.0f;
    }
    for (long row = (long)blockIdx.x * 16 + slot; row < N; row += (long)gridDim.x * 16) {
        float v[8], d[8];
        load8<BF16>(x, row, gl, v);
        load8<BF16>(gy, row, gl, d);
        const float mu = mean[row], rs = rstd[row];
        float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float xh = (v[j] - mu) * rs;
            const float dz = (xh * g[j] + bt[j] > 0.0f) ? d[j] : 0.0f;     // ReLU gate
            ag[j] += dz * xh;
            ab[j] += dz;
            const float dg = dz * g[j];
            v[j] = xh;
            d[j] = dg;
            s1 += dg;
            s2 += dg * xh;
        }
        const float m1 = group_sum<16>(s1) * (1.0f / kW), m2 = group_sum<16>(s2) * (1.0f / kW);
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = rs * (d[j] - m1 - v[j] * m2);
        store8<BF16>(gx, row, gl, o);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        red[0][slot][gl * 8 + j] = ag[j];
        red[1][slot][gl * 8 + j] = ab[j];
    }
    __syncthreads();
    {
        const int which = threadIdx.x >> 7, ch = threadIdx.x & 127;      // 256 threads = 2 x 128 channels
        float t = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) t += red[which][r][ch];
        atomicAdd((which ? gbeta : ggamma) + ch, t);
    }
}

namespace {
int check_ln(const void* a, const void* b, const void* c, const void* d, long N, int W, int dtype) {
    VER_REQUIRE(N >= 0, VER_EINVAL, "ver_ln_relu: negative row count");
    VER_REQUIRE(W == kW, VER_EUNSUPPORTED, "ver_ln_relu: width %d (built for %d)", W, kW);
    VER_REQUIRE(dtype == VER_F32 || dtype == VER_BF16, VER_EINVAL, "ver_ln_relu: dtype %d", dtype);
    if (N == 0) return VER_OK;
    VER_REQUIRE(a && b && c && d, VER_EINVAL, "ver_ln_relu: null pointer argument");
    VER_REQUIRE((((uintptr_t)a | (uintptr_t)d) & 15) == 0, VER_EINVAL, "ver_ln_relu: buffers must be 16-byte aligned");
    return VER_OK;
}
unsigned ln_blocks(long N) {
    const long want = (N + 15) / 16;
    return (unsigned)(want < 256L * 16 ? (want > 0 ? want : 1) : 256L * 16);
}
}  // namespace

extern "C" int ver_ln_relu_forward(const void* x, const float* gamma, const float* beta, void* y, float* mean,
                                   float* rstd, long N, int W, float eps, int dtype, void* stream) {
    int rc = check_ln(x, gamma, beta, y, N, W, dtype);
    if (rc) return rc;
    if (N == 0) return VER_OK;
    VER_REQUIRE(mean && rstd, VER_EINVAL, "ver_ln_relu_forward: null statistics pointer");
    if (dtype == VER_BF16)
        hipLaunchKernelGGL(k_ln_relu_fwd<true>, dim3(ln_blocks(N)), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, y,
                           mean, rstd, N, eps);
    else
        hipLaunchKernelGGL(k_ln_relu_fwd<false>, dim3(ln_blocks(N)), dim3(256), 0, (hipStream_t)stream, x, gamma, beta,
                           y, mean, rstd, N, eps);
    return ver_check_launch("ver_ln_relu_forward");
}

extern "C" int ver_ln_relu_backward(const void* x, const void* grad_y, const float* gamma, const float* beta,
                                    const float* mean, const float* rstd, void* grad_x, float* grad_gamma,
                                    float* grad_beta, long N, int W, int dtype, void* stream) {
    int rc = check_ln(x, gamma, beta, grad_x, N, W, dtype);
    if (rc) return rc;
    VER_REQUIRE(grad_gamma && grad_beta, VER_EINVAL, "ver_ln_relu_backward: null parameter-gradient pointer");
    hipStream_t st = (hipStream_t)stream;
    int zrc = ver_zero_async(grad_gamma, kW * sizeof(float), st);          // (kernel zero fills: ver_zero_async)
    if (!zrc) zrc = ver_zero_async(grad_beta, kW * sizeof(float), st);
    if (zrc) return zrc;
    hipError_t e = hipSuccess;
    if (e != hipSuccess) return ver_fail(VER_ELAUNCH, "ver_ln_relu_backward: memset: %s", hipGetErrorString(e));
    if (N == 0) return VER_OK;
    VER_REQUIRE(grad_y && mean && rstd, VER_EINVAL, "ver_ln_relu_backward: null pointer argument");
    if (dtype == VER_BF16)
        hipLaunchKernelGGL(k_ln_relu_bwd<true>, dim3(ln_blocks(N)), dim3(256), 0, st, x, grad_y, gamma, beta, mean, rstd,
                           grad_x, grad_gamma, grad_beta, N);
    else
        hipLaunchKernelGGL(k_ln_relu_bwd<false>, dim3(ln_blocks(N)), dim3(256), 0, st, x, grad_y, gamma, beta, mean,
                           rstd, grad_x, grad_gamma, grad_beta, N);
    return ver_check_launch("ver_ln_relu_backward");
}
