// Sigmoid focal loss over [N, C] logits with integer targets, forward (sum) and backward
// (include/ver_ops.h: ver_focal_loss_forward / ver_focal_loss_backward).
//
// The occupancy term of the reference's loss (dense_heads/voxelformer_occupancy_head.py:977-989,
// `loss_occupancy = FocalLoss(use_sigmoid, gamma=2, alpha=0.25)` of vocc.py:190-195 -> mmdet's
// py_sigmoid_focal_loss) runs over 504 000 x 16 logits per viewpoint.  As separate elementwise ops
// that is ~15 HBM passes over an fp32 [N,16] tensor each way; here one read-only pass for the sum
// and one read+write pass for the gradient, straight from the bf16 (or fp32) logits.
//
//   t = one_hot(target)[:, :C] (target == C -> background, all zero)      p = sigmoid(x)
//   loss = (alpha t + (1-alpha)(1-t)) * (t ? 1-p : p)^gamma * bce_with_logits(x, t)
//   t=1: d/dx = alpha (1-p)^g [ g p log p - (1-p) ]       t=0: d/dx = (1-alpha) p^g [ p - g (1-p) log(1-p) ]
//
// One thread owns 8 consecutive classes of a row (C % 8 == 0): one 16-byte (bf16) or two 16-byte
// (fp32) loads.  A target outside [0, C] makes the forward sum NaN (the reference raises in F.one_hot).
// The sum is reduced per workgroup into `partial[blockIdx]` (the caller adds the
// <= 4096 partials: deterministic, no float atomics).
#include "ver_common.h"

namespace {
struct Term {
    float loss, grad;
};

// FAST (bf16 logits, 1e-2 tolerance): log(1+e) through the hardware log instead of the ~60-instruction
// log1pf -- its 1e-7 absolute error only shows where exp(-|x|) < 1e-5, i.e. |x| > 11.5.
template <bool G2, bool FAST>
__device__ __forceinline__ Term focal_term(float x, bool pos, float gamma, float alpha) {
    if constexpr (FAST && G2) {
        // the bf16 / gamma = 2 form of the step, ~28 instructions per element: with s = 1 + exp(-|x|), p and q = 1 - p are
        // 1/s and e/s in the order the sign of x says (no subtraction), softplus(-x) = softplus(x) - x, and both branches are
        // c u^2 sp and +-c u^2 (u + 2 w sp) of (u, w, sp, c) = (q, p, softplus(-x), alpha) | (p, q, softplus(x), 1 - alpha)
        const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * fabsf(x));
        const float s1 = 1.0f + e;
        const float inv = __builtin_amdgcn_rcpf(s1), einv = e * inv;
        const bool nonneg = x >= 0.0f;
        const float pp = nonneg ? inv : einv, qq = nonneg ? einv : inv;
        const float sp_pos = fmaxf(x, 0.0f) + 0.6931471805599453f * __builtin_amdgcn_logf(s1);
        const float u = pos ? qq : pp, w = pos ? pp : qq;
        const float sp = pos ? sp_pos - x : sp_pos;
        const float cu2 = (pos ? alpha : 1.0f - alpha) * (u * u);
        Term t;
        t.loss = cu2 * sp;
        const float tt = __builtin_fmaf(w + w, sp, u);
        t.grad = (pos ? -cu2 : cu2) * tt;
        return t;
    }
    // log p = -softplus(-x), log(1-p) = -softplus(x); softplus(z) = max(z,0) + log1p(exp(-|z|))
    // FAST: the bare v_exp_f32 / v_log_f32 (the library forms wrap them in denormal-range scaling: a compare, a select and
    // an ldexp each): exp(-|x|) below 2^-126 flushes to zero, where it is far under the bf16 tolerance; 1 + e is in [1, 2]
    const float e = FAST ? __builtin_amdgcn_exp2f(-1.4426950408889634f * fabsf(x)) : __expf(-fabsf(x));
    const float l1p = FAST ? 0.6931471805599453f * __builtin_amdgcn_logf(1.0f + e) : log1pf(e);
    const float sp_pos = fmaxf(x, 0.0f) + l1p;    // softplus(x)  = -log(1-p)
    const float sp_neg = fmaxf(-x, 0.0f) + l1p;   // softplus(-x) = -log(p)
    // (FAST: the hardware reciprocal, 1 ulp, instead of the ~10-instruction IEEE division sequence)
    const float inv = FAST ? __builtin_amdgcn_rcpf(1.0f + e) : 1.0f / (1.0f + e);
    const float p = x >= 0.0f ? inv : e * inv;
    const float q = 1.0f - p;
    Term t;
    if (pos) {
        const float m = G2 ? q * q : powf(q, gamma);
        t.loss = alpha * m * sp_neg;
        t.grad = alpha * m * (-gamma * p * sp_neg - q);
    } else {
        const float m = G2 ? p * p : powf(p, gamma);
        t.loss = (1.0f - alpha) * m * sp_pos;
        t.grad = (1.0f - alpha) * m * (p + gamma * q * sp_pos);
    }
    return t;
}

template <bool BF16>
__device__ __forceinline__ void load_x8(const void* base, long v, float (&x)[8]) {
    if (BF16) {
        const uint4 t = reinterpret_cast<const uint4*>(base)[v];
        const uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            x[2 * j] = __uint_as_float(w[j] << 16);
            x[2 * j + 1] = __uint_as_float(w[j] & 0xffff0000u);
        }
    } else {
        const float4 a = reinterpret_cast<const float4*>(base)[2 * v];
        const float4 b = reinterpret_cast<const float4*>(base)[2 * v + 1];
        x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w;
        x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
    }
}

__device__ __forceinline__ uint32_t to_bf16(float f) {   // round to nearest even
    const uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
// two floats -> one packed bf16 pair, round to nearest even: gfx950's v_cvt_pk_bf16_f32 (one instruction for what to_bf16
// spells out in six per element)
__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}
}  // namespace

// WG (ver_focal_loss_forward_grad): the same pass also writes the UNSCALED gradient d loss[n,c] / d logits[n,c] (in the
// logits' dtype) to `grad`, which may be the logits buffer itself (a thread reads its eight logits before it writes their
// gradients): a training step that needs the loss value and the gradient, not the logits, then has no separate backward
// pass over the [N, C] tensor -- the consumer applies the scalar d(total) / d(loss sum) when it reads the gradient.
template <bool BF16, bool G2, bool WG = false, typename LT = int64_t>
__global__ __launch_bounds__(256) void k_focal_fwd(const void* logits, const LT* __restrict__ target,
                                                   float* __restrict__ partial, long nvec, int vec_per_row,
                                                   float gamma, float alpha, int* __restrict__ bad_labels,
                                                   void* grad = nullptr, int row_shift = -1) {
    __shared__ float red[4];
    float acc = 0.0f;
    bool bad = false;
    // row of a 16-byte vector: a shift when the row holds a power of two of them (16 classes: two) -- the general form is a
    // 64-bit division per vector, as many instructions as the eight focal terms it addresses
    auto row_of = [&](long v) -> long { return row_shift >= 0 ? (v >> row_shift) : v / vec_per_row; };
    auto one = [&](long v, long row, int64_t t64, const float (&x)[8]) {
        const int c0 = (int)(v - row * vec_per_row) * 8;
        const int tgt = (int)t64;
        float g[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const Term t = focal_term<G2, BF16>(x[j], tgt == c0 + j, gamma, alpha);
            acc += t.loss;
            g[j] = t.grad;
        }
        if constexpr (WG) {
            if (BF16) {
                uint4 t;
                t.x = pack_bf16(g[0], g[1]);
                t.y = pack_bf16(g[2], g[3]);
                t.z = pack_bf16(g[4], g[5]);
                t.w = pack_bf16(g[6], g[7]);
                reinterpret_cast<uint4*>(grad)[v] = t;
            } else {
                reinterpret_cast<float4*>(grad)[2 * v] = make_float4(g[0], g[1], g[2], g[3]);
                reinterpret_cast<float4*>(grad)[2 * v + 1] = make_float4(g[4], g[5], g[6], g[7]);
            }
        }
        // a label outside [0, C] (F.one_hot raises on it) poisons the sum: the loss comes out NaN instead of silently
        // counting the row as background -- checked here, in the pass that reads the labels anyway, so the host needs no
        // device->host synchronisation per step to be loud about it
        if ((uint64_t)t64 > (uint64_t)(vec_per_row * 8)) {
            acc = __builtin_nanf("");
            bad = true;
        }
    };
    // four vectors per thread in flight: one request per thread and trip left the pass latency bound (0.36 of the HBM peak)
    constexpr int U = 4;
    const long stride = (long)gridDim.x * 256;
    long v = (long)blockIdx.x * 256 + threadIdx.x;
    for (; v + (U - 1) * stride < nvec; v += U * stride) {
        long row[U];
        int64_t t64[U];
        float x[U][8];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            row[u] = row_of(v + u * stride);
            t64[u] = (int64_t)target[row[u]];
            load_x8<BF16>(logits, v + u * stride, x[u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) one(v + u * stride, row[u], t64[u], x[u]);
    }
    for (; v < nvec; v += stride) {
        const long row = row_of(v);
        float x[8];
        load_x8<BF16>(logits, v, x);
        one(v, row, (int64_t)target[row], x);
    }
    // ... and raises a sticky device-side flag: callers that clean NaNs out of their losses (the head's nan_to_num,
    // as in the reference) still learn about it, from an asynchronous copy of one int
    if (bad_labels && __any(bad) && (threadIdx.x & 63) == 0) atomicOr(bad_labels, 1);
    acc = group_sum<64>(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

template <bool BF16, bool G2>
__global__ __launch_bounds__(256) void k_focal_bwd(const void* __restrict__ logits, const int64_t* __restrict__ target,
                                                   const float* __restrict__ scale, void* __restrict__ grad, long nvec,
                                                   int vec_per_row, float gamma, float alpha, int row_shift) {
    const float s = scale[0];
    for (long v = (long)blockIdx.x * 256 + threadIdx.x; v < nvec; v += (long)gridDim.x * 256) {
        const long row = row_shift >= 0 ? (v >> row_shift) : v / vec_per_row;
        const int c0 = (int)(v - row * vec_per_row) * 8;
        const int tgt = (int)target[row];
        float x[8], g[8];
        load_x8<BF16>(logits, v, x);
#pragma unroll
        for (int j = 0; j < 8; ++j) g[j] = s * focal_term<G2, BF16>(x[j], tgt == c0 + j, gamma, alpha).grad;
        if (BF16) {
            uint4 t;
            t.x = pack_bf16(g[0], g[1]);
            t.y = pack_bf16(g[2], g[3]);
            t.z = pack_bf16(g[4], g[5]);
            t.w = pack_bf16(g[6], g[7]);
            reinterpret_cast<uint4*>(grad)[v] = t;
        } else {
            reinterpret_cast<float4*>(grad)[2 * v] = make_float4(g[0], g[1], g[2], g[3]);
            reinterpret_cast<float4*>(grad)[2 * v + 1] = make_float4(g[4], g[5], g[6], g[7]);
        }
    }
}

namespace {
int row_shift_of(int vec_per_row) {      // log2 when the row holds a power of two of 16-byte vectors, else -1
    int sh = 0;
    while ((1 << sh) < vec_per_row) ++sh;
    return (1 << sh) == vec_per_row ? sh : -1;
}
int check_focal(const char* who, const void* logits, const int64_t* target, long N, int C, int dtype) {
    VER_REQUIRE(N >= 0 && C > 0, VER_EINVAL, "%s: bad shape N=%ld C=%d", who, N, C);
    VER_REQUIRE(C % 8 == 0, VER_EUNSUPPORTED, "%s: class count %d is not a multiple of 8", who, C);
    VER_REQUIRE(dtype == VER_F32 || dtype == VER_BF16, VER_EINVAL, "%s: dtype %d", who, dtype);
    if (N == 0) return VER_OK;
    VER_REQUIRE(logits && target, VER_EINVAL, "%s: null pointer argument", who);
    VER_REQUIRE(((uintptr_t)logits & 15) == 0, VER_EINVAL, "%s: logits must be 16-byte aligned", who);
    return VER_OK;
}
}  // namespace

extern "C" int ver_focal_loss_blocks(long N, int C) {
    const long nvec = N * (long)(C / 8);
    const long want = (nvec + 255) / 256;
    return (int)(want < 1 ? 1 : (want > 4096 ? 4096 : want));
}

extern "C" int ver_focal_loss_forward(const void* logits, const int64_t* target, float* partial, long N, int C,
                                      float gamma, float alpha, int dtype, int32_t* bad_labels, void* stream) {
    int rc = check_focal("ver_focal_loss_forward", logits, target, N, C, dtype);
    if (rc) return rc;
    VER_REQUIRE(partial, VER_EINVAL, "ver_focal_loss_forward: null partial-sum buffer");
    const int blocks = ver_focal_loss_blocks(N, C);
    const long nvec = N * (long)(C / 8);
    hipStream_t st = (hipStream_t)stream;
    const bool g2 = gamma == 2.0f;
#define VER_FOCAL_FWD(BF, G2)                                                                                   \
    hipLaunchKernelGGL((k_focal_fwd<BF, G2>), dim3(blocks), dim3(256), 0, st, logits, target, partial, nvec, C / 8, \
                       gamma, alpha, bad_labels, (void*)nullptr, row_shift_of(C / 8))
    if (dtype == VER_BF16) {
        if (g2) VER_FOCAL_FWD(true, true); else VER_FOCAL_FWD(true, false);
    } else {
        if (g2) VER_FOCAL_FWD(false, true); else VER_FOCAL_FWD(false, false);
    }
#undef VER_FOCAL_FWD
    return ver_check_launch("ver_focal_loss_forward");
}

extern "C" int ver_focal_loss_forward_grad(const void* logits, const int64_t* target, float* partial, void* grad, long N,
                                           int C, float gamma, float alpha, int dtype, int32_t* bad_labels, void* stream) {
    int rc = check_focal("ver_focal_loss_forward_grad", logits, target, N, C, dtype);
    if (rc) return rc;
    VER_REQUIRE(partial, VER_EINVAL, "ver_focal_loss_forward_grad: null partial-sum buffer");
    if (N == 0) return VER_OK;
    VER_REQUIRE(grad && ((uintptr_t)grad & 15) == 0, VER_EINVAL, "ver_focal_loss_forward_grad: grad must be a 16-byte aligned buffer");
    const int blocks = ver_focal_loss_blocks(N, C);
    const long nvec = N * (long)(C / 8);
    hipStream_t st = (hipStream_t)stream;
    const bool g2 = gamma == 2.0f;
#define VER_FOCAL_FWG(BF, G2)                                                                                         \
    hipLaunchKernelGGL((k_focal_fwd<BF, G2, true>), dim3(blocks), dim3(256), 0, st, logits, target, partial, nvec, C / 8, \
                       gamma, alpha, bad_labels, grad, row_shift_of(C / 8))
    if (dtype == VER_BF16) {
        if (g2) VER_FOCAL_FWG(true, true); else VER_FOCAL_FWG(true, false);
    } else {
        if (g2) VER_FOCAL_FWG(false, true); else VER_FOCAL_FWG(false, false);
    }
#undef VER_FOCAL_FWG
    return ver_check_launch("ver_focal_loss_forward_grad");
}

// the same with labels as BYTES (C <= 254): a caller that has permuted / counted its labels as bytes hands them over as they are
// (8 bytes per label less to read here, no widening copy in front of the call); 255 -- a wrapped -1 -- is an invalid label
extern "C" int ver_focal_loss_forward_grad_u8(const void* logits, const uint8_t* target, float* partial, void* grad, long N,
                                              int C, float gamma, float alpha, int dtype, int32_t* bad_labels, void* stream) {
    int rc = check_focal("ver_focal_loss_forward_grad_u8", logits, reinterpret_cast<const int64_t*>(target), N, C, dtype);
    if (rc) return rc;
    VER_REQUIRE(C <= 254, VER_EINVAL, "ver_focal_loss_forward_grad_u8: %d classes do not fit byte labels", C);
    VER_REQUIRE(partial, VER_EINVAL, "ver_focal_loss_forward_grad_u8: null partial-sum buffer");
    if (N == 0) return VER_OK;
    VER_REQUIRE(grad && ((uintptr_t)grad & 15) == 0, VER_EINVAL, "ver_focal_loss_forward_grad_u8: grad must be a 16-byte aligned buffer");
    const int blocks = ver_focal_loss_blocks(N, C);
    const long nvec = N * (long)(C / 8);
    hipStream_t st = (hipStream_t)stream;
    const bool g2 = gamma == 2.0f;
#define VER_FOCAL_FWG8(BF, G2)                                                                                              \
    hipLaunchKernelGGL((k_focal_fwd<BF, G2, true, uint8_t>), dim3(blocks), dim3(256), 0, st, logits, target, partial, nvec, \
                       C / 8, gamma, alpha, bad_labels, grad, row_shift_of(C / 8))
    if (dtype == VER_BF16) {
        if (g2) VER_FOCAL_FWG8(true, true); else VER_FOCAL_FWG8(true, false);
    } else {
        if (g2) VER_FOCAL_FWG8(false, true); else VER_FOCAL_FWG8(false, false);
    }
#undef VER_FOCAL_FWG8
    return ver_check_launch("ver_focal_loss_forward_grad_u8");
}

extern "C" int ver_focal_loss_backward(const void* logits, const int64_t* target, const float* scale, void* grad,
                                       long N, int C, float gamma, float alpha, int dtype, void* stream) {
    int rc = check_focal("ver_focal_loss_backward", logits, target, N, C, dtype);
    if (rc) return rc;
    if (N == 0) return VER_OK;
    VER_REQUIRE(scale && grad, VER_EINVAL, "ver_focal_loss_backward: null pointer argument");
    VER_REQUIRE(((uintptr_t)grad & 15) == 0, VER_EINVAL, "ver_focal_loss_backward: grad must be 16-byte aligned");
    const int blocks = ver_focal_loss_blocks(N, C);
    const long nvec = N * (long)(C / 8);
    hipStream_t st = (hipStream_t)stream;
    const bool g2 = gamma == 2.0f;
#define VER_FOCAL_BWD(BF, G2)                                                                                       \
    hipLaunchKernelGGL((k_focal_bwd<BF, G2>), dim3(blocks), dim3(256), 0, st, logits, target, scale, grad, nvec, C / 8, \
                       gamma, alpha, row_shift_of(C / 8))
    if (dtype == VER_BF16) {
        if (g2) VER_FOCAL_BWD(true, true); else VER_FOCAL_BWD(true, false);
    } else {
        if (g2) VER_FOCAL_BWD(false, true); else VER_FOCAL_BWD(false, false);
    }
#undef VER_FOCAL_BWD
    return ver_check_launch("ver_focal_loss_backward");
}
