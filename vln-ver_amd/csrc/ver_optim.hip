// Gradient clipping by the global L2 norm + AdamW over a list of fp32 tensors, two launches per step
// (include/ver_ops.h: ver_clip_adamw_*).  The reference's step is mmcv's OptimizerHook(grad_clip=dict(max_norm=...)) in front
// of torch.optim.AdamW (projects/configs/verformer/vocc.py:268-274): `clip_grad_norm_` = a per-tensor norm pass, a norm of
// norms and a multiply pass over every gradient; AdamW = another pass over parameter, gradient and both moments.  Here:
//   k_sqnorm      : per chunk of `chunk` elements, sum of squares of the gradient -> partial[chunk index]   (one read of g)
//   k_clip_adamw  : every workgroup adds the partials up IN ORDER (deterministic), forms the clip factor
//                   clamp(max_norm / (norm + 1e-6), max = 1) (NaN handed on, as torch.clamp does) and updates its chunk: g' = g * factor (not written back),
//                   p *= 1 - lr * wd;  m = b1 m + (1 - b1) g';  v = b2 v + (1 - b2) g'^2;
//                   p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps)                 -- torch.optim.AdamW, amsgrad off
// 28 bytes per parameter in all (p, g, m, v read; p, m, v written) + 4 for the norm: the step's 158 M parameters are two
// streaming passes of 0.6 + 4.4 GB instead of 1.7 ms in four torch passes.
#include "ver_common.h"

namespace {
struct Chunk {
    int tensor;
    int first;            // first chunk of that tensor
};

__device__ __forceinline__ float block_sum(float v) {
    __shared__ float red[4];
#pragma unroll
    for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}
}  // namespace

// table: [4][n] device pointers (p | g | m | v), sizes [n], chunk_tensor [n_chunks], chunk_index [n_chunks] (index of the
// chunk inside its tensor)
__global__ __launch_bounds__(256) void k_sqnorm(const float* const* __restrict__ table, const long* __restrict__ sizes,
                                                const int* __restrict__ chunk_tensor, const int* __restrict__ chunk_index,
                                                int n, int chunk, float* __restrict__ partial, int* __restrict__ steps) {
    const int t = chunk_tensor[blockIdx.x];
    // per-tensor update counters kept on the device (ver_clip_adamw_step_tensors): the first chunk of a tensor counts this
    // update; k_clip_adamw, next on the stream, reads the new count for its bias corrections
    if (steps && chunk_index[blockIdx.x] == 0 && threadIdx.x == 0) steps[t] += 1;
    const long start = (long)chunk_index[blockIdx.x] * chunk;
    const long len = min((long)chunk, sizes[t] - start);
    const float* g = table[n + t] + start;
    float s = 0.f;
    if ((reinterpret_cast<uintptr_t>(g) & 15) == 0) {
        const long nv = len >> 2;
        for (long i = threadIdx.x; i < nv; i += 256) {
            const float4 q = reinterpret_cast<const float4*>(g)[i];
            s += q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w;
        }
        for (long i = (nv << 2) + threadIdx.x; i < len; i += 256) s += g[i] * g[i];
    } else {
        for (long i = threadIdx.x; i < len; i += 256) s += g[i] * g[i];
    }
    s = block_sum(s);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// hyper (or NULL: the scalar arguments hold for every tensor): [n][6] floats per tensor = lr, beta1, beta2, eps, weight decay,
// (unused); steps (with hyper): the number of updates of tensor t INCLUDING this one (k_sqnorm has counted it)
__global__ __launch_bounds__(256) void k_clip_adamw(float* const* __restrict__ table, const long* __restrict__ sizes,
                                                    const int* __restrict__ chunk_tensor, const int* __restrict__ chunk_index,
                                                    int n, int chunk, const float* __restrict__ partial, int n_chunks,
                                                    float max_norm, float lr, float beta1, float beta2, float eps, float decay,
                                                    float step_size, float inv_sqrt_bc2, float* __restrict__ norm_out,
                                                    const float* __restrict__ hyper, const int* __restrict__ steps) {
    // the same in-order sum in every workgroup: no second launch, no atomics, a bitwise reproducible clip factor
    float s = 0.f;
    for (int i = threadIdx.x; i < n_chunks; i += 256) s += partial[i];
    s = block_sum(s);
    const float norm = sqrtf(s);
    // torch.nn.utils.clip_grad_norm_: coef = max_norm / (norm + 1e-6), clamped to <= 1 by torch.clamp, which hands a NaN
    // on -- a non-finite norm poisons EVERY gradient (and with it every parameter), it does not leave the finite ones unclipped
    float factor = 1.0f;
    if (max_norm > 0.f) {
        const float coef = max_norm / (norm + 1e-6f);
        factor = coef < 1.0f ? coef : (coef != coef ? coef : 1.0f);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && norm_out) *norm_out = norm;
    const int t = chunk_tensor[blockIdx.x];
    const long start = (long)chunk_index[blockIdx.x] * chunk;
    const long len = min((long)chunk, sizes[t] - start);
    float* p = table[t] + start;
    const float* g = table[n + t] + start;
    float* m = table[2 * n + t] + start;
    float* v = table[3 * n + t] + start;
    if (hyper) {
        const float* h = hyper + 6 * t;
        const double st = (double)steps[t];
        lr = h[0], beta1 = h[1], beta2 = h[2], eps = h[3];
        decay = (float)(1.0 - (double)lr * (double)h[4]);
        step_size = (float)((double)lr / (1.0 - pow((double)beta1, st)));
        inv_sqrt_bc2 = (float)(1.0 / sqrt(1.0 - pow((double)beta2, st)));
    }
    auto upd = [&](float& pp, float gg, float& mm, float& vv) {
        gg *= factor;
        pp *= decay;
        mm = beta1 * mm + (1.f - beta1) * gg;
        vv = beta2 * vv + (1.f - beta2) * gg * gg;
        pp -= step_size * mm / (sqrtf(vv) * inv_sqrt_bc2 + eps);
    };
    const bool vec = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                       reinterpret_cast<uintptr_t>(v)) & 15) == 0;
    long done = 0;
    if (vec) {
        const long nv = len >> 2;
        for (long i = threadIdx.x; i < nv; i += 256) {
            float4 pp = reinterpret_cast<float4*>(p)[i], mm = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
            const float4 gg = reinterpret_cast<const float4*>(g)[i];
            upd(pp.x, gg.x, mm.x, vv.x), upd(pp.y, gg.y, mm.y, vv.y), upd(pp.z, gg.z, mm.z, vv.z), upd(pp.w, gg.w, mm.w, vv.w);
            reinterpret_cast<float4*>(p)[i] = pp, reinterpret_cast<float4*>(m)[i] = mm, reinterpret_cast<float4*>(v)[i] = vv;
        }
        done = nv << 2;
    }
    for (long i = done + threadIdx.x; i < len; i += 256) {
        float pp = p[i], mm = m[i], vv = v[i];
        upd(pp, g[i], mm, vv);
        p[i] = pp, m[i] = mm, v[i] = vv;
    }
}

extern "C" int ver_clip_adamw_step(void* const* table, const long* sizes, const int* chunk_tensor, const int* chunk_index,
                                   int n_tensors, int n_chunks, int chunk_elems, float* partial, float* norm_out,
                                   float max_norm, float lr, float beta1, float beta2, float eps, float weight_decay,
                                   long step, void* stream) {
    VER_REQUIRE(n_tensors >= 0 && n_chunks >= 0 && chunk_elems > 0 && chunk_elems % 4 == 0, VER_EINVAL,
                "ver_clip_adamw_step: bad sizes (%d tensors, %d chunks of %d)", n_tensors, n_chunks, chunk_elems);
    VER_REQUIRE(step >= 1, VER_EINVAL, "ver_clip_adamw_step: step %ld (the first update is step 1)", step);
    VER_REQUIRE(lr >= 0.f && beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f && eps >= 0.f && weight_decay >= 0.f,
                VER_EINVAL, "ver_clip_adamw_step: hyper-parameters out of range");
    if (n_tensors == 0 || n_chunks == 0) return VER_OK;
    VER_REQUIRE(table && sizes && chunk_tensor && chunk_index && partial, VER_EINVAL, "ver_clip_adamw_step: null pointer argument");
    hipStream_t st = (hipStream_t)stream;
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    hipLaunchKernelGGL(k_sqnorm, dim3(n_chunks), dim3(256), 0, st, (const float* const*)table, sizes, chunk_tensor, chunk_index,
                       n_tensors, chunk_elems, partial, (int*)nullptr);
    hipLaunchKernelGGL(k_clip_adamw, dim3(n_chunks), dim3(256), 0, st, (float* const*)table, sizes, chunk_tensor, chunk_index,
                       n_tensors, chunk_elems, (const float*)partial, n_chunks, max_norm, lr, beta1, beta2, eps,
                       (float)(1.0 - (double)lr * (double)weight_decay), (float)((double)lr / bc1), (float)(1.0 / sqrt(bc2)), norm_out,
                       (const float*)nullptr, (const int*)nullptr);
    return ver_check_launch("ver_clip_adamw_step");
}

extern "C" int ver_clip_adamw_step_tensors(void* const* table, const long* sizes, const int* chunk_tensor, const int* chunk_index,
                                           const float* hyper, int* steps, int n_tensors, int n_chunks, int chunk_elems,
                                           float* partial, float* norm_out, float max_norm, void* stream) {
    VER_REQUIRE(n_tensors >= 0 && n_chunks >= 0 && chunk_elems > 0 && chunk_elems % 4 == 0, VER_EINVAL,
                "ver_clip_adamw_step_tensors: bad sizes (%d tensors, %d chunks of %d)", n_tensors, n_chunks, chunk_elems);
    if (n_tensors == 0 || n_chunks == 0) return VER_OK;
    VER_REQUIRE(table && sizes && chunk_tensor && chunk_index && partial && hyper && steps, VER_EINVAL,
                "ver_clip_adamw_step_tensors: null pointer argument");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_sqnorm, dim3(n_chunks), dim3(256), 0, st, (const float* const*)table, sizes, chunk_tensor, chunk_index,
                       n_tensors, chunk_elems, partial, steps);
    hipLaunchKernelGGL(k_clip_adamw, dim3(n_chunks), dim3(256), 0, st, (float* const*)table, sizes, chunk_tensor, chunk_index,
                       n_tensors, chunk_elems, (const float*)partial, n_chunks, max_norm, 0.f, 0.f, 0.f, 0.f, 1.f, 0.f, 1.f, norm_out,
                       hyper, (const int*)steps);
    return ver_check_launch("ver_clip_adamw_step_tensors");
}
