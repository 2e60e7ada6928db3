// Fused multi-view gather for gfx950: projection + hit table + SCA forward/backward
// (C ABI in include/ver_ops.h; reference code being replaced cited there).
//
// Data layout in HBM (per launch, B viewpoints):
//   value  [B, Ncam, Nk, heads, HD]   one (camera, head) slice = Nk rows of HD contiguous
//                                      elements (384 B at HD=96 fp32) inside 768-wide token rows
//   uv     [B, Ncam, Nq, D, 2]  vis [B, Nq]  lists [B, Ncam, Nq] + counts [B, Ncam]
//   offsets/logits/slots are voxel-major: row (b, n) holds all heads.
//
// Forward kernel: one workgroup per (viewpoint, camera, head [, chunk of the camera's
// owned-voxel list]).  The 14x14xHD value tile of that (camera, head) is staged once into
// LDS with 16-byte coalesced loads (75 KB at HD=96 fp32 -> two workgroups per CU), then
// aligned groups of G=16 lanes each take one voxel: every lane carries HD/G channels, walks
// the 8 sampling points x 4 bilinear corners with ds_read_b64 and accumulates in registers.
// A voxel's output row is produced by exactly one workgroup (the lowest camera that sees it),
// which adds the other cameras' contributions straight from L2 -- so there are no atomics, no
// zero-fill pass, no padded rows, and the result is run-to-run deterministic.
//
// Backward kernel: same tiling with a second LDS tile that accumulates d(value) through
// ds_add_f32 and is flushed once; d(offsets), d(logits) are reduced over the G lanes with
// wave shuffles, softmax backward fused.
#include <type_traits>
#include "ver_common.h"

// ------------------------------------------------------------------------------------------
// projection + visibility (voxel_encoder.py:54-83,119-195).  No FMA contraction and the
// reference's operation order, so that the strict visibility inequalities see the same
// fp32 values as the reference's torch code.
__global__ __launch_bounds__(256) void k_project(const float* __restrict__ w2p,
                                                 const float* __restrict__ origin, float xmin, float ymin,
                                                 float zmin, float xr, float yr, float zr, int Ncam, int Z,
                                                 int H, int W, float img_w, float img_h,
                                                 float* __restrict__ uv, uint8_t* __restrict__ vis) {
    const int Nq = Z * H * W;
    const int n = blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (n >= Nq) return;
    const int i = n % W, j = (n / W) % H, k = n / (W * H);
    const float rx = __fdiv_rn((float)i + 0.5f, (float)W);
    const float ry = __fdiv_rn((float)j + 0.5f, (float)H);
    const float rz = __fdiv_rn((float)k + 0.5f, (float)Z);
    const float px = __fadd_rn(__fadd_rn(__fmul_rn(rx, xr), xmin), origin[b * 3 + 0]);
    const float py = __fadd_rn(__fadd_rn(__fmul_rn(ry, yr), ymin), origin[b * 3 + 1]);
    const float pz = __fadd_rn(__fadd_rn(__fmul_rn(rz, zr), zmin), origin[b * 3 + 2]);
    const float eps = 1e-5f;
    unsigned bits = 0;
    for (int c = 0; c < Ncam; ++c) {
        const float* m = w2p + ((size_t)b * Ncam + c) * 16;
        float q[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            float acc = __fmul_rn(m[r * 4 + 0], px);
            acc = __fadd_rn(acc, __fmul_rn(m[r * 4 + 1], py));
            acc = __fadd_rn(acc, __fmul_rn(m[r * 4 + 2], pz));
            q[r] = __fadd_rn(acc, m[r * 4 + 3]);
        }
        bool ok = q[2] > eps;
        const float depth = fmaxf(q[2], eps);
        const float u = __fdiv_rn(__fdiv_rn(q[0], depth), img_w);
        const float v = __fdiv_rn(__fdiv_rn(q[1], depth), img_h);
        ok = ok && (v > 0.0f) && (v < 1.0f) && (u < 1.0f) && (u > 0.0f);
        float* o = uv + (((size_t)b * Ncam + c) * Nq + n) * 2;
        o[0] = u;
        o[1] = v;
        bits |= ok ? (1u << c) : 0u;
    }
    vis[(size_t)b * Nq + n] = (uint8_t)bits;
}

// bev_mask [Ncam, B, Nq, D] (reference layout) -> vis bits
__global__ __launch_bounds__(256) void k_mask_to_vis(const uint8_t* __restrict__ mask, int B, int Ncam,
                                                     int Nq, int D, uint8_t* __restrict__ vis) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (n >= Nq) return;
    unsigned bits = 0;
    for (int c = 0; c < Ncam; ++c) {
        const uint8_t* p = mask + (((size_t)c * B + b) * Nq + n) * D;
        bool any = false;
        for (int d = 0; d < D; ++d) any = any || (p[d] != 0);
        bits |= any ? (1u << c) : 0u;
    }
    vis[(size_t)b * Nq + n] = (uint8_t)bits;
}

// ordered (ascending voxel id) compaction of the visible / owned voxels of camera c.
__global__ __launch_bounds__(256) void k_build_lists(const uint8_t* __restrict__ vis, int Ncam, int Nq,
                                                     int* __restrict__ vis_list, int* __restrict__ vis_cnt,
                                                     int* __restrict__ own_list, int* __restrict__ own_cnt) {
    __shared__ int wsum[2][4];
    const int c = blockIdx.x, b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned long long lt = lane ? (~0ull >> (64 - lane)) : 0ull;
    int* vl = vis_list + ((size_t)b * Ncam + c) * Nq;
    int* ol = own_list + ((size_t)b * Ncam + c) * Nq;
    int base_v = 0, base_o = 0;
    for (int n0 = 0; n0 < Nq; n0 += 256) {
        const int n = n0 + tid;
        const bool valid = n < Nq;
        const unsigned m = valid ? vis[(size_t)b * Nq + n] : 0u;
        const bool is_v = valid && ((m >> c) & 1u);
        const int owner = m ? (__ffs((int)m) - 1) : (n % Ncam);
        const bool is_o = valid && owner == c;
        const unsigned long long bv = __ballot(is_v), bo = __ballot(is_o);
        if (lane == 0) {
            wsum[0][wave] = __popcll(bv);
            wsum[1][wave] = __popcll(bo);
        }
        __syncthreads();
        int off_v = base_v, off_o = base_o, tot_v = 0, tot_o = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            if (w < wave) {
                off_v += wsum[0][w];
                off_o += wsum[1][w];
            }
            tot_v += wsum[0][w];
            tot_o += wsum[1][w];
        }
        if (is_v) vl[off_v + __popcll(bv & lt)] = n;
        if (is_o) ol[off_o + __popcll(bo & lt)] = n;
        base_v += tot_v;
        base_o += tot_o;
        __syncthreads();
    }
    if (tid == 0) {
        vis_cnt[b * Ncam + c] = base_v;
        own_cnt[b * Ncam + c] = base_o;
    }
}

// ------------------------------------------------------------------------------------------
// Channel map of a lane inside its group of G lanes.  A lane carries CPL = HD/G channels as
// (at most) two segments so that every LDS access is the widest naturally aligned one and the
// compiler never has to fall back to ds_read2_b64 (half rate, MI355X_MICROARCH.md LDS table):
//   segment 0: W0 = min(4, CPL) channels at  gl*W0            (ds_read_b128 / b64)
//   segment 1: W1 = CPL - W0   channels at  G*W0 + gl*W1      (ds_read_b128 / b64 / none)
// e.g. HD=96, G=16: channels [4gl,4gl+4) and [64+2gl, 64+2gl+2).
template <int HD, int G>
struct ChMap {
    static constexpr int CPL = HD / G;
    static constexpr int W0 = CPL >= 4 ? 4 : 2;
    static constexpr int W1 = CPL - W0;
    static_assert(CPL == 2 || CPL == 4 || CPL == 6 || CPL == 8, "unsupported channels per lane");
    __device__ __forceinline__ static int off0(int gl) { return gl * W0; }
    __device__ __forceinline__ static int off1(int gl) { return G * W0 + gl * W1; }
};

template <int N>
__device__ __forceinline__ void load_vec(const float* p, float* v) {
    if constexpr (N == 4) {
        const float4 t = *reinterpret_cast<const float4*>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else if constexpr (N == 2) {
        const float2 t = *reinterpret_cast<const float2*>(p);
        v[0] = t.x; v[1] = t.y;
    }
}
template <int N>
__device__ __forceinline__ void load_vec(const uint16_t* p, float* v) {
    if constexpr (N == 4) {
        const uint2 t = *reinterpret_cast<const uint2*>(p);
        v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
        v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
    } else if constexpr (N == 2) {
        const uint32_t t = *reinterpret_cast<const uint32_t*>(p);
        v[0] = __uint_as_float(t << 16); v[1] = __uint_as_float(t & 0xffff0000u);
    }
}
template <int N>
__device__ __forceinline__ void store_vec(float* p, const float* v) {
    if constexpr (N == 4) *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    else if constexpr (N == 2) *reinterpret_cast<float2*>(p) = make_float2(v[0], v[1]);
}

// row = pointer to channel 0 of the (key|voxel, head) row
template <int HD, int G, typename VT>
__device__ __forceinline__ void load_ch(const VT* row, int gl, float (&v)[HD / G]) {
    using M = ChMap<HD, G>;
    load_vec<M::W0>(row + M::off0(gl), v);
    load_vec<M::W1>(row + M::off1(gl), v + M::W0);
}
template <int HD, int G>
__device__ __forceinline__ void store_ch(float* row, int gl, const float (&v)[HD / G]) {
    using M = ChMap<HD, G>;
    store_vec<M::W0>(row + M::off0(gl), v);
    store_vec<M::W1>(row + M::off1(gl), v + M::W0);
}
template <int HD, int G>
__device__ __forceinline__ void atomic_add_ch(float* row, int gl, float coef, const float (&g)[HD / G]) {
    using M = ChMap<HD, G>;
#pragma unroll
    for (int j = 0; j < M::W0; ++j) atomicAdd(row + M::off0(gl) + j, coef * g[j]);
#pragma unroll
    for (int j = 0; j < M::W1; ++j) atomicAdd(row + M::off1(gl) + j, coef * g[M::W0 + j]);
}

template <int P>
__device__ __forceinline__ void softmax_points(const float* lg, float (&a)[P]) {
    float mx = lg[0];
#pragma unroll
    for (int p = 1; p < P; ++p) mx = fmaxf(mx, lg[p]);
    float s = 0.0f;
#pragma unroll
    for (int p = 0; p < P; ++p) {
        a[p] = expf(lg[p] - mx);
        s += a[p];
    }
#pragma unroll
    for (int p = 0; p < P; ++p) a[p] = __fdiv_rn(a[p], s);
}

// one camera's contribution to one voxel-head row: sum_p a[p] * bilinear(map, uv + off)
template <int HD, int G, int P, typename VT>
__device__ __forceinline__ void gather_camera(const VT* base, size_t rstride, const float* u, int D,
                                              const float (&ox)[P], const float (&oy)[P],
                                              const float (&a)[P], int mh, int mw, int gl,
                                              float (&acc)[HD / G]) {
    constexpr int CPL = HD / G;
#pragma unroll
    for (int p = 0; p < P; ++p) {
        const int d = (D == 1) ? 0 : (p % D);
        Bilinear s;
        bilinear_setup<false>(u[2 * d] + ox[p], u[2 * d + 1] + oy[p], mh, mw, s);
        if (!s.any) continue;
        float val[CPL];
#pragma unroll
        for (int j = 0; j < CPL; ++j) val[j] = 0.0f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (s.w[k] != 0.0f) {
                float v[CPL];
                load_ch<HD, G, VT>(base + (size_t)s.key[k] * rstride, gl, v);
#pragma unroll
                for (int j = 0; j < CPL; ++j) val[j] += s.w[k] * v[j];
            }
        }
#pragma unroll
        for (int j = 0; j < CPL; ++j) acc[j] += a[p] * val[j];
    }
}

template <int HD, typename VT>
__device__ __forceinline__ void stage_tile(VT* tile, const VT* src, size_t rstride, int Nk, int nthreads) {
    constexpr int VEC = 16 / sizeof(VT);
    constexpr int VPR = HD / VEC;
    static_assert(HD % VEC == 0, "head_dim row must be a whole number of 16-byte vectors");
    for (int i = threadIdx.x; i < Nk * VPR; i += nthreads) {
        const int k = i / VPR, j = i - k * VPR;
        *reinterpret_cast<uint4*>(tile + k * HD + j * VEC) =
            *reinterpret_cast<const uint4*>(src + (size_t)k * rstride + j * VEC);
    }
}

// ------------------------------------------------------------------------------------------
template <int HD, int G, int P, typename VT>
__global__ __launch_bounds__(256) void k_sca_fwd(const VT* __restrict__ value, const float* __restrict__ offs,
                                                 const float* __restrict__ logits,
                                                 const float* __restrict__ uv, const uint8_t* __restrict__ vis,
                                                 const int* __restrict__ own_list,
                                                 const int* __restrict__ own_cnt, float* __restrict__ slots,
                                                 int Ncam, int Nq, int D, int heads, int mh, int mw, int nchunks,
                                                 int chunk) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    VT* tile = reinterpret_cast<VT*>(smem);
    constexpr int CPL = HD / G;
    constexpr int VPW = VER_WAVE / G;
    const int Nk = mh * mw;
    int bid = blockIdx.x;
    const int ck = bid % nchunks;
    bid /= nchunks;
    const int h = bid % heads;
    bid /= heads;
    const int c = bid % Ncam;
    const int b = bid / Ncam;
    const int cnt = own_cnt[b * Ncam + c];
    const int start = ck * chunk;
    if (start >= cnt) return;
    const int end = min(cnt, start + chunk);
    const size_t rstride = (size_t)heads * HD;
    const VT* vb = value + (size_t)b * Ncam * Nk * rstride + (size_t)h * HD;   // camera 0 of viewpoint b
    stage_tile<HD, VT>(tile, vb + (size_t)c * Nk * rstride, rstride, Nk, 256);
    __syncthreads();

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int sub = lane / G, gl = lane % G;
    const int* list = own_list + ((size_t)b * Ncam + c) * Nq;
    for (int i = start + wave * VPW + sub; i < end; i += 4 * VPW) {
        const int n = list[i];
        const unsigned m = vis[(size_t)b * Nq + n];
        float acc[CPL];
#pragma unroll
        for (int j = 0; j < CPL; ++j) acc[j] = 0.0f;
        if (m) {
            const size_t qh = ((size_t)b * Nq + n) * heads + h;
            float a[P], ox[P], oy[P];
            softmax_points<P>(logits + qh * P, a);
            const float* of = offs + qh * P * 2;
#pragma unroll
            for (int p = 0; p < P; ++p) {
                ox[p] = __fdiv_rn(of[2 * p], (float)mw);
                oy[p] = __fdiv_rn(of[2 * p + 1], (float)mh);
            }
            unsigned mm = m;
            while (mm) {
                const int cc = __ffs((int)mm) - 1;
                mm &= mm - 1;
                const float* u = uv + (((size_t)b * Ncam + cc) * Nq + n) * D * 2;
                float cam[CPL];
#pragma unroll
                for (int j = 0; j < CPL; ++j) cam[j] = 0.0f;
                if (cc == c)
                    gather_camera<HD, G, P, VT>(tile, (size_t)HD, u, D, ox, oy, a, mh, mw, gl, cam);
                else
                    gather_camera<HD, G, P, VT>(vb + (size_t)cc * Nk * rstride, rstride, u, D, ox, oy, a, mh,
                                                mw, gl, cam);
#pragma unroll
                for (int j = 0; j < CPL; ++j) acc[j] += cam[j];
            }
            const float cf = (float)__popc(m);
#pragma unroll
            for (int j = 0; j < CPL; ++j) acc[j] = __fdiv_rn(acc[j], cf);
        }
        store_ch<HD, G>(slots + ((size_t)b * Nq + n) * heads * HD + (size_t)h * HD, gl, acc);
    }
}

// ------------------------------------------------------------------------------------------
template <int HD, int G, int P, typename VT>
__global__ __launch_bounds__(512) void k_sca_bwd(const VT* __restrict__ value, const float* __restrict__ offs,
                                                 const float* __restrict__ logits,
                                                 const float* __restrict__ uv, const uint8_t* __restrict__ vis,
                                                 const int* __restrict__ vis_list,
                                                 const int* __restrict__ vis_cnt,
                                                 const float* __restrict__ gslots, float* gvalue, float* goffs,
                                                 float* glogits, int Ncam, int Nq, int D, int heads, int mh,
                                                 int mw, int nchunks, int chunk) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int CPL = HD / G;
    constexpr int VPW = VER_WAVE / G;
    const int Nk = mh * mw;
    float* gtile = reinterpret_cast<float*>(smem);                       // [Nk][HD] fp32 accumulators
    VT* tile = reinterpret_cast<VT*>(smem + (size_t)Nk * HD * sizeof(float));
    int bid = blockIdx.x;
    const int ck = bid % nchunks;
    bid /= nchunks;
    const int h = bid % heads;
    bid /= heads;
    const int c = bid % Ncam;
    const int b = bid / Ncam;
    const int cnt = vis_cnt[b * Ncam + c];
    const int start = ck * chunk;
    const bool atomic_flush = nchunks > 1;          // gvalue pre-zeroed by the host wrapper in that case
    if (start >= cnt && (atomic_flush || ck != 0)) return;
    const int end = min(cnt, start + chunk);
    const size_t rstride = (size_t)heads * HD;
    const size_t tbase = ((size_t)b * Ncam + c) * Nk * rstride + (size_t)h * HD;
    for (int i = threadIdx.x; i < Nk * HD; i += 512) gtile[i] = 0.0f;
    if (start < cnt) stage_tile<HD, VT>(tile, value + tbase, rstride, Nk, 512);
    __syncthreads();

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int sub = lane / G, gl = lane % G;
    const int* list = vis_list + ((size_t)b * Ncam + c) * Nq;
    // all G lanes of a group run the same trip count; groups past `end` idle through the shuffles
    const int iters = (end - start + 8 * VPW - 1) / (8 * VPW);
    for (int it = 0; it < iters; ++it) {
        const int i = start + it * 8 * VPW + wave * VPW + sub;
        const bool live = i < end;
        const int n = live ? list[i] : 0;
        const unsigned m = live ? vis[(size_t)b * Nq + n] : 1u;
        const float cf = (float)__popc(m);
        const size_t qh = ((size_t)b * Nq + n) * heads + h;
        float g[CPL];
        if (live) {
            load_ch<HD, G, float>(gslots + ((size_t)b * Nq + n) * heads * HD + (size_t)h * HD, gl, g);
#pragma unroll
            for (int j = 0; j < CPL; ++j) g[j] = __fdiv_rn(g[j], cf);
        } else {
#pragma unroll
            for (int j = 0; j < CPL; ++j) g[j] = 0.0f;
        }
        float a[P], dA[P], dX[P], dY[P];
        softmax_points<P>(logits + qh * P, a);
        const float* of = offs + qh * P * 2;
        const float* u = uv + (((size_t)b * Ncam + c) * Nq + n) * D * 2;
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const int d = (D == 1) ? 0 : (p % D);
            Bilinear s;
            bilinear_setup<true>(u[2 * d] + __fdiv_rn(of[2 * p], (float)mw),
                                 u[2 * d + 1] + __fdiv_rn(of[2 * p + 1], (float)mh), mh, mw, s);
            float sa = 0.0f, sx = 0.0f, sy = 0.0f;
            if (live && s.any) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (s.w[k] == 0.0f && s.gx[k] == 0.0f && s.gy[k] == 0.0f) continue;
                    float v[CPL];
                    load_ch<HD, G, VT>(tile + (size_t)s.key[k] * HD, gl, v);
                    float dk = 0.0f;
#pragma unroll
                    for (int j = 0; j < CPL; ++j) dk += g[j] * v[j];
                    sa += s.w[k] * dk;
                    sx += s.gx[k] * dk;
                    sy += s.gy[k] * dk;
                    const float coef = a[p] * s.w[k];
                    if (coef != 0.0f) atomic_add_ch<HD, G>(gtile + (size_t)s.key[k] * HD, gl, coef, g);
                }
            }
            dA[p] = group_sum<G>(sa);
            dX[p] = group_sum<G>(sx);
            dY[p] = group_sum<G>(sy);
        }
        if (live && gl == 0) {
            float dot = 0.0f;
#pragma unroll
            for (int p = 0; p < P; ++p) dot += a[p] * dA[p];
            float* go = goffs + qh * P * 2;
            float* gw = glogits + qh * P;
            // x_pix = (u + off/W)*W - 0.5  =>  d x_pix / d off = 1
            if (cf == 1.0f) {
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    go[2 * p] = a[p] * dX[p];
                    go[2 * p + 1] = a[p] * dY[p];
                    gw[p] = a[p] * (dA[p] - dot);
                }
            } else {
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    atomicAdd(go + 2 * p, a[p] * dX[p]);
                    atomicAdd(go + 2 * p + 1, a[p] * dY[p]);
                    atomicAdd(gw + p, a[p] * (dA[p] - dot));
                }
            }
        }
    }
    __syncthreads();
    float* gv = gvalue + tbase;
    if (!atomic_flush) {
        constexpr int VPR = HD / 4;
        for (int i = threadIdx.x; i < Nk * VPR; i += 512) {
            const int k = i / VPR, j = i - k * VPR;
            *reinterpret_cast<float4*>(gv + (size_t)k * rstride + j * 4) =
                *reinterpret_cast<const float4*>(gtile + k * HD + j * 4);
        }
    } else {
        for (int i = threadIdx.x; i < Nk * HD; i += 512) {
            const int k = i / HD, j = i - k * HD;
            const float t = gtile[i];
            if (t != 0.0f) atomicAdd(gv + (size_t)k * rstride + j, t);
        }
    }
}

// ------------------------------------------------------------------------------------------
namespace {

constexpr int kFwdChunk = 2048;
constexpr int kBwdChunk = 4096;
constexpr size_t kMaxLds = 160 * 1024;

template <typename F>
int dispatch_shape(int hd, int points, F&& f) {
#define VER_CASE(HD_, G_)                                                                              \
    if (hd == HD_) {                                                                                   \
        if (points == 8) return f(std::integral_constant<int, HD_>(), std::integral_constant<int, G_>(), \
                                  std::integral_constant<int, 8>());                                   \
        return f(std::integral_constant<int, HD_>(), std::integral_constant<int, G_>(),                \
                 std::integral_constant<int, 4>());                                                    \
    }
    VER_CASE(8, 4)
    VER_CASE(16, 8)
    VER_CASE(32, 16)
    VER_CASE(64, 16)
    VER_CASE(96, 16)
    VER_CASE(128, 16)
#undef VER_CASE
    return ver_fail(VER_EUNSUPPORTED, "ver_sca: head_dim %d not in {8,16,32,64,96,128}", hd);
}

int check_sca(const void* value, int vdt, const void* a, const void* b, const void* c, const void* d,
              const void* e, const void* f, int B, int Ncam, int Nq, int D, int heads, int hd, int points,
              int mh, int mw) {
    VER_REQUIRE(value && a && b && c && d && e && f, VER_EINVAL, "ver_sca: null pointer argument");
    VER_REQUIRE(vdt == VER_F32, VER_EUNSUPPORTED, "ver_sca: value_dtype %d not built (fp32 only)", vdt);
    VER_REQUIRE(B >= 0 && Nq >= 0, VER_EINVAL, "ver_sca: negative batch/voxel count");
    VER_REQUIRE(Ncam >= 1 && Ncam <= 8, VER_EUNSUPPORTED, "ver_sca: Ncam %d outside 1..8", Ncam);
    VER_REQUIRE(heads > 0 && mh > 0 && mw > 0 && D > 0, VER_EINVAL, "ver_sca: non-positive size");
    VER_REQUIRE(points == 4 || points == 8, VER_EUNSUPPORTED, "ver_sca: points %d not in {4,8}", points);
    VER_REQUIRE(points % D == 0, VER_EINVAL, "ver_sca: anchors D=%d must divide points=%d", D, points);
    VER_REQUIRE(((uintptr_t)value & 15) == 0, VER_EINVAL, "ver_sca: value must be 16-byte aligned");
    (void)hd;
    return VER_OK;
}

}  // namespace

extern "C" int ver_project_points(const float* world2pixel, const float* origin, const float* pc_range, int B,
                                  int Ncam, int bev_z, int bev_h, int bev_w, float img_w, float img_h,
                                  float* uv, uint8_t* vis, int32_t* vis_list, int32_t* vis_cnt,
                                  int32_t* own_list, int32_t* own_cnt, void* stream) {
    VER_REQUIRE(world2pixel && origin && pc_range && uv && vis && vis_list && vis_cnt && own_list && own_cnt,
                VER_EINVAL, "ver_project_points: null pointer argument");
    VER_REQUIRE(Ncam >= 1 && Ncam <= 8, VER_EUNSUPPORTED, "ver_project_points: Ncam %d outside 1..8", Ncam);
    VER_REQUIRE(B >= 0 && bev_z > 0 && bev_h > 0 && bev_w > 0, VER_EINVAL, "ver_project_points: bad grid");
    if (B == 0) return VER_OK;
    const int Nq = bev_z * bev_h * bev_w;
    hipStream_t st = (hipStream_t)stream;
    // (max - min) in double then to fp32, as the reference's Python scalars do (voxel_encoder.py:146-151)
    const float xr = (float)((double)pc_range[3] - (double)pc_range[0]);
    const float yr = (float)((double)pc_range[4] - (double)pc_range[1]);
    const float zr = (float)((double)pc_range[5] - (double)pc_range[2]);
    hipLaunchKernelGGL(k_project, dim3((Nq + 255) / 256, B), dim3(256), 0, st, world2pixel, origin, pc_range[0],
                       pc_range[1], pc_range[2], xr, yr, zr, Ncam, bev_z, bev_h, bev_w, img_w, img_h, uv, vis);
    int rc = ver_check_launch("ver_project_points/k_project");
    if (rc) return rc;
    hipLaunchKernelGGL(k_build_lists, dim3(Ncam, B), dim3(256), 0, st, vis, Ncam, Nq, vis_list, vis_cnt,
                       own_list, own_cnt);
    return ver_check_launch("ver_project_points/k_build_lists");
}

extern "C" int ver_hits_from_mask(const uint8_t* bev_mask, int B, int Ncam, int Nq, int D, uint8_t* vis,
                                  int32_t* vis_list, int32_t* vis_cnt, int32_t* own_list, int32_t* own_cnt,
                                  void* stream) {
    VER_REQUIRE(bev_mask && vis && vis_list && vis_cnt && own_list && own_cnt, VER_EINVAL,
                "ver_hits_from_mask: null pointer argument");
    VER_REQUIRE(Ncam >= 1 && Ncam <= 8, VER_EUNSUPPORTED, "ver_hits_from_mask: Ncam %d outside 1..8", Ncam);
    VER_REQUIRE(B >= 0 && Nq > 0 && D > 0, VER_EINVAL, "ver_hits_from_mask: bad sizes");
    if (B == 0) return VER_OK;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_mask_to_vis, dim3((Nq + 255) / 256, B), dim3(256), 0, st, bev_mask, B, Ncam, Nq, D, vis);
    int rc = ver_check_launch("ver_hits_from_mask/k_mask_to_vis");
    if (rc) return rc;
    hipLaunchKernelGGL(k_build_lists, dim3(Ncam, B), dim3(256), 0, st, vis, Ncam, Nq, vis_list, vis_cnt,
                       own_list, own_cnt);
    return ver_check_launch("ver_hits_from_mask/k_build_lists");
}

extern "C" int ver_sca_forward(const void* value, int value_dtype, const float* offsets, const float* logits,
                               const float* uv, const uint8_t* vis, const int32_t* own_list,
                               const int32_t* own_cnt, float* slots, int B, int Ncam, int Nq, int D, int heads,
                               int head_dim, int points, int map_h, int map_w, void* stream) {
    int rc = check_sca(value, value_dtype, offsets, logits, uv, vis, own_list, own_cnt, B, Ncam, Nq, D, heads,
                       head_dim, points, map_h, map_w);
    if (rc) return rc;
    VER_REQUIRE(slots, VER_EINVAL, "ver_sca_forward: slots is null");
    if (B == 0 || Nq == 0) return VER_OK;
    const size_t lds = (size_t)map_h * map_w * head_dim * sizeof(float);
    VER_REQUIRE(lds <= kMaxLds, VER_EUNSUPPORTED,
                "ver_sca_forward: %dx%dx%d value tile (%zu B) exceeds the 160 KiB LDS", map_h, map_w, head_dim,
                lds);
    const int nchunks = (Nq + kFwdChunk - 1) / kFwdChunk;
    hipStream_t st = (hipStream_t)stream;
    return dispatch_shape(head_dim, points, [&](auto hd, auto g, auto pp) {
        constexpr int HD = decltype(hd)::value, G = decltype(g)::value, P = decltype(pp)::value;
        auto kern = k_sca_fwd<HD, G, P, float>;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return ver_fail(VER_ELAUNCH, "ver_sca_forward: LDS attribute: %s", hipGetErrorString(e));
        const unsigned blocks = (unsigned)B * Ncam * heads * nchunks;
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, st, (const float*)value, offsets, logits, uv, vis,
                           own_list, own_cnt, slots, Ncam, Nq, D, heads, map_h, map_w, nchunks, kFwdChunk);
        return ver_check_launch("ver_sca_forward");
    });
}

extern "C" int ver_sca_backward(const void* value, int value_dtype, const float* offsets, const float* logits,
                                const float* uv, const uint8_t* vis, const int32_t* vis_list,
                                const int32_t* vis_cnt, const float* grad_slots, float* grad_value,
                                float* grad_offsets, float* grad_logits, int B, int Ncam, int Nq, int D,
                                int heads, int head_dim, int points, int map_h, int map_w, void* stream) {
    int rc = check_sca(value, value_dtype, offsets, logits, uv, vis, vis_list, vis_cnt, B, Ncam, Nq, D, heads,
                       head_dim, points, map_h, map_w);
    if (rc) return rc;
    VER_REQUIRE(grad_slots && grad_value && grad_offsets && grad_logits, VER_EINVAL,
                "ver_sca_backward: null gradient pointer");
    if (B == 0 || Nq == 0) return VER_OK;
    const size_t lds = (size_t)map_h * map_w * head_dim * (sizeof(float) + sizeof(float));
    VER_REQUIRE(lds <= kMaxLds, VER_EUNSUPPORTED,
                "ver_sca_backward: %dx%dx%d tiles (%zu B) exceed LDS", map_h, map_w, head_dim, lds);
    const int nchunks = (Nq + kBwdChunk - 1) / kBwdChunk;
    hipStream_t st = (hipStream_t)stream;
    const size_t nsmall = (size_t)B * Nq * heads * points;
    hipError_t e = hipMemsetAsync(grad_offsets, 0, nsmall * 2 * sizeof(float), st);
    if (e == hipSuccess) e = hipMemsetAsync(grad_logits, 0, nsmall * sizeof(float), st);
    if (e == hipSuccess && nchunks > 1)
        e = hipMemsetAsync(grad_value, 0, (size_t)B * Ncam * map_h * map_w * heads * head_dim * sizeof(float), st);
    if (e != hipSuccess) return ver_fail(VER_ELAUNCH, "ver_sca_backward: memset: %s", hipGetErrorString(e));
    return dispatch_shape(head_dim, points, [&](auto hd, auto g, auto pp) {
        constexpr int HD = decltype(hd)::value, G = decltype(g)::value, P = decltype(pp)::value;
        auto kern = k_sca_bwd<HD, G, P, float>;
        hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e2 != hipSuccess)
            return ver_fail(VER_ELAUNCH, "ver_sca_backward: LDS attribute: %s", hipGetErrorString(e2));
        const unsigned blocks = (unsigned)B * Ncam * heads * nchunks;
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), lds, st, (const float*)value, offsets, logits, uv, vis,
                           vis_list, vis_cnt, grad_slots, grad_value, grad_offsets, grad_logits, Ncam, Nq, D,
                           heads, map_h, map_w, nchunks, kBwdChunk);
        return ver_check_launch("ver_sca_backward");
    });
}
