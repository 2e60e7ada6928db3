// Fused multi-view gather for gfx950: projection + hit table + SCA forward/backward
// (C ABI in include/ver_ops.h; reference code being replaced cited there).
//
// Data layout in HBM (per launch, B viewpoints):
//   value  [B, Ncam, Nk, heads, HD]   one (camera, head) slice = Nk rows of HD contiguous
//                                      elements (384 B at HD=96 fp32) inside 768-wide token rows
//   uv     [B, Ncam, Nq, D, 2]  vis [B, Nq]  lists [B, Ncam, Nq] + counts [B, Ncam]
//   offsets/logits/slots are voxel-major: row (b, n) holds all heads.
//
// Forward (ver_sca_forward): k_zero_rows zero-fills the output rows of voxels NOT seen by exactly one camera,
// then one workgroup per (viewpoint, camera, head [, list chunk]) stages the 14x14xHD value tile of its
// (camera, head) into LDS by LDS-DMA and gathers from it (k_sca_fwd_cs for 8 points and head_dim % 32 == 0,
// k_sca_fwd otherwise).  Rows of voxels seen by exactly one camera are written with plain stores; rows seen by
// several cameras are accumulated with fp32 atomic adds on the zeroed rows, so the DETERMINISM CONTRACT is:
// bitwise run-to-run reproducible as long as no voxel is seen by more than two cameras (two addends commute);
// with three or more the last bit may vary with the order of the adds.  Voxels seen by no camera get exact zeros.
//
// On bf16 tiles k_sca_fwd_cs converts the staged tile to fp16 in LDS and accumulates the <= 8 points of a (voxel, head,
// corner) with v_pk_fma_f16 (VER_SCA_FWD_MATH=0: bf16 tile unpacked to fp32 per use, exact fp32 accumulation).
//
// Backward (ver_sca_backward), by shape:
//   * bf16 tiles, 8 points, head_dim % 32 == 0, tile + buffers <= 160 KB  ->  k_sca_bwd_mm: ONE kernel on the matrix
//     cores.  Per 32-voxel chunk D^T = V x G^T (all tile-row x voxel dots, bf16 hi + lo split of the grad rows) gives
//     d(offsets) / d(logits); the sampling events are added into a dense S^T in LDS as 2^-30 fixed point (integer LDS
//     atomics: exact in any order) and d(value) = S^T x G accumulates in registers over all chunks: d(value) is bitwise
//     reproducible and is written in bf16 (ver_sca_backward_grad_dtype tells the caller);
//   * fp32 tiles / other shapes with 16 lanes per voxel  ->  k_sca_bwd_off (d offsets, d logits; the forward's
//     structure) + k_sca_bwd_val (d value by a counting sort of the sampling events per tile row, no floating-point
//     atomics in LDS; the order of a row's events inside the sort is not fixed, so d(value) may differ in the last
//     bit run to run, as the reference's CUDA backward does);
//   * head_dim < 32 or large maps  ->  k_sca_bwd, a single LDS-atomic kernel.
#include <cstdlib>
#include <type_traits>
#include "ver_common.h"

constexpr int kFwdThreads = 1024;

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

// ordered (ascending voxel id) compaction: per camera the voxels it sees (= the reference's
// `indexes[c]`), and per viewpoint (built by the camera-0 workgroup) the voxels NOT seen by exactly
// one camera -- their output rows are zero-filled before the gather (unseen: stay zero; seen by
// several cameras: accumulated with atomics).  fwd_list is the forward kernel's work order for the
// camera: voxels only this camera sees first (plain stores), voxels shared with other cameras from the
// back of the array (atomic adds); fwd_cnt = {#single, #shared}.
__global__ __launch_bounds__(256) void k_build_lists(const uint8_t* __restrict__ vis, int Ncam, int Nq,
                                                     int* __restrict__ vis_list, int* __restrict__ vis_cnt,
                                                     int* __restrict__ zero_list, int* __restrict__ zero_cnt,
                                                     int* __restrict__ fwd_list, int* __restrict__ fwd_cnt) {
    __shared__ int wsum[4][4];
    const int c = blockIdx.x, b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned long long lt = lane ? (~0ull >> (64 - lane)) : 0ull;
    int* vl = vis_list + ((size_t)b * Ncam + c) * Nq;
    int* zl = zero_list + (size_t)b * Nq;
    int* fl = fwd_list + ((size_t)b * Ncam + c) * Nq;
    int base_v = 0, base_z = 0, base_s = 0, base_m = 0;
    for (int n0 = 0; n0 < Nq; n0 += 256) {
        const int n = n0 + tid;
        const bool valid = n < Nq;
        const unsigned m = valid ? vis[(size_t)b * Nq + n] : 0u;
        const bool is_v = valid && ((m >> c) & 1u);
        const bool is_z = valid && c == 0 && __popc(m) != 1;
        const bool is_s = is_v && __popc(m) == 1, is_m = is_v && __popc(m) > 1;
        const unsigned long long bv = __ballot(is_v), bz = __ballot(is_z), bs = __ballot(is_s), bm = __ballot(is_m);
        if (lane == 0) {
            wsum[0][wave] = __popcll(bv);
            wsum[1][wave] = __popcll(bz);
            wsum[2][wave] = __popcll(bs);
            wsum[3][wave] = __popcll(bm);
        }
        __syncthreads();
        int off[4] = {base_v, base_z, base_s, base_m}, tot[4] = {0, 0, 0, 0};
#pragma unroll
        for (int w = 0; w < 4; ++w) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (w < wave) off[q] += wsum[q][w];
                tot[q] += wsum[q][w];
            }
        }
        if (is_v) vl[off[0] + __popcll(bv & lt)] = n;
        if (is_z) zl[off[1] + __popcll(bz & lt)] = n;
        if (is_s) fl[off[2] + __popcll(bs & lt)] = n;                    // single-camera voxels from the front
        if (is_m) fl[Nq - 1 - (off[3] + __popcll(bm & lt))] = n;         // multi-camera voxels from the back
        base_v += tot[0];
        base_z += tot[1];
        base_s += tot[2];
        base_m += tot[3];
        __syncthreads();
    }
    if (tid == 0) {
        vis_cnt[b * Ncam + c] = base_v;
        fwd_cnt[(b * Ncam + c) * 2 + 0] = base_s;
        fwd_cnt[(b * Ncam + c) * 2 + 1] = base_m;
        if (c == 0) zero_cnt[b] = base_z;
    }
}

// zero the output rows listed in zero_list (C floats per row), 8 rows per workgroup
__global__ __launch_bounds__(256) void k_zero_rows(const int* __restrict__ zero_list,
                                                   const int* __restrict__ zero_cnt, float* __restrict__ slots,
                                                   int Nq, int C) {
    const int b = blockIdx.y;
    const int e = blockIdx.x * 8 + (threadIdx.x >> 5);
    if (e >= zero_cnt[b]) return;
    const int n = zero_list[(size_t)b * Nq + e];
    float4* row = reinterpret_cast<float4*>(slots + ((size_t)b * Nq + n) * C);
    for (int i = threadIdx.x & 31; i < C / 4; i += 32) row[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
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
    // G = 16, 16-byte segment-0 reads: a ds_read_b128 is served in lane groups {0-3,12-15,20-27}, ... that mix
    // two voxels' lanes; with rows of HD*4 B = 32 banks (mod 64) apart the plain map collides 2-way whenever
    // the two tile rows differ in parity.  Swapping the lane quads 8-11 <-> 12-15 makes every group's two
    // bank sets complementary AND invariant under the 32-bank shift: conflict-free for any row pair.
    __device__ __forceinline__ static int off0(int gl) {
        if (G == 16 && W0 == 4) gl = (gl & 8) ? (gl ^ 4) : gl;
        return gl * W0;
    }
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
// NT: non-temporal store, for the forward's output rows (written once, read by a later kernel: the forward
// gather is 5 % faster with it; the backward kernels showed no gain and keep plain stores).
template <int N, bool NT = false>
__device__ __forceinline__ void store_vec(float* p, const float* v) {
    typedef float f32x4_t __attribute__((ext_vector_type(4)));
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    if constexpr (N == 4) {
        if constexpr (NT) __builtin_nontemporal_store(f32x4_t{v[0], v[1], v[2], v[3]}, reinterpret_cast<f32x4_t*>(p));
        else *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    } else if constexpr (N == 2) {
        if constexpr (NT) __builtin_nontemporal_store(f32x2_t{v[0], v[1]}, reinterpret_cast<f32x2_t*>(p));
        else *reinterpret_cast<float2*>(p) = make_float2(v[0], v[1]);
    }
}

// row = pointer to channel 0 of the (key|voxel, head) row
template <int HD, int G, typename VT>
__device__ __forceinline__ void load_ch(const VT* row, int gl, float (&v)[HD / G]) {
    using M = ChMap<HD, G>;
    load_vec<M::W0>(row + M::off0(gl), v);
    load_vec<M::W1>(row + M::off1(gl), v + M::W0);
}
template <int HD, int G, bool NT = false>
__device__ __forceinline__ void store_ch(float* row, int gl, const float (&v)[HD / G]) {
    using M = ChMap<HD, G>;
    store_vec<M::W0, NT>(row + M::off0(gl), v);
    store_vec<M::W1, NT>(row + M::off1(gl), v + M::W0);
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

// Stage the [Nk][HD] value tile of one (camera, head) into LDS with LDS-DMA
// (global_load_lds_dwordx4: 16 B per lane straight into LDS, no VGPR round trip, every wave
// keeps all of its ~18 requests in flight at once).  The LDS image is dense row-major, so the
// lane-linear destination (wave-uniform base + lane*16) is exactly chunk q = q0 + lane; the
// per-lane SOURCE address carries the row stride of the 768-wide token rows.
template <int HD, typename VT>
__device__ __forceinline__ void stage_tile(VT* tile, const VT* src, size_t rstride, int Nk, int wave, int nwaves) {
    constexpr int VEC = 16 / sizeof(VT);
    constexpr int VPR = HD / VEC;
    static_assert(HD % VEC == 0, "head_dim row must be a whole number of 16-byte vectors");
    const int total = Nk * VPR;
    const int lane = threadIdx.x & 63;
    for (int q0 = wave * 64; q0 < total; q0 += nwaves * 64) {
        const int q = q0 + lane;
        if (q < total) {
            const int k = q / VPR, j = q - k * VPR;
            const VT* g = src + (size_t)k * rstride + j * VEC;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                             (__attribute__((address_space(3))) void*)(tile + (size_t)q0 * VEC),
                                             16, 0, 0);
        }
    }
}

__device__ __forceinline__ void stage_wait() {
    __builtin_amdgcn_s_waitcnt(0);   // vmcnt(0): this wave's LDS-DMA requests have landed
    __syncthreads();
}

// ------------------------------------------------------------------------------------------
// Forward.  One workgroup of 16 waves per (viewpoint, camera, head group [, list chunk]) walks
// its heads with a double-buffered LDS tile:
//   * the last kFwdLoaders waves are LOADERS: they only issue LDS-DMA (global_load_lds_dwordx4)
//     for the next head's 14x14xHD tile while the other waves work on the current one, so after
//     the first tile the HBM stream of the value tensor hides behind compute (one barrier / head);
//   * the other waves are CONSUMERS.  A wave takes 4 voxels per iteration, one per 16-lane DPP row,
//     with two lane roles inside a row so that per-sample arithmetic is done once, not once per
//     channel lane:
//       phase A  lane = (point, corner subset): loads ITS logit / offset / uv, softmax over the P
//                points by DPP butterfly, sets up the bilinear footprint and keeps
//                {softmax weight * corner weight / #cams} and the tile rows of its corners;
//       phase B  lane = channel chunk (HD/16 channels): walks the P points; the point's weights and
//                rows arrive by DPP row_newbcast (fused into the FMAs -- no LDS traffic, no
//                ds_bpermute), then 4 corner reads (ds_read_b128 + ds_read_b64) + FMAs.
//     Global operand loads are software pipelined one iteration ahead (voxel ids two ahead).
// Every (camera, visible voxel) pair is handled by the workgroup that has that camera's tile in
// LDS.  Rows of voxels seen by exactly one camera are plain stores; rows seen by several cameras
// were zeroed by k_zero_rows and are accumulated with fp32 atomics (2 addends commute exactly, so
// results stay bitwise reproducible as long as no voxel is seen by more than two cameras).
#ifndef VER_FWD_LOADERS
#define VER_FWD_LOADERS 0
#endif
constexpr int kFwdLoaders = VER_FWD_LOADERS;   // 0: every wave issues its share of the next tile's LDS-DMA

#ifdef VER_DEBUG_TIMELINE
// timeline of 32 probe workgroups of k_sca_fwd_cs (scratch/r04/timeline_cs.py; build with -DVER_DEBUG_TIMELINE):
// g_tl[probe][wave][event] = s_memtime
constexpr int kTlProbes = 32;
__device__ long long g_tl[kTlProbes * 16 * 64];
__device__ __forceinline__ int tl_probe() {
    const int nb = gridDim.x;
    for (int i = 0; i < kTlProbes; ++i)
        if ((int)blockIdx.x == (int)(((long)nb * (2 * i + 1)) / (2 * kTlProbes))) return i;
    return -1;
}
// (VER_TLQ: the probe index looked up ONCE per kernel -- `const int tlq_pr = tl_probe();` -- the 32-step search of VER_TL costs a
//  stamp ~2 k cycles, more than the phases of k_sca_bwd_mm it is meant to time)
#define VER_TLQ_INIT const int tlq_pr = tl_probe()
#define VER_TLQ(ev)                                                                                       \
    do {                                                                                                  \
        if (tlq_pr >= 0 && (threadIdx.x & 63) == 0 && (ev) < 64)                                           \
            g_tl[(tlq_pr * 16 + (threadIdx.x >> 6)) * 64 + (ev)] = (long long)__builtin_amdgcn_s_memtime(); \
    } while (0)
#define VER_TL(ev)                                                                                        \
    do {                                                                                                  \
        const int pr_ = tl_probe();                                                                       \
        if (pr_ >= 0 && (threadIdx.x & 63) == 0 && (ev) < 64)                                              \
            g_tl[(pr_ * 16 + (threadIdx.x >> 6)) * 64 + (ev)] = (long long)__builtin_amdgcn_s_memtime();  \
    } while (0)
extern "C" int ver_timeline_read(long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tl), n * sizeof(long long));
}
#else
#define VER_TL(ev) do { } while (0)
#define VER_TLQ(ev) do { } while (0)
#define VER_TLQ_INIT do { } while (0)
#endif

template <int N>
__device__ __forceinline__ float row_bcast_f(float v) {   // every lane reads lane N of its own 16-lane row
    // bound_ctrl = true: every lane of a row_newbcast is written, so no "old" value has to be
    // materialised in front of the v_mov_b32_dpp (it was one extra v_mov per broadcast)
    return __builtin_bit_cast(float,
                              __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x150 + N, 0xf, 0xf, true));
}
template <int N>
__device__ __forceinline__ unsigned row_bcast_u(unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x150 + N, 0xf, 0xf, true);
}

template <int HD, int G>
__device__ __forceinline__ void emit_row(float* row, int gl, const float (&acc)[HD / G], bool single) {
    if (single) {
        store_ch<HD, G, true>(row, gl, acc);
    } else {
        atomic_add_ch<HD, G>(row, gl, 1.0f, acc);
    }
}

typedef __attribute__((address_space(3))) const unsigned char lds_byte;

// LDS byte address = base + (row offset held by lane N of this 16-lane row).  (Fusing the broadcast
// into the add as inline-asm v_add_u32_dpp was measured 25 % SLOWER: the asm statements defeat the
// scheduler's interleaving of address arithmetic, ds_reads and FMAs.)
template <int N>
__device__ __forceinline__ unsigned row_bcast_u(unsigned v);
template <int N>
__device__ __forceinline__ unsigned bcast_add(unsigned koff, unsigned base) {
    return base + row_bcast_u<N>(koff);
}

template <typename VT>
__device__ __forceinline__ const VT* lds_ptr(unsigned addr) {
    return reinterpret_cast<const VT*>((const unsigned char*)(lds_byte*)(uintptr_t)addr);
}

// one point of phase B: weights / row byte offsets of point PT come from lanes PT*LPP .. of the row.
// base0 / base1 = LDS byte address of this lane's two channel segments (ChMap<HD,16>) in tile row 0.
template <int HD, int P, int PT, typename VT>
__device__ __forceinline__ void consume_point(unsigned base0, unsigned base1, const float (&wsel)[2],
                                              const unsigned (&ksel)[2], float (&acc)[HD / 16]) {
    constexpr int LPP = 16 / P;          // lanes per point (2 for P=8, 4 for P=4)
    constexpr int CPN = 4 / LPP;         // corners per lane (2 / 1)
    using M = ChMap<HD, 16>;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int reg = t % CPN;
        constexpr int L1 = PT * LPP + (LPP > 1 ? 1 : 0), L2 = PT * LPP + (LPP > 2 ? 2 : 0),
                      L3 = PT * LPP + (LPP > 3 ? 3 : 0);
        float w;
        unsigned a0, a1 = 0;             // LDS byte addresses of the corner's tile row, this lane's segments
        if (t / CPN == 0) {
            w = row_bcast_f<PT * LPP>(wsel[reg]);
            a0 = bcast_add<PT * LPP>(ksel[reg], base0);
            if (M::W1 > 0) a1 = bcast_add<PT * LPP>(ksel[reg], base1);
        } else if (t / CPN == 1) {
            w = row_bcast_f<L1>(wsel[reg]);
            a0 = bcast_add<L1>(ksel[reg], base0);
            if (M::W1 > 0) a1 = bcast_add<L1>(ksel[reg], base1);
        } else if (t / CPN == 2) {
            w = row_bcast_f<L2>(wsel[reg]);
            a0 = bcast_add<L2>(ksel[reg], base0);
            if (M::W1 > 0) a1 = bcast_add<L2>(ksel[reg], base1);
        } else {
            w = row_bcast_f<L3>(wsel[reg]);
            a0 = bcast_add<L3>(ksel[reg], base0);
            if (M::W1 > 0) a1 = bcast_add<L3>(ksel[reg], base1);
        }
        float v[M::CPL];
        load_vec<M::W0>(lds_ptr<VT>(a0), v);
        load_vec<M::W1>(lds_ptr<VT>(a1), v + M::W0);
#pragma unroll
        for (int j = 0; j < M::CPL; ++j) acc[j] += w * v[j];
    }
}

template <int HD, int P, int PT, typename VT>
struct PointLoop {
    __device__ __forceinline__ static void run(unsigned base0, unsigned base1, const float (&wsel)[2],
                                               const unsigned (&ksel)[2], float (&acc)[HD / 16]) {
        consume_point<HD, P, PT, VT>(base0, base1, wsel, ksel, acc);
        PointLoop<HD, P, PT + 1, VT>::run(base0, base1, wsel, ksel, acc);
    }
};
template <int HD, int P, typename VT>
struct PointLoop<HD, P, P, VT> {
    __device__ __forceinline__ static void run(unsigned, unsigned, const float (&)[2], const unsigned (&)[2],
                                               float (&)[HD / 16]) {}
};

template <int HD, int G, int P, typename VT>
__global__ __launch_bounds__(kFwdThreads) void k_sca_fwd(
    const VT* __restrict__ value, const float* __restrict__ offs, const float* __restrict__ logits,
    const float* __restrict__ uv, const uint8_t* __restrict__ vis, const int* __restrict__ vis_list,
    const int* __restrict__ vis_cnt, float* slots, int Ncam, int Nq, int D, int heads, int mh, int mw,
    int nchunks, int chunk, int hsplit, int nbuf) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int CPL = HD / G;
    const int nwaves = (int)(blockDim.x >> 6);                         // 16 (or 8 with two workgroups per CU)
    const int NCONS = nwaves - kFwdLoaders;                            // consumer waves
    const int Nk = mh * mw;
    const size_t tile_elems = (size_t)Nk * HD;
    VT* tiles = reinterpret_cast<VT*>(smem);
    int bid = blockIdx.x;
    const int ck = bid % nchunks;
    bid /= nchunks;
    const int hs = bid % hsplit;
    bid /= hsplit;
    const int c = bid % Ncam;
    const int b = bid / Ncam;
    const int cnt = vis_cnt[b * Ncam + c];
    const int start = ck * chunk;
    if (start >= cnt) return;
    const int end = min(cnt, start + chunk);
    const int heads_per = heads / hsplit, h0 = hs * heads_per;
    const size_t rstride = (size_t)heads * HD;
    const VT* vown = value + ((size_t)b * Ncam + c) * Nk * rstride;   // this camera, head 0

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool loader = kFwdLoaders > 0 && wave >= NCONS;
    const int dma_wave = kFwdLoaders > 0 ? wave - NCONS : wave;
    const int kDmaWaves = kFwdLoaders > 0 ? kFwdLoaders : nwaves;
    const bool issues_dma = kFwdLoaders == 0 || loader;
    const int* list = vis_list + ((size_t)b * Ncam + c) * Nq;
    const float inv_w = 1.0f / (float)mw, inv_h = 1.0f / (float)mh;

    if (issues_dma && nbuf == 2) stage_tile<HD, VT>(tiles, vown + (size_t)h0 * HD, rstride, Nk, dma_wave, kDmaWaves);

    for (int hh = 0; hh < heads_per; ++hh) {
        const int h = h0 + hh;
        const int cur = nbuf == 2 ? (hh & 1) : 0;
        VT* tile = tiles + cur * tile_elems;
        if (nbuf == 1) {
            __syncthreads();                          // consumers are done with the previous head
            if (issues_dma) stage_tile<HD, VT>(tile, vown + (size_t)h * HD, rstride, Nk, dma_wave, kDmaWaves);
        }
        if (issues_dma) __builtin_amdgcn_s_waitcnt(0);    // this wave's share of the head's tile has landed
        __syncthreads();
        if (issues_dma && nbuf == 2 && hh + 1 < heads_per)
            stage_tile<HD, VT>(tiles + (cur ^ 1) * tile_elems, vown + (size_t)(h + 1) * HD, rstride, Nk, dma_wave,
                               kDmaWaves);
        if (loader) continue;

        // ------------------------------------------------------------ consumers
        if constexpr (G == 16) {
            constexpr int LPP = 16 / P, CPN = 4 / LPP;
            const int STEP = NCONS * 4;
            const int row = lane >> 4, lr = lane & 15;
            const int ap = lr / LPP, asub = lr % LPP;      // phase A: point, corner subset
            const int ad = (D == 1) ? 0 : (ap % D);
            struct Sample {
                unsigned m;
                float lg;
                float2 of, u;
            };
            auto load_id = [&](int base) -> int {
                const int ia = base + row;
                return ia < end ? list[ia] : -1;
            };
            auto load_sample = [&](int n) -> Sample {
                Sample sm;
                const int nn = n < 0 ? 0 : n;
                const size_t qh = ((size_t)b * Nq + nn) * heads + h;
                sm.m = n < 0 ? 0u : (unsigned)vis[(size_t)b * Nq + nn];
                sm.lg = logits[qh * P + ap];
                sm.of = *reinterpret_cast<const float2*>(offs + (qh * P + ap) * 2);
                sm.u = *reinterpret_cast<const float2*>(uv + ((((size_t)b * Ncam + c) * Nq + nn) * D + ad) * 2);
                return sm;
            };
            // LDS byte addresses of this lane's two channel segments in row 0 of the current tile
            using M16 = ChMap<HD, 16>;
            const unsigned tile_lds = (unsigned)(uintptr_t)(lds_byte*)reinterpret_cast<const unsigned char*>(tile);
            const unsigned base_seg0 = tile_lds + M16::off0(lr) * (unsigned)sizeof(VT);
            const unsigned base_seg1 = tile_lds + M16::off1(lr) * (unsigned)sizeof(VT);
            const int base0 = start + wave * 4;
            int n_cur = load_id(base0);
            int n_nxt = load_id(base0 + STEP);
            Sample s_cur = load_sample(n_cur);

            for (int base = base0; base < end; base += STEP) {
                const Sample s_nxt = load_sample(n_nxt);
                const int n_nxt2 = load_id(base + 2 * STEP);
                // ---------------- phase A
                const unsigned m = s_cur.m;
                float wsel[2];
                unsigned ksel[2];                      // byte offsets of this lane's corner rows
                constexpr unsigned kRowBytes = HD * sizeof(VT);
                {
                    const float mx = group_max<16>(s_cur.lg);
                    const float e = __expf(s_cur.lg - mx);
                    const float ssum = group_sum<16>(asub == 0 ? e : 0.0f);
                    const float a = m ? e * __builtin_amdgcn_rcpf(ssum * (float)__popc(m)) : 0.0f;   // 1 ulp
                    Bilinear s;
                    bilinear_setup<false>(s_cur.u.x + s_cur.of.x * inv_w, s_cur.u.y + s_cur.of.y * inv_h, mh, mw, s);
                    if constexpr (CPN == 2) {
                        wsel[0] = a * (asub ? s.w[2] : s.w[0]);
                        wsel[1] = a * (asub ? s.w[3] : s.w[1]);
                        ksel[0] = (unsigned)(asub ? s.key[2] : s.key[0]) * kRowBytes;
                        ksel[1] = (unsigned)(asub ? s.key[3] : s.key[1]) * kRowBytes;
                    } else {
                        const float w01 = (asub & 1) ? s.w[1] : s.w[0], w23 = (asub & 1) ? s.w[3] : s.w[2];
                        const int k01 = (asub & 1) ? s.key[1] : s.key[0], k23 = (asub & 1) ? s.key[3] : s.key[2];
                        wsel[0] = a * ((asub & 2) ? w23 : w01);
                        wsel[1] = 0.0f;
                        ksel[0] = (unsigned)((asub & 2) ? k23 : k01) * kRowBytes;
                        ksel[1] = 0u;
                    }
                }
                // ---------------- phase B
                float acc[CPL];
#pragma unroll
                for (int j = 0; j < CPL; ++j) acc[j] = 0.0f;
                // the fused DPP adds are inline asm: the hazard recogniser does not see that they read
                // ksel through DPP, so keep two wait states behind the VALU ops that produced it
                PointLoop<HD, P, 0, VT>::run(base_seg0, base_seg1, wsel, ksel, acc);
                if (n_cur >= 0)
                    emit_row<HD, G>(slots + ((size_t)b * Nq + n_cur) * heads * HD + (size_t)h * HD, lr, acc,
                                    __popc(m) == 1);
                n_cur = n_nxt;
                n_nxt = n_nxt2;
                s_cur = s_nxt;
            }
        } else {
            // narrow heads (HD < 32, test sizes): plain per-lane path, G lanes per voxel
            constexpr int VPW = VER_WAVE / G;
            const int sub = lane / G, gl = lane % G;
            for (int i = start + wave * VPW + sub; i < end; i += NCONS * VPW) {
                const int nb = list[i];
                const unsigned mb = vis[(size_t)b * Nq + nb];
                const size_t qh = ((size_t)b * Nq + nb) * heads + h;
                float a[P], ox[P], oy[P];
                softmax_points<P>(logits + qh * P, a);
                const float* of = offs + qh * P * 2;
                const float icnt = 1.0f / (float)__popc(mb);
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    a[p] *= icnt;
                    ox[p] = of[2 * p] * inv_w;
                    oy[p] = of[2 * p + 1] * inv_h;
                }
                float acc[CPL];
#pragma unroll
                for (int j = 0; j < CPL; ++j) acc[j] = 0.0f;
                const float* u = uv + (((size_t)b * Ncam + c) * Nq + nb) * D * 2;
                gather_camera<HD, G, P, VT>(tile, (size_t)HD, u, D, ox, oy, a, mh, mw, gl, acc);
                emit_row<HD, G>(slots + ((size_t)b * Nq + nb) * heads * HD + (size_t)h * HD, gl, acc,
                                __popc(mb) == 1);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Forward, "corner slots" (P = 8 points, head_dim % 32 == 0; fp32 and bf16 tiles).  What round 1's kernels were
// bound by, measured (scratch/r02/, profiles/r02_*):
//   * the LDS pipe: per corner row and wave k_sca_fwd8 issued two ds_bpermute_b32 (9 LDS cycles each on gfx950) and
//     three ds_read_b64 whose four voxels per 32-lane service group sat on random tile rows (2.1-way bank conflicts on
//     average, ubench_lds.hip): 72 % LDS-busy, 43 % VALU-busy;
//   * s_waitcnt vmcnt(0) in front of every operand use, because stores / atomics / clamped loads sat under branches
//     and the compiler cannot count conditional vector-memory operations;
//   * one wave finishing a camera's shared-voxel list alone while the workgroup's other waves idled.
// This kernel:
//   * makes the four 8-lane slots of a half wave the FOUR BILINEAR CORNERS OF ONE SAMPLE and keeps the tile PLANAR in
//     LDS: plane i holds channels [32i, 32i+32) of every tile row, so a plane row is 128 B (fp32) or 64 B (bf16) and
//     the corner rows (r, r+1, r+W, r+W+1) of a 14-wide map fall on four disjoint bank sets -- every read of the
//     gather is conflict free by construction (rows that coincide at the map border broadcast).  The planes are too
//     far apart for the compiler to fuse a row's reads into half-rate ds_read2.  LDS-DMA writes lane-linear 16-byte
//     chunks, so the planar image only changes the per-lane SOURCE address;
//   * walks two voxels per wave at a time (one per half wave), one sampling point per step; the {corner weight, row
//     offset} records of a pair travel through a 512-B per-wave LDS table (four conflict-free ds_read_b128 per slot
//     and pair instead of two ds_bpermute per corner row), published one pair ahead;
//   * folds the four corner partial sums once per (voxel, head): v_permlane16_swap on pairs of accumulators (a
//     transposing reduction across the two 16-lane rows of a half wave) + one DPP row_ror:8 add; a lane then owns
//     2 x HD/32 adjacent channels of its voxel's output row;
//   * issues every global load and store unconditionally (clamped indices, dead lanes -> g_sca_dummy_row / zero adds),
//     so the compiler's vmcnt waits are exact and never drain the young output stores.
// Result on MI355X: LDS bank conflicts 29.7 M -> 2.6 M cycles per launch, LDS-busy 72 % -> 28 %, and the kernel is now
// bound by VALU issue (60 % busy: 12 FMAs + 12 bf16 unpacks per lane and corner row) -- see DESIGN.md section 3.1
// for why the launch time did not move.
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
template <typename T>
__device__ __forceinline__ T* lds_ptr_mut(unsigned addr) {
    return reinterpret_cast<T*>((unsigned char*)(__attribute__((address_space(3))) unsigned char*)(uintptr_t)addr);
}
__device__ __forceinline__ void permlane16_swap(float& a, float& b) {
    // rows (16 lanes) 1 and 3 of `a` <-> rows 0 and 2 of `b`.  (The ROCm 7.2 builtin returns the first
    // result twice; the s_nop covers the VALU-write -> permlane-read wait states the compiler cannot see.)
    asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}

// Rows that dead lanes (the pad entry of an odd list) write to, so that every store of the gather loop is
// unconditional: with stores or loads under a branch the compiler cannot count the vector-memory operations in
// flight and falls back to s_waitcnt vmcnt(0) in front of every operand use.
__device__ float g_sca_dummy_row[256 * 256];        // one 1-KiB slice per (blockIdx & 255): no chip-wide hot line

// ------------------------------------------------------------------------------------------
// Forward, corner-slot kernel with a merged work list, sample compaction and (optionally) streaming loader waves.
//
// A workgroup walks a contiguous range of (viewpoint, camera, chunk, head group) units, one (camera, head) tile
// at a time:
//   nload > 0  persistent form: the last `nload` waves only stream tiles (LDS-DMA of tile i+1 while the consumers
//              work on tile i; two tile buffers, one barrier per tile);
//   nload == 0 every wave stages its share of the tile, then all of them gather from it (one buffer; several
//              workgroups per CU overlap each other's staging).
// The consumer waves split a tile's voxel PAIRS into contiguous ranges over the merged list
// [voxels only this camera sees, padded to a whole pair | voxels shared with other cameras]: a pair is either all
// plain stores or all atomic adds (wave-uniform).  Voxel ids / visibility of a unit are fetched once for all its
// heads, and the first operands of a tile are requested before the tile barrier.
// Samples whose footprint misses the map entirely (or whose voxel is dead) have an all-zero record: phase A
// COMPACTS the live samples of a voxel to the front of its record row, and a pair only walks
// max(live samples of its two voxels) points -- with the reference's initial ring of offsets (1..8 px on a 14-px
// map, spatial_cross_attention.py:255-270) more than a third of the samples are outside.
// MATH (bf16 tiles only): how the gather multiplies a tile row by its weight.
//   0  the tile stays bf16 in LDS; every use unpacks to fp32 (v_and / v_lshlrev) and accumulates with v_pk_fma_f32
//   2  each wave converts the chunks it staged to fp16 IN PLACE (exact for bf16 values with |x| in [6.1e-5, 65504]:
//      8 mantissa bits fit in 11; RTZ saturates above and truncates into the subnormals below, abs. error < 6e-8),
//      weights travel as packed fp16 pairs, and the <= 8 points of a (voxel, head, corner) are accumulated with
//      v_pk_fma_f16 -- two channels per instruction, no unpack; the corner fold runs on the packed sums, everything
//      after it (camera sum, division) in fp32.  Error ~5e-4 of the partial sums (bf16 bound of the north star: 1e-2).
//   (1 = fp16 tile with fp32 accumulation: the compiler turns it into v_cvt_f32_f16 + v_pk_fma_f32, as many
//    instructions as mode 0 -- measured 3 % slower, 437 vs 424 us per launch; not dispatched.)
// HM (head-major value, head_major_views > 0): the tile is one contiguous block of HBM and is staged as it lies, ROW-major in
// LDS (a tile row = HD elements); the corner rows (r, r+1, r+14, r+15) of a 14-wide map of 192-byte rows still fall on four
// disjoint 64-byte bank groups (0, 192, 128, 64 mod 256; profiles/r02_ubench_lds.txt), so the gather stays conflict free.
// Division of the unit / tile indices by the launch's (run-time) small divisors.  A general integer division is ~25
// instructions and the prologue of a workgroup held fifteen of them (two in 64 bits) -- ~40 % of the launch's scalar
// instructions (35 M per 192-viewpoint launch in round 3's counters), executed by every wave in front of its first DMA request.
// The host passes floor(2^32 / d) + 1 per divisor: one s_mul_hi_u32, exact for n < 2^32 / d (checked by the launcher).
struct CsMagic {
    unsigned hsplit, nchunks, ncam, hper, ncons;
};
__host__ __device__ __forceinline__ unsigned cs_magic(int d) { return d > 1 ? (unsigned)(0x100000000ull / (unsigned)d) + 1u : 0u; }
__device__ __forceinline__ int cs_div(int n, int d, unsigned m) { return d == 1 ? n : (int)__umulhi((unsigned)n, m); }

// ONE: the launch shape that ships (one tile per workgroup, no loader waves, one voxel chunk) as compile-time constants --
// the unit / head loops, the loader path and their registers (63 spilled SGPRs in the general form) fold away.
#ifndef VER_CS_DMA_AUX
#define VER_CS_DMA_AUX 0        // cache policy bits of the reference-layout tile DMA (experiment hook; 2 = non-temporal: measured no better)
#endif
template <int HD, typename VT, int NKT, int MATH = 0, bool HM = false, bool ONE = false>
__global__ __launch_bounds__(1024) void k_sca_fwd_cs(
    const VT* __restrict__ value, const float* __restrict__ offs, const float* __restrict__ logits,
    const float* __restrict__ uv, const uint8_t* __restrict__ vis, const int* __restrict__ fwd_list,
    const int* __restrict__ fwd_cnt, float* slots, int Ncam, int Nq, int D, int heads, int mh, int mw,
    int nchunks_a, int chunk, int hsplit, int units_total, int units_per_wg_a, int nload_a, int head_major_views, int reverse,
    int heads_per_a, CsMagic mg) {
    const int nchunks = ONE ? 1 : nchunks_a, units_per_wg = ONE ? 1 : units_per_wg_a, nload = ONE ? 0 : nload_a;
    const int heads_per = ONE ? 1 : heads_per_a;
    // reverse: walk the units from the LAST viewpoint to the first.  The value tensor (347 MB at 192 viewpoints) and the
    // offsets / logits were written just before this launch, in ascending row order, through a 256-MB memory-side cache:
    // reading them back in the SAME order finds the oldest lines already evicted, reading in the opposite order starts with
    // the ones still resident.
    // head_major_views: 0 = value in the reference's layout [B, Ncam, Nk, heads, HD] (a tile row is HD elements inside a
    // heads*HD-wide token row); B > 0 = head-major [heads, B, Ncam, Nk, HD]: a (camera, head) tile is ONE contiguous block
    // of HBM -- every 1-KB LDS-DMA instruction then covers 8 whole 128-byte lines instead of ~11 partial ones, and no line
    // is shared with another head's workgroup, so the tile streams with the non-temporal policy (-20 us of 300 per launch)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int P = 8;
    constexpr bool F32 = sizeof(VT) == 4;
    constexpr int NV = HD / 32;
    constexpr unsigned RB = (HM ? HD : 32) * sizeof(VT);            // bytes between tile rows in LDS
    constexpr int CPR = 32 * sizeof(VT) / 16, EPC = 16 / sizeof(VT);
    static_assert(HD % 32 == 0, "8 lanes x vectors of 4 channels");
    static_assert(MATH == 0 || !F32, "the fp16 modes are for bf16 tiles");
    typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
    const int nwaves = (int)(blockDim.x >> 6);
    const int ncons = nwaves - nload;
    const int nbuf = nload > 0 ? 2 : 1;
    const int Nk = NKT ? NKT : mh * mw;
    const unsigned plane = HM ? 32u * (unsigned)sizeof(VT) : (unsigned)Nk * RB;      // bytes between a row's 32-channel groups
    const unsigned tile_bytes = ((unsigned)(NV * Nk * 32 * sizeof(VT)) + 15u) & ~15u;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t rstride = (size_t)heads * HD;
    // XCD-aware placement: workgroup b runs on XCD b % 8 and every XCD has its own L2, so the grid is dealt to the
    // XCDs in contiguous blocks -- the heads of one (viewpoint, camera) share its voxel lists, visibility, uv and the
    // 128-byte lines of the offsets / logits / value rows, and now meet them in ONE L2 instead of eight.
    const int wg_per_xcd = (int)(gridDim.x >> 3);                 // the grid is a multiple of 8
    const int wg = (int)(blockIdx.x & 7) * wg_per_xcd + (int)(blockIdx.x >> 3);
    const int u0 = wg * units_per_wg, u1 = min(u0 + units_per_wg, units_total);
    const int ntiles = (u1 - u0) * heads_per;
    if (ntiles <= 0) return;
    VER_TL(0);

    // Tile staging.  A lone loader wave runs its address arithmetic as one dependent chain, so the per-DMA work is a
    // handful of instructions: the lane's (plane, row) position advances incrementally and the tile's base pointer
    // is wave-uniform.  (Two integer divisions per DMA cost a loader wave 23 k cycles per 75-KB tile.)
    const int dma_wave = nload ? wave - ncons : wave, dma_waves = nload ? nload : nwaves;
    const int total_chunks = NV * Nk * CPR;
    const int rows_per_step = 64 * dma_waves / CPR;                // plane rows covered by one step of all DMA waves
    const int q_first = dma_wave * 64 + lane;
    const int jc = q_first % CPR;
    const int i_first = (q_first / CPR) / Nk, k_first = (q_first / CPR) - i_first * Nk;
    auto stage = [&](int i) {
        const int iu = cs_div(i, heads_per, mg.hper);
        int r = u0 + iu;
        if (reverse) r = units_total - 1 - r;
        const int rq = cs_div(r, hsplit, mg.hsplit), hs = r - rq * hsplit;
        r = cs_div(rq, nchunks, mg.nchunks);
        const int b = cs_div(r, Ncam, mg.ncam), c = r - b * Ncam;
        const int h = hs * heads_per + (i - iu * heads_per);
        VT* dst = reinterpret_cast<VT*>(smem + (nbuf == 2 ? (i & 1) : 0) * tile_bytes);
#ifdef VER_ABL_NODMA
        if (i >= 0) return;
#endif
        if constexpr (HM) {
            // chunk q of the tile is chunk q of the block: no address arithmetic beyond the lane offset
            const VT* src = value + ((((size_t)h * head_major_views + b) * Ncam + c) * Nk) * HD;
            for (int q0 = dma_wave * 64; q0 < total_chunks; q0 += 64 * dma_waves) {
                if (q0 + lane < total_chunks)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)(q0 + lane) * EPC),
                                                     (__attribute__((address_space(3))) void*)(dst + (size_t)q0 * EPC), 16, 0, 2);
            }
            return;
        }
        const VT* src = value + ((size_t)b * Ncam + c) * Nk * rstride + (size_t)h * HD + jc * EPC;
        int pi = i_first, pk = k_first;
        for (int q0 = dma_wave * 64; q0 < total_chunks; q0 += 64 * dma_waves) {
            if (q0 + lane < total_chunks) {
                const VT* g = src + (size_t)pk * rstride + pi * 32;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                                 (__attribute__((address_space(3))) void*)(dst + (size_t)q0 * EPC), 16, 0, VER_CS_DMA_AUX);
            }
            pk += rows_per_step;
            while (pk >= Nk) {
                pk -= Nk;
                pi += 1;
            }
        }
    };

    // bf16 -> fp16 in place, by the wave that staged the chunk (its own s_waitcnt covers the DMA): no extra barrier
    auto to_f16 = [&](int i) {
#ifdef VER_ABL_NOCONV
        return;
#endif
        if constexpr (MATH != 0) {
            const unsigned dst = (unsigned)(uintptr_t)(lds_byte*)reinterpret_cast<const unsigned char*>(smem) +
                                 (nbuf == 2 ? (unsigned)(i & 1) : 0u) * tile_bytes;
#ifdef VER_CS_CONV_SERIAL
            for (int q0 = dma_wave * 64; q0 < total_chunks; q0 += 64 * dma_waves) {
                if (q0 + lane < total_chunks) {
                    const unsigned a = dst + (unsigned)(q0 + lane) * 16u;
                    u32x4_t w = *lds_ptr<u32x4_t>(a);
                    u32x4_t o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const unsigned x = e == 0 ? w.x : e == 1 ? w.y : e == 2 ? w.z : w.w;
                        const auto hh = __builtin_amdgcn_cvt_pkrtz(__uint_as_float(x << 16), __uint_as_float(x & 0xffff0000u));
                        const unsigned r = __builtin_bit_cast(unsigned, hh);
                        if (e == 0) o.x = r; else if (e == 1) o.y = r; else if (e == 2) o.z = r; else o.w = r;
                    }
                    *lds_ptr_mut<u32x4_t>(a) = o;
                }
            }
#else
            // five chunks per lane in flight: the serial form (read, wait, convert, write per chunk) ran ten dependent LDS
            // round trips per tile -- ~10 k cycles of every workgroup's ~60 k-cycle prologue (scratch/r04/timeline_cs.py).
            // Reads past the end are clamped to the last chunk (which its owner may already have converted: the value is
            // dropped), only the store is predicated.
            constexpr int kBatch = 5;
            const int step = 64 * dma_waves;
            for (int q0 = dma_wave * 64 + lane; q0 < total_chunks; q0 += kBatch * step) {
                u32x4_t w[kBatch];
#pragma unroll
                for (int k = 0; k < kBatch; ++k)
                    w[k] = *lds_ptr<u32x4_t>(dst + (unsigned)min(q0 + k * step, total_chunks - 1) * 16u);
#pragma unroll
                for (int k = 0; k < kBatch; ++k) {
                    u32x4_t o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const unsigned x = e == 0 ? w[k].x : e == 1 ? w[k].y : e == 2 ? w[k].z : w[k].w;
                        const auto hh = __builtin_amdgcn_cvt_pkrtz(__uint_as_float(x << 16), __uint_as_float(x & 0xffff0000u));
                        const unsigned r = __builtin_bit_cast(unsigned, hh);
                        if (e == 0) o.x = r; else if (e == 1) o.y = r; else if (e == 2) o.z = r; else o.w = r;
                    }
                    if (q0 + k * step < total_chunks) *lds_ptr_mut<u32x4_t>(dst + (unsigned)(q0 + k * step) * 16u) = o;
                }
            }
#endif
        }
    };

    if (nload > 0 && wave >= ncons) {
        // ------------------------------------------------------------ loader waves: stream the tiles
        stage(0);
        for (int i = 0; i < ntiles; ++i) {
            if (i < 7) VER_TL(8 * i + 10);
            __builtin_amdgcn_s_waitcnt(0);            // this wave's share of tile i has landed
            if (i < 7) VER_TL(8 * i + 11);
            to_f16(i);
            if (i < 7) VER_TL(8 * i + 12);
            __syncthreads();                          // tile i complete; the consumers are done with tile i - 1
            if (i < 7) VER_TL(8 * i + 13);
            if (i + 1 < ntiles) stage(i + 1);
        }
        return;
    }

    // ---------------------------------------------------------------- consumer waves
    const int s4 = lane >> 4, v2 = (lane >> 3) & 1, l8 = lane & 7;   // phase A: pair of the iteration, voxel of the pair, point
    const int half = lane >> 5, rho = (lane >> 4) & 1, sigma = (lane >> 3) & 1;
    const int ad = (D == 1) ? 0 : (l8 % D);
    const float inv_w = 1.0f / (float)mw, inv_h = 1.0f / (float)mh;
    const unsigned smem_lds = (unsigned)(uintptr_t)(lds_byte*)reinterpret_cast<const unsigned char*>(smem);
    // records {weight, row offset} of a voxel pair: [voxel of the pair][corner][slot], 8 B each: a corner slot reads
    // its 8 sample slots as four conflict-free ds_read_b128 (corner rows are 64 B = 16 banks apart)
    const unsigned rec_wave = smem_lds + (unsigned)nbuf * tile_bytes + (unsigned)wave * 512u;
    const unsigned rec_wr0 = rec_wave + (unsigned)v2 * 256u;                                      // + slot * 8 + corner * 64
    const unsigned rec_rd = rec_wave + (unsigned)half * 256u + (unsigned)((lane >> 3) & 3) * 64u;
    const unsigned lane_off = (unsigned)l8 * (F32 ? 16u : 8u);
    struct Sample {
        unsigned m;
        float lg;
        float2 of, u;
    };
    int ti = 0;                                       // tile counter of this workgroup
#ifndef VER_CS_NO_DMA_FIRST
    // The first tile's LDS-DMA depends on nothing but the block index: it goes out before the counts, the voxel ids and
    // the first samples are fetched (three dependent memory round trips that used to run IN FRONT of it).
    if (nload == 0) stage(0);
#endif
#ifndef VER_CS_NO_PRIO
    // a new workgroup's waves are the youngest of their SIMDs and lose every issue arbitration against the older waves'
    // gather loops: the prologue (requests, tile conversion) runs at raised priority, the gather itself at the default
    __builtin_amdgcn_s_setprio(3);
#endif
    for (int un = u0; un < u1; ++un) {
        int r = reverse ? units_total - 1 - un : un;
        const int rq = cs_div(r, hsplit, mg.hsplit), hs = r - rq * hsplit;
        r = cs_div(rq, nchunks, mg.nchunks);
        const int ck = rq - r * nchunks;
        const int b = cs_div(r, Ncam, mg.ncam), c = r - b * Ncam;
        const int h0 = hs * heads_per;
        const int n_single = fwd_cnt[(b * Ncam + c) * 2], n_multi = fwd_cnt[(b * Ncam + c) * 2 + 1];
        const int w_start = ck * chunk;
        const int s_n = max(0, min(n_single - w_start, chunk)), m_n = max(0, min(n_multi - w_start, chunk));
        const int s_n2 = (s_n + 1) & ~1, s_pairs = s_n2 >> 1;
        const int TP = s_pairs + ((m_n + 1) >> 1);
        const int p_lo = cs_div(wave * TP, ncons, mg.ncons), p_hi = cs_div((wave + 1) * TP, ncons, mg.ncons);   // < 2^28: TP < 2^24
        const int iters = (p_hi - p_lo + 3) >> 2;
        const int* list = fwd_list + ((size_t)b * Ncam + c) * Nq;
        // list entry of this lane in wave iteration `it`; dead entries (pad, odd tail, pairs of other waves) return
        // -(id of a live voxel of the same region) - 1
        auto load_id = [&](int it) -> int {
            const int pp = min(p_lo + 4 * it + s4, TP - 1);
            const int e = 2 * pp + v2;
            const bool multi = e >= s_n2;
            const int idx = multi ? e - s_n2 : e;
            const int lim = multi ? m_n : s_n;
            const int pos = multi ? (Nq - 1 - w_start - min(idx, lim - 1)) : (w_start + min(idx, lim - 1));
            const int n = list[pos];
            return (idx < lim && p_lo + 4 * it + s4 < p_hi) ? n : -n - 1;
        };
        int un0 = -1, un1 = -1, un2 = -1;
        if (iters > 0) {
            un0 = load_id(0);
            un1 = load_id(1);
            un2 = load_id(2);
        }
        for (int hh = 0; hh < heads_per; ++hh, ++ti) {
            const int h = h0 + hh;
            if (nload == 0) {
#ifndef VER_CS_NO_DMA_FIRST
                if (ti) {
                    __syncthreads();                  // everyone is done with the previous tile
                    stage(ti);
                }
#else
                if (ti) __syncthreads();              // everyone is done with the previous tile
                stage(ti);
#endif
            }
            // per-tile WAVE-UNIFORM base pointers; a lane's address is base + 32-bit byte offset (one 24-bit multiply-add),
            // which the memory instructions take as SGPR base + VGPR offset -- no 64-bit VALU arithmetic per access
            const char* lg_base = reinterpret_cast<const char*>(logits + ((size_t)b * Nq * heads + h) * P);
            const char* of_base = reinterpret_cast<const char*>(offs + ((size_t)b * Nq * heads + h) * P * 2);
            const char* uv_base = reinterpret_cast<const char*>(uv + ((size_t)b * Ncam + c) * Nq * D * 2);
            const uint8_t* vis_base = vis + (size_t)b * Nq;
            char* out_base = reinterpret_cast<char*>(slots + ((size_t)b * Nq * heads + h) * HD);
            const unsigned hp4 = (unsigned)(heads * P * 4), hhd4 = (unsigned)(heads * HD * 4), d8 = (unsigned)(D * 8);
            auto load_sample = [&](int n) -> Sample {
                Sample sm;
                const unsigned nn = (unsigned)(n < 0 ? -n - 1 : n);          // < 2^24 voxels (checked by the host wrapper)
                sm.m = (unsigned)vis_base[nn];           // raw: a dead entry's mask is cleared where the record is used
                sm.lg = *reinterpret_cast<const float*>(lg_base + (__umul24(nn, hp4) + (unsigned)l8 * 4u));
                sm.of = *reinterpret_cast<const float2*>(of_base + (__umul24(nn, hp4) * 2u + (unsigned)l8 * 8u));
                sm.u = *reinterpret_cast<const float2*>(uv_base + (__umul24(nn, d8) + (unsigned)ad * 8u));
                return sm;
            };
            int n0 = un0, n1 = un1, n2 = un2;
            Sample s0 = {};
            if (iters > 0) s0 = load_sample(n0);      // requested before the tile barrier: the latency hides behind it
            VER_TL(1);
            if (nload == 0) {
                __builtin_amdgcn_s_waitcnt(0);        // this wave's share of the tile has landed
                VER_TL(2);
                to_f16(ti);
            }
            VER_TL(3);
            if (nload > 0 && ti < 7) VER_TL(8 * ti + 14);
            __syncthreads();                          // tile ti is complete
#ifndef VER_CS_NO_PRIO
            __builtin_amdgcn_s_setprio(0);
#endif
            VER_TL(4);
            if (nload > 0 && ti < 7) VER_TL(8 * ti + 15);
            const unsigned base4 = smem_lds + (nbuf == 2 ? (unsigned)(ti & 1) * tile_bytes : 0u) + lane_off;
#ifndef VER_CS_NO_PRESTORE_WAIT
            __builtin_amdgcn_s_waitcnt(0x0f70);       // the first iteration's operands (requested before the tile barrier)
#endif
            // operands: voxel ids two wave iterations ahead, sample records one ahead (all loads and stores of the loop
            // are unconditional, so the compiler's vmcnt waits never have to drain the young output stores)
            for (int it = 0; it < iters; ++it) {
                const Sample s1 = load_sample(n1);
                const int n3 = load_id(it + 3);
                // ---------------- phase A: lane = sampling point l8 of voxel (s4, v2)
                const unsigned m = n0 < 0 ? 0u : s0.m;
                float w[4];
                unsigned k[4];
                int cnt;                              // live samples of this lane's voxel
                unsigned rec_wr;
                {
                    const float mx = group_max<8>(s0.lg);
                    const float e = __expf(s0.lg - mx);
                    const float ssum = group_sum<8>(e);
                    const float a = m ? e * __builtin_amdgcn_rcpf(ssum * (float)__popc(m)) : 0.0f;
                    Bilinear s;
                    bilinear_setup<false>(s0.u.x + s0.of.x * inv_w, s0.u.y + s0.of.y * inv_h, mh, mw, s);
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        w[t] = a * s.w[t];
                        k[t] = (unsigned)s.key[t] * RB;            // row offset inside a plane
                    }
                    // compaction: live samples first (in point order), all-zero records behind them
                    const bool live = (w[0] + w[1] + w[2] + w[3]) != 0.0f;      // weights are >= 0
                    const unsigned long long bal = __ballot(live);
                    const unsigned gm = (unsigned)(bal >> (lane & ~7)) & 0xffu;
                    const unsigned below = (1u << l8) - 1u;
                    cnt = __popc(gm);
                    const int slot = live ? __popc(gm & below) : cnt + __popc(~gm & below);
                    rec_wr = rec_wr0 + (unsigned)slot * 8u;
                }
                // ---------------- phase B: two voxels per sub-iteration, lane = (voxel, corner, 4*NV channels)
                const int nsub = min(4, p_hi - p_lo - 4 * it);
                if (it == 1) VER_TL(6);
                // Records travel from the phase-A lanes to the corner slots through the wave's 512-B LDS table.  Other
                // lanes of the wave read them: LDS serves one wave's requests in order, so no wait is needed in hardware,
                // but the compiler must know (without the fence it forwarded the previous pair's loads to the lanes that
                // did not store -- legal for a single thread, wrong here).
                u32x4_t recs[P / 2];
                // the record {weight word, row offset}: the weight is fp32, or (w, w) as packed fp16 in mode 2
                typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
                u32x2_t rec[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if constexpr (MATH == 2) {
                        const _Float16 hw = (_Float16)w[t];
                        rec[t].x = __builtin_bit_cast(unsigned, h2_t{hw, hw});
                    } else {
                        rec[t].x = __float_as_uint(w[t]);
                    }
                    rec[t].y = k[t];
                }
                if (s4 == 0) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) *lds_ptr_mut<u32x2_t>(rec_wr + (unsigned)t * 64u) = rec[t];
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
                for (int pp = 0; pp < P / 2; ++pp) recs[pp] = *lds_ptr<u32x4_t>(rec_rd + (unsigned)pp * 16u);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (j > 0 && j >= nsub) break;                 // wave-uniform (an iteration exists for at least one pair)
#ifdef VER_ABL_NOATOMIC
                    const bool atomic = false;
#else
                    const bool atomic = p_lo + 4 * it + j >= s_pairs;   // wave-uniform: the pair is in the shared region
#endif
#if defined(VER_ABL_MEMONLY)
                    const int npts = min(0, max(__builtin_amdgcn_readlane(cnt, 16 * j), __builtin_amdgcn_readlane(cnt, 16 * j + 8)));
#elif defined(VER_ABL_NOPOINTS)
                    const int npts = min(1, max(__builtin_amdgcn_readlane(cnt, 16 * j), __builtin_amdgcn_readlane(cnt, 16 * j + 8)));
#else
                    const int npts = max(__builtin_amdgcn_readlane(cnt, 16 * j), __builtin_amdgcn_readlane(cnt, 16 * j + 8));
#endif
                    // `recs` holds this pair's records (requested during the previous pair's epilogue).  Publish the next
                    // pair's now -- LDS serves the wave's requests in order, so the reads of this pair's records are
                    // already done -- and request them after the FMA loop, when `recs` is free again: neither the
                    // write -> read round trip through LDS nor the read latency is exposed (85 us of a 474-us launch
                    // when they were).
                    if (j + 1 < nsub && s4 == j + 1) {
#pragma unroll
                        for (int t = 0; t < 4; ++t) *lds_ptr_mut<u32x2_t>(rec_wr + (unsigned)t * 64u) = rec[t];
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    float acc[4 * NV];
                    h2_t acc2[2 * NV];                             // mode 2: the same 4 * NV channels as packed fp16 pairs
#pragma unroll
                    for (int i = 0; i < 4 * NV; ++i) acc[i] = 0.0f;
#pragma unroll
                    for (int i = 0; i < 2 * NV; ++i) acc2[i] = h2_t{(_Float16)0.0f, (_Float16)0.0f};
                    // tile vectors of a point are requested one point ahead of their FMAs (explicitly: left to itself the
                    // compiler, short of registers, funnelled every read through one register pair with a full
                    // lgkmcnt(0) wait in front of each group of FMAs)
                    typedef typename std::conditional<F32, f32x4_t, u32x2_t>::type tv_t;
                    auto rec_wbits = [&](int pt) { return (pt & 1) ? recs[pt / 2].z : recs[pt / 2].x; };
                    auto rec_a = [&](int pt) { return base4 + ((pt & 1) ? recs[pt / 2].w : recs[pt / 2].y); };
                    // (volatile: an ordinary load whose only use is in the next point's block gets sunk into it)
                    auto tile_ld = [&](unsigned addr) -> tv_t {
                        return *(const volatile __attribute__((address_space(3))) tv_t*)(uintptr_t)addr;
                    };
                    tv_t tb[2][NV];                                // two buffers by point parity: no register copies
#pragma unroll
                    for (int i = 0; i < NV; ++i) tb[0][i] = tile_ld(rec_a(0) + (unsigned)i * plane);
#pragma unroll
                    for (int pt = 0; pt < P; ++pt) {
                        if (pt >= npts) break;                     // wave-uniform: the pair has no more live samples
                        const unsigned wbit = rec_wbits(pt);
                        const float wb = __uint_as_float(wbit);
                        if (pt + 1 < P) {
#pragma unroll
                            for (int i = 0; i < NV; ++i) tb[(pt + 1) & 1][i] = tile_ld(rec_a(pt + 1) + (unsigned)i * plane);
                        }
#pragma unroll
                        for (int i = 0; i < NV; ++i) {
                            const tv_t tc = tb[pt & 1][i];
                            if constexpr (MATH == 2) {
                                const h2_t w2 = __builtin_bit_cast(h2_t, wbit);
                                acc2[2 * i] = __builtin_elementwise_fma(__builtin_bit_cast(h2_t, (unsigned)tc.x), w2, acc2[2 * i]);
                                acc2[2 * i + 1] = __builtin_elementwise_fma(__builtin_bit_cast(h2_t, (unsigned)tc.y), w2, acc2[2 * i + 1]);
                            } else {
                                float v[4];
                                if constexpr (F32) {
                                    v[0] = tc.x; v[1] = tc.y; v[2] = tc.z; v[3] = tc.w;
                                } else if constexpr (MATH == 1) {
                                    const h2_t a = __builtin_bit_cast(h2_t, (unsigned)tc.x), b2 = __builtin_bit_cast(h2_t, (unsigned)tc.y);
                                    v[0] = (float)a.x; v[1] = (float)a.y; v[2] = (float)b2.x; v[3] = (float)b2.y;
                                } else {
                                    v[0] = __uint_as_float(tc.x << 16); v[1] = __uint_as_float(tc.x & 0xffff0000u);
                                    v[2] = __uint_as_float(tc.y << 16); v[3] = __uint_as_float(tc.y & 0xffff0000u);
                                }
#pragma unroll
                                for (int q = 0; q < 4; ++q) acc[i * 4 + q] = __builtin_fmaf(wb, v[q], acc[i * 4 + q]);
                            }
                        }
                    }
                    if (it == 1 && j == 0) VER_TL(7);
                    if (j + 1 < nsub) {                            // next pair's records: the latency hides behind the epilogue
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
                        for (int pp = 0; pp < P / 2; ++pp) recs[pp] = *lds_ptr<u32x4_t>(rec_rd + (unsigned)pp * 16u);
                    }
                    // fold the four corner slots: rows of 16 lanes first (transposing: even rows keep channels {0,1}
                    // of each vector, odd rows {2,3}), then the two slots of a row
                    float out[2 * NV];
                    if constexpr (MATH == 2) {
                        // the same fold on the packed pairs: (ch0, ch1) and (ch2, ch3) of a vector are one register each
#pragma unroll
                        for (int i = 0; i < NV; ++i) {
                            float a = __builtin_bit_cast(float, acc2[2 * i]), b2 = __builtin_bit_cast(float, acc2[2 * i + 1]);
                            permlane16_swap(a, b2);
                            h2_t sum = __builtin_bit_cast(h2_t, a) + __builtin_bit_cast(h2_t, b2);
                            sum = sum + __builtin_bit_cast(h2_t, dpp_mov<0x128>(__builtin_bit_cast(float, sum)));      // row_ror:8
                            out[2 * i + 0] = (float)sum.x;
                            out[2 * i + 1] = (float)sum.y;
                        }
                    } else {
#pragma unroll
                        for (int i = 0; i < NV; ++i) {
                            permlane16_swap(acc[i * 4 + 0], acc[i * 4 + 2]);
                            permlane16_swap(acc[i * 4 + 1], acc[i * 4 + 3]);
                            out[2 * i + 0] = acc[i * 4 + 0] + acc[i * 4 + 2];
                            out[2 * i + 1] = acc[i * 4 + 1] + acc[i * 4 + 3];
                        }
#pragma unroll
                        for (int i = 0; i < 2 * NV; ++i) out[i] += dpp_mov<0x128>(out[i]);      // row_ror:8
                    }
#ifndef VER_CS_NO_PRESTORE_WAIT
                    // gfx950 counts loads and stores in ONE counter, and the number of stores of an iteration is not a
                    // compile-time constant: the compiler's wait for the next iteration's operands (requested at the top of
                    // this one) is a vmcnt(0) at the top of the next -- right behind this iteration's last store.  Waiting
                    // HERE, in front of the iteration's first store, covers the same requests a phase A and a pair later.
                    if (j == 0) __builtin_amdgcn_s_waitcnt(0x0f70);       // vmcnt(0) only
#endif
                    const int na = __builtin_amdgcn_readlane(n0, 16 * j), nb = __builtin_amdgcn_readlane(n0, 16 * j + 8);
                    const int n = half ? nb : na;
                    const unsigned rowoff = __umul24((unsigned)(n < 0 ? -n - 1 : n), hhd4) + (unsigned)(l8 * 4 + rho * 2) * 4u;
                    float* live_row = reinterpret_cast<float*>(out_base + rowoff);
                    // the only dead voxel of the plain-store region is the pad entry of an odd single count: last pair
                    const bool has_pad = !atomic && (s_n & 1) && p_lo + 4 * it + j == s_pairs - 1;     // wave-uniform
#ifdef VER_ABL_NOSTORE
                    if (out[0] == 12345.678f)
#endif
                    if (!atomic) {
                        // both slots of a row hold the sums and share the stores
                        float* row = live_row;
#ifdef VER_ABL_DUMMYOUT
                        row = g_sca_dummy_row + (blockIdx.x & 255) * 256 + l8 * 4 + rho * 2;      // timing: no HBM write stream
#else
                        if (has_pad && n < 0) row = g_sca_dummy_row + (blockIdx.x & 255) * 256 + l8 * 4 + rho * 2;
#endif
#pragma unroll
                        for (int kk = 0; kk < (NV + 1) / 2; ++kk) {
                            float o2[2];
                            float* dst;
                            if (2 * kk + 1 < NV) {
                                float a0 = out[4 * kk], a1 = out[4 * kk + 1], b0 = out[4 * kk + 2], b1 = out[4 * kk + 3];
                                asm("" : "+v"(b0), "+v"(b1));      // (keeps the selects out of scratch memory)
                                o2[0] = sigma ? b0 : a0;
                                o2[1] = sigma ? b1 : a1;
                                dst = row + (2 * kk + sigma) * 32;
                            } else {
                                // odd plane count: both slots of a row hold the last vector's two sums -- each writes ONE
                                // of them (4-byte stores, 32 lanes = one 128-byte line) instead of both writing both
                                float a0 = out[4 * kk], a1 = out[4 * kk + 1];
                                asm("" : "+v"(a1));
                                const float one = sigma ? a1 : a0;
                                __builtin_nontemporal_store(one, row + 2 * kk * 32 + sigma);
                                continue;
                            }
                            store_vec<2, true>(dst, o2);
                        }
                    } else {
                        float* row = live_row;                     // dead voxel (odd tail): adds zeros to a live shared row
#pragma unroll
                        for (int i = 0; i < NV; ++i) {
                            float a0 = out[2 * i], b0 = out[2 * i + 1];
                            asm("" : "+v"(b0));
                            const float val = n < 0 ? 0.0f : (sigma ? b0 : a0);
                            atomicAdd(row + i * 32 + sigma, val);
                        }
                    }
                }
                n0 = n1; n1 = n2; n2 = n3;
                s0 = s1;
                VER_TL(8 + it);
            }
            VER_TL(5);
        }
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
    if (start < cnt) stage_tile<HD, VT>(tile, value + tbase, rstride, Nk, threadIdx.x >> 6, 8);
    for (int i = threadIdx.x; i < Nk * HD / 4; i += 512)
        reinterpret_cast<float4*>(gtile)[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    stage_wait();

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
// Backward, split in two atomic-free (in LDS) kernels for head_dim >= 32 (16-lane groups):
//
//  k_sca_bwd_off  d(offsets), d(logits): the forward kernel's structure (double-buffered LDS-DMA
//     tiles over heads, phase A / phase B, DPP broadcast).  Phase B forms the per-corner dots
//     <g, V[row]> (16-lane DPP butterflies); the phase-A lane that owns the corner picks its dot
//     up and combines it with its own bilinear weight / slope; softmax backward is a second
//     butterfly over the row.  Rows of voxels seen by one camera are plain stores, others fp32
//     atomics on the zeroed buffers.
//  k_sca_bwd_val  d(value): does not need the value tile at all.  Per (viewpoint, camera, head) and
//     sub-chunk of 160 visible voxels: grad rows -> LDS, every (voxel, point, corner) becomes an
//     event {tile row, softmax*corner weight/#cams}; the events are counting-sorted by tile row with
//     integer LDS atomics, and each 16-lane group then accumulates ITS tile rows in registers
//     (4 rows x HD/16 channels per lane) from the LDS-resident grad rows.  One plain store per
//     tile row at the end; no floating-point atomics (the retired kernel spent 52 % of its wave
//     cycles waiting on ds_add_f32, which retires ~0.5 lane/clk/CU).
constexpr int kValSub = 160;           // voxels per sub-chunk of k_sca_bwd_val
constexpr int kValThreads = 1024;      // (512 threads x 64..96 voxels, two workgroups per CU: same time)
constexpr int kValGroups = kValThreads / 16;   // 16-lane groups, each owns tile rows g, g + kValGroups, ...
constexpr int kValMaxRows = 256;       // tile rows (map_h*map_w) the register accumulators cover

template <int HD, int P, typename VT>
__global__ __launch_bounds__(kFwdThreads) void k_sca_bwd_off(
    const VT* __restrict__ value, const float* __restrict__ offs, const float* __restrict__ logits,
    const float* __restrict__ uv, const uint8_t* __restrict__ vis, const int* __restrict__ vis_list,
    const int* __restrict__ vis_cnt, const float* __restrict__ gslots, float* goffs, float* glogits, int Ncam,
    int Nq, int D, int heads, int mh, int mw, int nchunks, int chunk, int hsplit, int nbuf) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int CPL = HD / 16;
    constexpr int LPP = 16 / P, CPN = 4 / LPP;
    const int nwaves_rt = (int)(blockDim.x >> 6);
    const int STEP = nwaves_rt * 4;
    using M16 = ChMap<HD, 16>;
    const int Nk = mh * mw;
    const size_t tile_elems = (size_t)Nk * HD;
    VT* tiles = reinterpret_cast<VT*>(smem);
    int bid = blockIdx.x;
    const int ck = bid % nchunks;
    bid /= nchunks;
    const int hs = bid % hsplit;
    bid /= hsplit;
    const int c = bid % Ncam;
    const int b = bid / Ncam;
    const int cnt = vis_cnt[b * Ncam + c];
    const int start = ck * chunk;
    if (start >= cnt) return;
    const int end = min(cnt, start + chunk);
    const int heads_per = heads / hsplit, h0 = hs * heads_per;
    const size_t rstride = (size_t)heads * HD;
    const VT* vown = value + ((size_t)b * Ncam + c) * Nk * rstride;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row = lane >> 4, lr = lane & 15;
    const int ap = lr / LPP, asub = lr % LPP;
    const int ad = (D == 1) ? 0 : (ap % D);
    const int* list = vis_list + ((size_t)b * Ncam + c) * Nq;
    const float inv_w = 1.0f / (float)mw, inv_h = 1.0f / (float)mh;
    constexpr unsigned kRowBytes = HD * sizeof(VT);

    if (nbuf == 2) stage_tile<HD, VT>(tiles, vown + (size_t)h0 * HD, rstride, Nk, wave, nwaves_rt);
    for (int hh = 0; hh < heads_per; ++hh) {
        const int h = h0 + hh;
        const int cur = nbuf == 2 ? (hh & 1) : 0;
        VT* tile = tiles + cur * tile_elems;
        if (nbuf == 1) {
            __syncthreads();
            stage_tile<HD, VT>(tile, vown + (size_t)h * HD, rstride, Nk, wave, nwaves_rt);
        }
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        if (nbuf == 2 && hh + 1 < heads_per)
            stage_tile<HD, VT>(tiles + (cur ^ 1) * tile_elems, vown + (size_t)(h + 1) * HD, rstride, Nk, wave,
                               nwaves_rt);
        const unsigned char* tile0 = reinterpret_cast<const unsigned char*>(tile + M16::off0(lr));
        const unsigned char* tile1 = reinterpret_cast<const unsigned char*>(tile + M16::off1(lr));

        // operands are software pipelined like the forward kernel's: voxel ids two iterations ahead, the
        // sample record (logit, offset, uv, grad row) one ahead -- otherwise every iteration starts with
        // two dependent global-memory latencies (the kernel was bound by exactly that)
        struct Sample {
            unsigned m;
            float lg;
            float2 of, u;
            float g[CPL];
        };
        auto load_id = [&](int base) -> int {
            const int ia = base + row;
            return ia < end ? list[ia] : -1;
        };
        auto load_sample = [&](int n) -> Sample {
            Sample sm;
            const int nn = n < 0 ? 0 : n;
            const size_t qh = ((size_t)b * Nq + nn) * heads + h;
            sm.m = n < 0 ? 0u : (unsigned)vis[(size_t)b * Nq + nn];
            sm.lg = logits[qh * P + ap];
            sm.of = *reinterpret_cast<const float2*>(offs + (qh * P + ap) * 2);
            sm.u = *reinterpret_cast<const float2*>(uv + ((((size_t)b * Ncam + c) * Nq + nn) * D + ad) * 2);
            load_ch<HD, 16, float>(gslots + ((size_t)b * Nq + nn) * heads * HD + (size_t)h * HD, lr, sm.g);
            return sm;
        };
        const int base0 = start + wave * 4;
        int n_cur = load_id(base0);
        int n_nxt = load_id(base0 + STEP);
        Sample s_cur = load_sample(n_cur);
        for (int base = base0; base < end; base += STEP) {
            const Sample s_nxt = load_sample(n_nxt);
            const int n_nxt2 = load_id(base + 2 * STEP);
            const bool alive = n_cur >= 0;
            const int n = alive ? n_cur : 0;
            const unsigned m = s_cur.m;
            const size_t qh = ((size_t)b * Nq + n) * heads + h;
            const float lg = s_cur.lg;
            const float2 of = s_cur.of;
            const float2 u = s_cur.u;
            float g[CPL];
            const float icnt = m ? 1.0f / (float)__popc(m) : 0.0f;
#pragma unroll
            for (int j = 0; j < CPL; ++j) g[j] = s_cur.g[j] * icnt;
            // ---------------- phase A
            const float mx = group_max<16>(lg);
            const float e = __expf(lg - mx);
            const float a = e / group_sum<16>(asub == 0 ? e : 0.0f);
            Bilinear s;
            bilinear_setup<true>(u.x + of.x * inv_w, u.y + of.y * inv_h, mh, mw, s);
            float wsel[2], gxs[2], gys[2];
            unsigned ksel[2];
            if constexpr (CPN == 2) {
                // explicit selects: an index like s.w[2 * asub + r] sends the whole struct to scratch memory
                wsel[0] = asub ? s.w[2] : s.w[0];
                wsel[1] = asub ? s.w[3] : s.w[1];
                gxs[0] = asub ? s.gx[2] : s.gx[0];
                gxs[1] = asub ? s.gx[3] : s.gx[1];
                gys[0] = asub ? s.gy[2] : s.gy[0];
                gys[1] = asub ? s.gy[3] : s.gy[1];
                ksel[0] = (unsigned)(asub ? s.key[2] : s.key[0]) * kRowBytes;
                ksel[1] = (unsigned)(asub ? s.key[3] : s.key[1]) * kRowBytes;
            } else {
                const int q = asub & 3;
                wsel[0] = q == 0 ? s.w[0] : q == 1 ? s.w[1] : q == 2 ? s.w[2] : s.w[3];
                gxs[0] = q == 0 ? s.gx[0] : q == 1 ? s.gx[1] : q == 2 ? s.gx[2] : s.gx[3];
                gys[0] = q == 0 ? s.gy[0] : q == 1 ? s.gy[1] : q == 2 ? s.gy[2] : s.gy[3];
                ksel[0] = (unsigned)(q == 0 ? s.key[0] : q == 1 ? s.key[1] : q == 2 ? s.key[2] : s.key[3]) * kRowBytes;
                wsel[1] = gxs[1] = gys[1] = 0.0f;
                ksel[1] = 0u;
            }
            // ---------------- phase B: dots of the grad row with the four corner rows of every point
            float dsel[2] = {0.0f, 0.0f};
            auto one_point = [&](auto pt_c) {
                constexpr int PT = decltype(pt_c)::value;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int reg = t % CPN;
                    unsigned koff;
                    if (t / CPN == 0) koff = row_bcast_u<PT * LPP + 0>(ksel[reg]);
                    else if (t / CPN == 1) koff = row_bcast_u<PT * LPP + (LPP > 1 ? 1 : 0)>(ksel[reg]);
                    else if (t / CPN == 2) koff = row_bcast_u<PT * LPP + (LPP > 2 ? 2 : 0)>(ksel[reg]);
                    else koff = row_bcast_u<PT * LPP + (LPP > 3 ? 3 : 0)>(ksel[reg]);
                    float v[CPL];
                    load_vec<M16::W0>(reinterpret_cast<const VT*>(tile0 + koff), v);
                    load_vec<M16::W1>(reinterpret_cast<const VT*>(tile1 + koff), v + M16::W0);
                    float d = 0.0f;
#pragma unroll
                    for (int j = 0; j < CPL; ++j) d += g[j] * v[j];
                    d = group_sum<16>(d);
                    if (ap == PT && asub == t / CPN) dsel[reg] = d;
                }
            };
            one_point(std::integral_constant<int, 0>());
            one_point(std::integral_constant<int, 1>());
            one_point(std::integral_constant<int, 2>());
            one_point(std::integral_constant<int, 3>());
            if constexpr (P == 8) {
                one_point(std::integral_constant<int, 4>());
                one_point(std::integral_constant<int, 5>());
                one_point(std::integral_constant<int, 6>());
                one_point(std::integral_constant<int, 7>());
            }
            // ---------------- back in the phase-A role: combine the point's corners, softmax backward
            float sa = wsel[0] * dsel[0] + wsel[1] * dsel[1];
            float sx = gxs[0] * dsel[0] + gxs[1] * dsel[1];
            float sy = gys[0] * dsel[0] + gys[1] * dsel[1];
            sa = group_sum<LPP>(sa);
            sx = group_sum<LPP>(sx);
            sy = group_sum<LPP>(sy);
            const float dot = group_sum<16>(asub == 0 ? a * sa : 0.0f);
            if (alive && m && asub == 0) {
                float* go = goffs + (qh * P + ap) * 2;
                float* gw = glogits + qh * P + ap;
                const float gx = a * sx, gy = a * sy, gl = a * (sa - dot);   // d x_pix / d offset = 1
                if (!(m & (m - 1))) {
                    *reinterpret_cast<float2*>(go) = make_float2(gx, gy);
                    *gw = gl;
                } else {
                    atomicAdd(go, gx);
                    atomicAdd(go + 1, gy);
                    atomicAdd(gw, gl);
                }
            }
            n_cur = n_nxt;
            n_nxt = n_nxt2;
            s_cur = s_nxt;
        }
    }
}

template <int HD, int P>
__global__ __launch_bounds__(kValThreads) void k_sca_bwd_val(
    const float* __restrict__ offs, const float* __restrict__ logits, const float* __restrict__ uv,
    const uint8_t* __restrict__ vis, const int* __restrict__ vis_list, const int* __restrict__ vis_cnt,
    const float* __restrict__ gslots, float* gvalue, int Ncam, int Nq, int D, int heads, int mh, int mw,
    int nchunks, int chunk) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int CPL = HD / 16;
    constexpr int NEV = kValSub * P * 4;               // events per sub-chunk
    constexpr int NR = kValMaxRows / kValGroups;               // tile rows per lane group
    float* G = reinterpret_cast<float*>(smem);                                   // [kValSub][HD]
    float* ev_coef = G + kValSub * HD;                                           // [NEV]
    unsigned short* ev_key = reinterpret_cast<unsigned short*>(ev_coef + NEV);   // [NEV]
    unsigned short* sorted = ev_key + NEV;                                       // [NEV]
    int* rcnt = reinterpret_cast<int*>(sorted + NEV);                            // [kValMaxRows]
    int* rstart = rcnt + kValMaxRows;                                            // [kValMaxRows]
    int* rcur = rstart + kValMaxRows;                                            // [kValMaxRows]
    const int Nk = mh * mw;
    int bid = blockIdx.x;
    const int ck = bid % nchunks;
    bid /= nchunks;
    const int h = bid % heads;
    bid /= heads;
    const int c = bid % Ncam;
    const int b = bid / Ncam;
    const int cnt = vis_cnt[b * Ncam + c];
    const int start = ck * chunk;
    const bool atomic_flush = nchunks > 1;             // gvalue pre-zeroed by the host wrapper then
    if (start >= cnt && (atomic_flush || ck != 0)) return;
    const int end = min(cnt, start + chunk);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int gidx = tid >> 4, lr = tid & 15;           // 64 lane groups of 16
    const int* list = vis_list + ((size_t)b * Ncam + c) * Nq;
    const float inv_w = 1.0f / (float)mw, inv_h = 1.0f / (float)mh;
    float acc[NR][CPL];
#pragma unroll
    for (int j = 0; j < NR; ++j)
#pragma unroll
        for (int i = 0; i < CPL; ++i) acc[j][i] = 0.0f;

    for (int s0 = start; s0 < end; s0 += kValSub) {
        const int nsub = min(kValSub, end - s0);
        for (int i = tid; i < kValMaxRows; i += kValThreads) rcnt[i] = 0;
        // grad rows of the sub-chunk's voxels, scaled by 1/#cams
        for (int v = gidx; v < nsub; v += kValThreads / 16) {
            const int n = list[s0 + v];
            const float icnt = 1.0f / (float)__popc((unsigned)vis[(size_t)b * Nq + n]);
            float g[CPL];
            load_ch<HD, 16, float>(gslots + ((size_t)b * Nq + n) * heads * HD + (size_t)h * HD, lr, g);
#pragma unroll
            for (int j = 0; j < CPL; ++j) g[j] *= icnt;
            store_ch<HD, 16>(G + v * HD, lr, g);
        }
        __syncthreads();
        // events: one lane per (voxel, point); groups of P lanes share a voxel (softmax butterfly)
        for (int q0 = wave * 64; q0 < nsub * P; q0 += kValThreads) {
            const int q = q0 + lane;
            const bool live = q < nsub * P;
            const int v = live ? q / P : 0, p = q % P;
            const int n = list[s0 + v];
            const size_t qh = ((size_t)b * Nq + n) * heads + h;
            const float lg = logits[qh * P + p];
            const float2 of = *reinterpret_cast<const float2*>(offs + (qh * P + p) * 2);
            const int d = (D == 1) ? 0 : (p % D);
            const float2 u = *reinterpret_cast<const float2*>(uv + ((((size_t)b * Ncam + c) * Nq + n) * D + d) * 2);
            const float mx = group_max<P>(lg);
            const float e = __expf(lg - mx);
            const float a = e / group_sum<P>(e);
            Bilinear s;
            bilinear_setup<false>(u.x + of.x * inv_w, u.y + of.y * inv_h, mh, mw, s);
            if (live) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float coef = a * s.w[t];
                    ev_coef[q * 4 + t] = coef;
                    ev_key[q * 4 + t] = (unsigned short)s.key[t];
                    if (coef != 0.0f) atomicAdd(&rcnt[s.key[t]], 1);
                }
            }
        }
        __syncthreads();
        if (wave == 0) {                                 // exclusive scan over the (<= 256) tile rows
            int loc[4], run = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                loc[i] = run;
                run += rcnt[lane * 4 + i];
            }
            int inc = run;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int t = __shfl_up(inc, o, VER_WAVE);
                if (lane >= o) inc += t;
            }
            const int base = inc - run;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                rstart[lane * 4 + i] = base + loc[i];
                rcur[lane * 4 + i] = base + loc[i];
            }
        }
        __syncthreads();
        for (int e = tid; e < nsub * P * 4; e += kValThreads)
            if (ev_coef[e] != 0.0f) sorted[atomicAdd(&rcur[ev_key[e]], 1)] = (unsigned short)e;
        __syncthreads();
        // every lane group accumulates its tile rows from the LDS-resident grad rows
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            const int r = gidx + kValGroups * j;
            if (r < Nk) {
                const int i0 = rstart[r], i1 = i0 + rcnt[r];
                for (int i = i0; i < i1; ++i) {
                    const int e = sorted[i];
                    const float coef = ev_coef[e];
                    float g[CPL];
                    load_ch<HD, 16, float>(G + (e / (P * 4)) * HD, lr, g);
#pragma unroll
                    for (int k = 0; k < CPL; ++k) acc[j][k] += coef * g[k];
                }
            }
        }
        __syncthreads();
    }
    const size_t rstride = (size_t)heads * HD;
    float* gv = gvalue + ((size_t)b * Ncam + c) * Nk * rstride + (size_t)h * HD;
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        const int r = gidx + kValGroups * j;
        if (r < Nk) {
            if (!atomic_flush) store_ch<HD, 16>(gv + (size_t)r * rstride, lr, acc[j]);
            else atomic_add_ch<HD, 16>(gv + (size_t)r * rstride, lr, 1.0f, acc[j]);
        }
    }
}

// ------------------------------------------------------------------------------------------
// Backward on the matrix cores (bf16 value tiles, 8 points, head_dim % 32 == 0): d(offsets), d(logits) AND d(value)
// in one kernel.  Unlike the forward, the backward has dense structure to give the MFMA units:
//   * every d(offset) / d(logit) is a combination of dots <g[v], V[k]> between a voxel's grad row and a tile row.
//     D^T = V x G^T for ALL (tile row k, voxel v) pairs of a 32-voxel chunk is 13 x 2 x 3 MFMA tiles x 2 (G split
//     into bf16 hi + lo); a sampling point then just picks its four dots.  6.5x more dots than needed, and still
//     a fraction of what the per-corner dot products cost on the VALU (12 FMAs + 12 unpacks + a 16-lane reduction
//     per corner row);
//   * d(value) = S^T x G is a true reduction over voxels (S[v][k] = softmax weight x bilinear weight of every
//     sample of voxel v that touches row k): the 4 x 8 x 32 events of a chunk are added into a dense fp32
//     S^T[k][v] in LDS (ds_add_f32 resolves two samples on one row), which then feeds 13 x 6 MFMA tiles x 3
//     (S and G each split hi + lo, lo x lo dropped: 16 mantissa bits in both factors).
// One workgroup (4 waves) per (viewpoint, camera, head): the tile is staged once by LDS-DMA, the camera's voxels
// are walked in chunks of 32 in the forward's work order (only-this-camera voxels padded to a multiple of 8, then the
// shared ones: a wave's 8 voxels are all plain stores or all atomics), a thread = (voxel, sampling point) in the
// scalar phases.  G is read once, nothing is sorted, and there are no floating-point atomics on global memory
// except for voxels seen by several cameras.
typedef __bf16 mm_bf16x8 __attribute__((ext_vector_type(8)));
typedef short mm_s16x4 __attribute__((ext_vector_type(4)));
typedef short mm_s16x8 __attribute__((ext_vector_type(8)));
#ifndef VER_MM_WAVES
#define VER_MM_WAVES 4
#endif
// waves per workgroup of k_sca_bwd_mm.  4 (two workgroups per CU, 246 VGPRs) measured 591 us per 192-viewpoint launch;
// 8 (one workgroup per CU at 179 VGPRs) 771 us; 8 squeezed into 128 VGPRs for two workgroups per CU spills: 1759 us.
constexpr int kMmWaves = VER_MM_WAVES;
constexpr int kMmDss = 36;            // floats per row of the D / S^T buffer [tile row][32 voxels + pad]: rows 4 banks apart

__device__ __forceinline__ void mm_split(const float (&f)[8], mm_bf16x8& hi, mm_bf16x8& lo) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const __bf16 h = (__bf16)f[j];
        hi[j] = h;
        lo[j] = (__bf16)(f[j] - (float)h);
    }
}

// GVT: d(value) is written as float or as bf16 (uint16_t).  NW waves per workgroup: the first four carry the scalar phases
// (256 threads = 32 voxels x 8 points), all NW share the matrix phases, the zero fill and the tile staging.
// GST: dtype of grad_slots.  float: the rows are split into bf16 hi + lo (16 mantissa bits).  uint16_t (bf16, what the
// output_proj GEMM hands back under bf16 autocast -- VER_SCA_GRAD_SLOTS_BF16): the rows are used as they are, ONE bf16 term
// (exact except for the 1/3, 1/5, ... camera-count factors of voxels seen by 3+ cameras, which round to bf16): half the bytes
// of the largest operand, no lo fragments, a third fewer MFMAs.
template <int HD, int NKT, typename GVT, int NW, typename GST = float>
__global__ __launch_bounds__(NW * 64, 2) void k_sca_bwd_mm(
    const uint16_t* __restrict__ value, const float* __restrict__ offs, const float* __restrict__ logits,
    const float* __restrict__ uv, const uint8_t* __restrict__ vis, const int* __restrict__ fwd_list,
    const int* __restrict__ fwd_cnt, const GST* __restrict__ gslots, GVT* __restrict__ gvalue, float* goffs,
    float* glogits, int Ncam, int Nq, int D, int heads, int mh, int mw, int total_wgs, int head_major_views, int reverse) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int P = 8, NT = HD / 16, KS = HD / 32, MI = 16 / NW;
    static_assert(HD % 32 == 0, "k-steps of 32 channels");
    constexpr bool GBF = sizeof(GST) == 2;             // grad rows given in bf16: single-term operands
    const int Nk = NKT ? NKT : mh * mw;
    const int MT = (Nk + 15) >> 4;                    // tile-row tiles of 16 (the last one reads past the tile: see below)
    uint16_t* tile = reinterpret_cast<uint16_t*>(smem);                                   // [Nk][HD] bf16
    float* DS = reinterpret_cast<float*>(smem + (((size_t)Nk * HD * 2 + 15) & ~(size_t)15));   // [Nk][kMmDss] fp32
    uint16_t* Gh = reinterpret_cast<uint16_t*>(DS + (size_t)Nk * kMmDss);                  // [32][HD] bf16 hi
    uint16_t* Gl = Gh + 32 * HD;                                                           // [32][HD] bf16 lo
    // (rows Nk .. 16 MT - 1 of the last tile-row tile are read from whatever follows the tile / DS in LDS: they only
    //  feed D rows that are never stored and d(value) rows that are never written; the host wrapper checks the slack)
    // XCD-aware placement (see k_sca_fwd_cs): the heads of a (viewpoint, camera) share lists, uv and the lines of the
    // grad / offset / logit rows
    int bid = (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3);
    if (bid >= total_wgs) return;
    // last viewpoint first: the grad rows (531 MB at 192 viewpoints) were written just before this launch in ascending row
    // order through the 256-MB memory-side cache -- the end of the tensor is what is still resident (k_sca_fwd_cs: `reverse`)
    if (reverse) bid = total_wgs - 1 - bid;
    const int h = bid % heads;
    bid /= heads;
    const int c = bid % Ncam, b = bid / Ncam;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    // (said out loud: without it every `wave + NW * i < tiles` below is a run-time branch, and a fragment read cannot be hoisted
    //  over a branch -- the D^T phase compiled to "ds_read_b128, s_waitcnt lgkmcnt(0), one MFMA" 21 times per chunk)
    __builtin_assume(wave >= 0 && wave < NW);
    const int cc = lane & 15, g = lane >> 4;
    const bool scalar = tid < 256;                    // this thread is a (voxel, point) of the scalar phases
    const int vloc = (tid & 255) >> 3, p = tid & 7;   // voxel of the chunk, sampling point
    const int ad = (D == 1) ? 0 : (p % D);
    const float inv_w = 1.0f / (float)mw, inv_h = 1.0f / (float)mh;
    const size_t rstride = (size_t)heads * HD;
    const int n_single = fwd_cnt[(b * Ncam + c) * 2], n_multi = fwd_cnt[(b * Ncam + c) * 2 + 1];
    const int s8 = (n_single + 7) & ~7;               // a wave's 8 voxels never straddle the two regions
    const int total = s8 + n_multi;
    const int* list = fwd_list + ((size_t)b * Ncam + c) * Nq;

    // (head-major value: the tile is one contiguous block, a row is HD elements; d(value) below keeps the reference layout,
    //  which is what the weight-gradient GEMM of value_proj reads as a plain [rows, heads*HD] matrix)
    if (head_major_views)
        stage_tile<HD, uint16_t>(tile, value + ((((size_t)h * head_major_views + b) * Ncam + c) * Nk) * HD, (size_t)HD, Nk, wave, NW);
    else
        stage_tile<HD, uint16_t>(tile, value + ((size_t)b * Ncam + c) * Nk * rstride + (size_t)h * HD, rstride, Nk, wave, NW);

    const f32x4_t zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
    f32x4_t acc[MI][NT];                              // d(value): tile-row tiles wave, wave + NW, ...; lane (channel cc, rows 4g..4g+3)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mi][nt] = zero4;

    // operands of this thread's (voxel, point) and its share of the voxel's grad row, software pipelined over the
    // chunks: voxel ids two chunks ahead, everything that hangs off the id one chunk ahead (a chunk is ~2 us of work;
    // without this every chunk started with two dependent global-memory latencies)
    constexpr int SEG = HD / 8;                        // floats of the grad row this thread converts (12)
    // (the grad-row segment stays as LOADED: unpacking bf16 -> float at load time made the compiler wait for the loads it had
    //  just issued -- s_waitcnt vmcnt(3) / vmcnt(2) in every chunk, a global round trip in front of the first barrier)
    typedef typename std::conditional<GBF, uint2, float4>::type GRaw;
    struct Ops {
        unsigned m;
        float lg;
        float2 of, u;
        GRaw graw[SEG / 4];
    };
    auto entry_of = [&](int e0, bool& live, bool& multi) -> int {     // voxel id (clamped to a live one), flags
        const int e = e0 + vloc;
        multi = e >= s8;
        const int idx = multi ? e - s8 : e;
        const int lim = multi ? n_multi : n_single;
        live = e < total && idx < lim;
        const int pos = multi ? Nq - 1 - min(idx, max(lim - 1, 0)) : min(idx, max(lim - 1, 0));
        return lim > 0 ? list[pos] : 0;                // (an empty region's slots of the list are uninitialised)
    };
    auto load_ops = [&](int n) -> Ops {
        Ops o;
        const size_t qh = ((size_t)b * Nq + n) * heads + h;
        o.m = (unsigned)vis[(size_t)b * Nq + n];
        o.lg = logits[qh * P + p];
        o.of = *reinterpret_cast<const float2*>(offs + (qh * P + p) * 2);
        o.u = *reinterpret_cast<const float2*>(uv + ((((size_t)b * Ncam + c) * Nq + n) * D + ad) * 2);
        const GST* grow = gslots + qh * HD + p * SEG;
#pragma unroll
        for (int i = 0; i < SEG / 4; ++i) o.graw[i] = *reinterpret_cast<const GRaw*>(grow + 4 * i);
        return o;
    };
    bool live_c = false, multi_c = false, live_n = false, multi_n = false;
    int n_c = (scalar && total > 0) ? entry_of(0, live_c, multi_c) : 0;
    int n_n = (scalar && total > 32) ? entry_of(32, live_n, multi_n) : 0;
    Ops ops = {};
    if (scalar && total > 0) ops = load_ops(n_c);
    VER_TLQ_INIT;
    VER_TLQ(0);
    for (int e0 = 0; e0 < total; e0 += 32) {
        if (e0 == 64) VER_TLQ(1);
        const bool live = live_c, multi = multi_c;     // multi is wave-uniform
        const int n = n_c;
        const unsigned m = live ? ops.m : 0u;
        const float icnt = m ? 1.0f / (float)__popc(m) : 0.0f;
        const size_t qh = ((size_t)b * Nq + n) * heads + h;
        const float lg = ops.lg;
        const float2 of = ops.of, u = ops.u;
        if (scalar) {
            uint16_t* gh = Gh + vloc * HD + p * SEG;
            uint16_t* gl = Gl + vloc * HD + p * SEG;
#pragma unroll
            for (int i = 0; i < SEG / 4; ++i) {
                __bf16 hi[4], lo[4];
                float gf[4];
                if constexpr (GBF) {
                    const uint2 t2 = ops.graw[i];
                    gf[0] = __uint_as_float(t2.x << 16); gf[1] = __uint_as_float(t2.x & 0xffff0000u);
                    gf[2] = __uint_as_float(t2.y << 16); gf[3] = __uint_as_float(t2.y & 0xffff0000u);
                } else {
                    const float4 t4 = ops.graw[i];
                    gf[0] = t4.x; gf[1] = t4.y; gf[2] = t4.z; gf[3] = t4.w;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float gs = gf[j] * icnt;
                    hi[j] = (__bf16)gs;
                    lo[j] = (__bf16)(gs - (float)hi[j]);
                }
                *reinterpret_cast<uint2*>(gh + 4 * i) = *reinterpret_cast<const uint2*>(hi);
                if constexpr (!GBF) *reinterpret_cast<uint2*>(gl + 4 * i) = *reinterpret_cast<const uint2*>(lo);
            }
        }
        // next chunk's operands and the id after that: in flight during the rest of this chunk
        n_c = n_n;
        live_c = live_n;
        multi_c = multi_n;
        if (scalar && e0 + 32 < total) ops = load_ops(n_c);
        if (scalar && e0 + 64 < total) n_n = entry_of(e0 + 64, live_n, multi_n);
        if (e0 == 64) VER_TLQ(2);
        if (e0 == 0) __builtin_amdgcn_s_waitcnt(0);   // the tile's LDS-DMA has landed
        __syncthreads();
        if (e0 == 64) VER_TLQ(3);
        // ---------------- D^T[k][v] = sum_ch V[k][ch] G[v][ch]   (A = tile rows, B = grad rows, both row-major in LDS)
        // The 2 x KS grad-row fragments (hi, lo) are read once per chunk; the MT x 2 output tiles are dealt to the waves
        // one by one (26 tiles over 4 waves: 7, 7, 6, 6).
        {
            mm_bf16x8 gbh[2][KS], gbl[2][KS];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    gbh[nt][ks] = *reinterpret_cast<const mm_bf16x8*>(Gh + (nt * 16 + cc) * HD + ks * 32 + 8 * g);
                    if constexpr (!GBF) gbl[nt][ks] = *reinterpret_cast<const mm_bf16x8*>(Gl + (nt * 16 + cc) * HD + ks * 32 + 8 * g);
                    else gbl[nt][ks] = gbh[nt][ks];
                }
            // (unrolled: the 7 tiles of a wave are independent accumulation chains for the scheduler to interleave;
            //  a chain of 2 KS dependent MFMAs per tile was 780 cycles per tile when executed one tile at a time)
            constexpr int UMAX = NKT ? (2 * ((NKT + 15) / 16) + NW - 1) / NW : 32 / NW;
            f32x4_t d[UMAX];
#pragma unroll
            for (int ui = 0; ui < UMAX; ++ui) d[ui] = zero4;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
                for (int ui = 0; ui < UMAX; ++ui) {
                    // (with the range of `wave` known -- __builtin_assume above -- only a wave's LAST tile is conditional;
                    //  computing it unconditionally on a clamped index measured the same: 425-430 against 429-431 us)
                    const int un = wave + NW * ui;
                    if (un < 2 * MT) {
                        const int mt = un >> 1, nt = un & 1;
                        const mm_bf16x8 a = *reinterpret_cast<const mm_bf16x8*>(tile + (size_t)(mt * 16 + cc) * HD + ks * 32 + 8 * g);
                        const mm_bf16x8 bhh = nt ? gbh[1][ks] : gbh[0][ks], bll = nt ? gbl[1][ks] : gbl[0][ks];
                        d[ui] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bhh, d[ui], 0, 0, 0);
                        if constexpr (!GBF) d[ui] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bll, d[ui], 0, 0, 0);
                    }
                }
            }
#pragma unroll
            for (int ui = 0; ui < UMAX; ++ui) {
                const int un = wave + NW * ui;
                if (un < 2 * MT) {
                    const int mt = un >> 1, nt = un & 1;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int row = mt * 16 + 4 * g + i;
                        if (row < Nk) DS[row * kMmDss + nt * 16 + cc] = d[ui][i];
                    }
                }
            }
        }
        if (e0 == 64) VER_TLQ(4);
        __syncthreads();
        if (e0 == 64) VER_TLQ(5);
        // ---------------- this thread's sample: pick its four dots, d(offset), d(logit), and its four events
        float coef[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        int key[4] = {0, 0, 0, 0};
        if (scalar) {
            const float mx = group_max<8>(lg);
            const float ex = __expf(lg - mx);
            const float a = ex / group_sum<8>(ex);
            Bilinear s;
            bilinear_setup<true>(u.x + of.x * inv_w, u.y + of.y * inv_h, mh, mw, s);
            float sa = 0.0f, sx = 0.0f, sy = 0.0f;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float dot = DS[s.key[t] * kMmDss + vloc];
                sa += s.w[t] * dot;
                sx += s.gx[t] * dot;
                sy += s.gy[t] * dot;
                coef[t] = live ? a * s.w[t] : 0.0f;
                key[t] = s.key[t];
            }
            const float dsum = group_sum<8>(a * sa);
            const float gx = a * sx, gy = a * sy, gl = a * (sa - dsum);         // d x_pix / d offset = 1
            if (live) {
                float* go = goffs + (qh * P + p) * 2;
                float* gw = glogits + qh * P + p;
                if (!multi) {
                    *reinterpret_cast<float2*>(go) = make_float2(gx, gy);
                    *gw = gl;
                } else {
                    atomicAdd(go, gx);
                    atomicAdd(go + 1, gy);
                    atomicAdd(gw, gl);
                }
            }
        }
        if (e0 == 64) VER_TLQ(6);
        // ---------------- S^T[k][v]: zero, then add the chunk's events.  COLUMN-LOCAL (round 6): a sample reads its dots from
        // column vloc of DS and adds its events into column vloc, and the eight points of a voxel sit in one wave -- so the
        // columns 8 w .. 8 w + 7 belong to scalar wave w from the sample phase to the scatter: it zeroes them itself (rows on the
        // lanes, 32 bytes per row) and no workgroup barrier is needed in between (LDS operations of a wave complete in order).
        // Two barriers and a workgroup-wide zero pass (0.9 k of a chunk's 8 k cycles) less.
        if (scalar) {
            float* col = DS + 8 * wave;
            for (int r = lane; r < Nk; r += 64) {
                *reinterpret_cast<float4*>(col + r * kMmDss) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                *reinterpret_cast<float4*>(col + r * kMmDss + 4) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            }
        }
        if (e0 == 64) VER_TLQ(7);
        if (e0 == 64) VER_TLQ(8);
#pragma unroll
        for (int t = 0; t < 4; ++t)          // S <= 1: 2^-30 fixed point, integer LDS atomics (float ones retire ~0.5 lane/clk)
            if (coef[t] != 0.0f)
                atomicAdd(reinterpret_cast<unsigned*>(DS) + key[t] * kMmDss + vloc, __float2uint_rn(coef[t] * 1073741824.0f));
        if (e0 == 64) VER_TLQ(9);
        __syncthreads();
        if (e0 == 64) VER_TLQ(10);
        // ---------------- d(value)[k][ch] += sum_v S^T[k][v] G[v][ch]   (B = grad rows read transposed: 8 voxels of one channel)
        mm_bf16x8 bh[NT], bl[NT];
        {
            const unsigned toff = (unsigned)((8 * g + (cc >> 2)) * HD + 4 * (cc & 3));
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const mm_s16x4 h0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((mm_s16x4 __attribute__((address_space(3)))*)(Gh + toff + nt * 16));
                const mm_s16x4 h1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((mm_s16x4 __attribute__((address_space(3)))*)(Gh + toff + nt * 16 + 4 * HD));
                bh[nt] = __builtin_bit_cast(mm_bf16x8, (mm_s16x8)__builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7));
                if constexpr (!GBF) {
                    const mm_s16x4 l0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((mm_s16x4 __attribute__((address_space(3)))*)(Gl + toff + nt * 16));
                    const mm_s16x4 l1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((mm_s16x4 __attribute__((address_space(3)))*)(Gl + toff + nt * 16 + 4 * HD));
                    bl[nt] = __builtin_bit_cast(mm_bf16x8, (mm_s16x8)__builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7));
                } else {
                    bl[nt] = bh[nt];
                }
            }
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int mt = wave + NW * mi;
            if (mt < MT) {
                const float* srow = DS + (size_t)(mt * 16 + cc) * kMmDss + 8 * g;
                const uint4 q0 = *reinterpret_cast<const uint4*>(srow), q1 = *reinterpret_cast<const uint4*>(srow + 4);
                constexpr float kFix = 1.0f / 1073741824.0f;
                const float f[8] = {(float)q0.x * kFix, (float)q0.y * kFix, (float)q0.z * kFix, (float)q0.w * kFix,
                                    (float)q1.x * kFix, (float)q1.y * kFix, (float)q1.z * kFix, (float)q1.w * kFix};
                mm_bf16x8 ah, al;
                mm_split(f, ah, al);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    acc[mi][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[nt], acc[mi][nt], 0, 0, 0);
                    if constexpr (!GBF) acc[mi][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[nt], acc[mi][nt], 0, 0, 0);
                    acc[mi][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[nt], acc[mi][nt], 0, 0, 0);
                }
            }
        }
        if (e0 == 64) VER_TLQ(11);
        __syncthreads();
        if (e0 == 64) VER_TLQ(12);
    }
    VER_TLQ(13);
    // ---------------- d(value) tile of this (camera, head): written in full (zeros for a camera that sees nothing)
    GVT* gv = gvalue + ((size_t)b * Ncam + c) * Nk * rstride + (size_t)h * HD;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int mt = wave + NW * mi;
        if (mt < MT) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = mt * 16 + 4 * g + i;
                if (row < Nk) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        if constexpr (sizeof(GVT) == 4) {
                            gv[(size_t)row * rstride + nt * 16 + cc] = acc[mi][nt][i];
                        } else {
                            const __bf16 r16 = (__bf16)acc[mi][nt][i];
                            gv[(size_t)row * rstride + nt * 16 + cc] = __builtin_bit_cast(uint16_t, r16);
                        }
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
namespace {

int env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}

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
    VER_REQUIRE(vdt == VER_F32 || vdt == VER_BF16, VER_EINVAL, "ver_sca: value_dtype %d is neither VER_F32 nor VER_BF16",
                vdt);
    VER_REQUIRE(vdt == VER_F32 || hd % 8 == 0, VER_EUNSUPPORTED, "ver_sca: bf16 value needs head_dim %% 8 == 0");
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
                                  int32_t* zero_list, int32_t* zero_cnt, int32_t* fwd_list, int32_t* fwd_cnt,
                                  void* stream) {
    VER_REQUIRE(world2pixel && origin && pc_range && uv && vis && vis_list && vis_cnt && zero_list && zero_cnt &&
                    fwd_list && fwd_cnt,
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
                       zero_list, zero_cnt, fwd_list, fwd_cnt);
    return ver_check_launch("ver_project_points/k_build_lists");
}

extern "C" int ver_hits_from_mask(const uint8_t* bev_mask, int B, int Ncam, int Nq, int D, uint8_t* vis,
                                  int32_t* vis_list, int32_t* vis_cnt, int32_t* zero_list, int32_t* zero_cnt,
                                  int32_t* fwd_list, int32_t* fwd_cnt, void* stream) {
    VER_REQUIRE(bev_mask && vis && vis_list && vis_cnt && zero_list && zero_cnt && fwd_list && fwd_cnt, VER_EINVAL,
                "ver_hits_from_mask: null pointer argument");
    VER_REQUIRE(Ncam >= 1 && Ncam <= 8, VER_EUNSUPPORTED, "ver_hits_from_mask: Ncam %d outside 1..8", Ncam);
    VER_REQUIRE(B >= 0 && Nq > 0 && D > 0, VER_EINVAL, "ver_hits_from_mask: bad sizes");
    if (B == 0) return VER_OK;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_mask_to_vis, dim3((Nq + 255) / 256, B), dim3(256), 0, st, bev_mask, B, Ncam, Nq, D, vis);
    int rc = ver_check_launch("ver_hits_from_mask/k_mask_to_vis");
    if (rc) return rc;
    hipLaunchKernelGGL(k_build_lists, dim3(Ncam, B), dim3(256), 0, st, vis, Ncam, Nq, vis_list, vis_cnt,
                       zero_list, zero_cnt, fwd_list, fwd_cnt);
    return ver_check_launch("ver_hits_from_mask/k_build_lists");
}

static bool sca_bwd_use_mm() {       // read ONCE per process: the dtype query and the dispatch can never disagree
    static const int v = env_int("VER_SCA_BWD_MM", 1);
    return v != 0;
}

// value layouts ver_sca_forward / ver_sca_backward read: the reference's [B, Ncam, Nk, heads, HD] always; head-major
// [heads, B, Ncam, Nk, HD] (VER_SCA_VALUE_HEAD_MAJOR) where the fast kernels are built for it
extern "C" int ver_sca_head_major_supported(int value_dtype, int head_dim, int points, int map_h, int map_w) {
    static const int use_cs = env_int("VER_SCA_FWD_CS", 1);
    static const int cs_math = env_int("VER_SCA_FWD_MATH", 2);
    static const int cs_nload = env_int("VER_SCA_CS_NLOAD", 0);
    return (value_dtype == VER_BF16 && points == 8 && map_h == 14 && map_w == 14 && use_cs && cs_math == 2 && cs_nload == 0 &&
            (head_dim == 32 || head_dim == 64 || head_dim == 96 || head_dim == 128) && sca_bwd_use_mm()) ? 1 : 0;
}

// The zero fill of ver_sca_forward as a call of its own: it depends on the hit table only, so a caller can run it on a
// side stream under the projections that precede the gather and pass VER_SCA_ROWS_PREZEROED.
extern "C" int ver_sca_zero_rows(const int32_t* zero_list, const int32_t* zero_cnt, float* slots, int B, int Nq,
                                 int row_floats, void* stream) {
    VER_REQUIRE(B >= 0 && Nq >= 0 && row_floats > 0, VER_EINVAL, "ver_sca_zero_rows: bad sizes B=%d Nq=%d row=%d", B, Nq,
                row_floats);
    VER_REQUIRE(row_floats % 4 == 0, VER_EUNSUPPORTED, "ver_sca_zero_rows: row width not a multiple of 4");
    if (B == 0 || Nq == 0) return VER_OK;
    VER_REQUIRE(zero_list && zero_cnt && slots, VER_EINVAL, "ver_sca_zero_rows: null pointer argument");
    hipLaunchKernelGGL(k_zero_rows, dim3((Nq + 7) / 8, B), dim3(256), 0, (hipStream_t)stream, zero_list, zero_cnt, slots,
                       Nq, row_floats);
    return ver_check_launch("ver_sca_zero_rows");
}

extern "C" int ver_sca_forward(const void* value, int value_dtype, const float* offsets, const float* logits,
                               const float* uv, const uint8_t* vis, const int32_t* vis_list,
                               const int32_t* vis_cnt, const int32_t* zero_list, const int32_t* zero_cnt,
                               const int32_t* fwd_list, const int32_t* fwd_cnt,
                               float* slots, int B, int Ncam, int Nq, int D, int heads,
                               int head_dim, int points, int map_h, int map_w, int flags, void* stream) {
    int rc = check_sca(value, value_dtype, offsets, logits, uv, vis, vis_list, vis_cnt, B, Ncam, Nq, D, heads,
                       head_dim, points, map_h, map_w);
    if (rc) return rc;
    VER_REQUIRE(slots && zero_list && zero_cnt && fwd_list && fwd_cnt, VER_EINVAL,
                "ver_sca_forward: null pointer argument");
    VER_REQUIRE(Nq < (1 << 24), VER_EUNSUPPORTED, "ver_sca_forward: more than 2^24 voxels per viewpoint");
    VER_REQUIRE((heads * head_dim) % 4 == 0, VER_EUNSUPPORTED, "ver_sca_forward: row width not a multiple of 4");
    if (B == 0 || Nq == 0) return VER_OK;
    const size_t esz = value_dtype == VER_BF16 ? 2 : 4;
    const size_t tile_bytes = (size_t)map_h * map_w * head_dim * esz;
    VER_REQUIRE(tile_bytes <= kMaxLds, VER_EUNSUPPORTED,
                "ver_sca_forward: %dx%dx%d value tile (%zu B) exceeds the 160 KiB LDS", map_h, map_w, head_dim,
                tile_bytes);
    VER_REQUIRE(map_h * map_w <= 65536, VER_EUNSUPPORTED, "ver_sca_forward: map larger than 65536 tokens");
    // two ways to overlap the tile stream with the gather: one 16-wave workgroup per CU with a
    // double-buffered tile, or two 8-wave workgroups per CU with one tile each (finer work split:
    // 32 instead of 64 voxel slots per pass over a camera's ~140 visible voxels)
    static const int fwd_threads = [] {
        const char* e = getenv("VER_SCA_FWD_THREADS");
        const int t = e ? atoi(e) : 512;               // measured: 512 beats 1024 by 5-15 % (B = 32..256)
        return (t == 256 || t == 512 || t == 1024) ? t : 512;
    }();
    // double-buffer the tile inside the workgroup when the CU's LDS holds it for every resident
    // workgroup (one of 16 waves, or two of 8 waves: bf16 tiles of 14x14x96 fit four times)
    static const int force_nbuf = [] {
        const char* e = getenv("VER_SCA_FWD_NBUF");
        return e ? atoi(e) : 0;
    }();
    const size_t wgs_per_cu = kFwdThreads / fwd_threads;
    int nbuf = 2 * wgs_per_cu * tile_bytes <= kMaxLds ? 2 : 1;
    if (force_nbuf == 1 || (force_nbuf == 2 && 2 * tile_bytes <= kMaxLds)) nbuf = force_nbuf;
    const size_t lds = tile_bytes * nbuf;
    const int nchunks = (Nq + kFwdChunk - 1) / kFwdChunk;
    // heads are walked inside a workgroup (the tile stream is double buffered); split them over
    // several workgroups only while the grid would not yet fill the 256 CUs a few times over
    static const long min_wgs = [] {
        const char* e = getenv("VER_SCA_FWD_MIN_WGS");
        return e ? atol(e) : 6144L;                    // heads split until >= 12 workgroups per residency slot
                                                       // (2 per CU): 2-5 % over 1536 at B = 64..256
    }();
    int hsplit = 1;
    while (hsplit < heads && heads % (hsplit * 2) == 0 && (long)B * Ncam * hsplit * nchunks < min_wgs) hsplit *= 2;
    hipStream_t st = (hipStream_t)stream;
    VER_REQUIRE((flags & ~(VER_SCA_ROWS_PREZEROED | VER_SCA_VALUE_HEAD_MAJOR)) == 0, VER_EINVAL,
                "ver_sca_forward: unknown flags 0x%x", flags);
    const bool head_major = flags & VER_SCA_VALUE_HEAD_MAJOR;
    VER_REQUIRE(!head_major || ver_sca_head_major_supported(value_dtype, head_dim, points, map_h, map_w), VER_EUNSUPPORTED,
                "ver_sca_forward: the head-major value layout is built for bf16 tiles, 8 points, head_dim %% 32 == 0, 14x14 maps");
    if (!(flags & VER_SCA_ROWS_PREZEROED)) {
        hipLaunchKernelGGL(k_zero_rows, dim3((Nq + 7) / 8, B), dim3(256), 0, st, zero_list, zero_cnt, slots, Nq,
                           heads * head_dim);
        rc = ver_check_launch("ver_sca_forward/k_zero_rows");
        if (rc) return rc;
    }
    return dispatch_shape(head_dim, points, [&](auto hd, auto g, auto pp) {
        constexpr int HD = decltype(hd)::value, G = decltype(g)::value, P = decltype(pp)::value;
        auto launch = [&](auto kern, auto vptr) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess)
                return ver_fail(VER_ELAUNCH, "ver_sca_forward: LDS attribute: %s", hipGetErrorString(e));
            const unsigned blocks = (unsigned)B * Ncam * hsplit * nchunks;
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(fwd_threads), lds, st, vptr, offsets, logits, uv, vis, vis_list,
                               vis_cnt, slots, Ncam, Nq, D, heads, map_h, map_w, nchunks, kFwdChunk, hsplit, nbuf);
            return ver_check_launch("ver_sca_forward");
        };
        if constexpr (P == 8 && HD % 32 == 0) {
            // corner-slot kernel (fp32 and bf16 tiles); VER_SCA_FWD_CS=0 falls back to the generic 16-lane kernel.
            // Launch shape = measured optima at the vocc size (scratch/r02/sweep_*.sh); the environment variables
            // exist for scratch/bench_gather.py sweeps:
            //   threads   256 (bf16 tiles: four single-tile workgroups per CU) / 512 (fp32 tiles: two per CU)
            //   nload     0: every wave stages its share of the tile, one buffer; > 0: that many loader waves stream
            //             tiles through two buffers for a workgroup that walks many tiles (measured slower here: one
            //             16-wave workgroup per CU cannot hold more waves than four small ones, and its tiles split
            //             into 4-5 pairs per wave with a second, nearly empty phase-A pass)
            //   hsplit    heads are split over this many units (8: one head per unit)
            //   units/wg  1: one unit per workgroup (0: as many as it takes to fill the CUs `wgs_per_cu` times)
            static const int use_cs = env_int("VER_SCA_FWD_CS", 1);
            const bool f32 = value_dtype == VER_F32;
            static const int cs_threads_f32 = env_int("VER_SCA_CS_THREADS_F32", 512);
            static const int cs_threads_bf16 = env_int("VER_SCA_CS_THREADS_BF16", 256);
            static const int cs_nload = env_int("VER_SCA_CS_NLOAD", 0);
            static const int cs_hsplit = env_int("VER_SCA_CS_HSPLIT", 8);
            static const int cs_upw = env_int("VER_SCA_CS_UNITS_PER_WG", 1);
            static const int cs_wgs_per_cu = env_int("VER_SCA_CS_WGS_PER_CU", 1);
            static const int cs_reverse = env_int("VER_SCA_CS_REVERSE", 1);
            int pt = f32 ? cs_threads_f32 : cs_threads_bf16;
            if (pt % 64 != 0 || pt < 128 || pt > 1024) pt = f32 ? 512 : 256;
            const int nlp = (cs_nload >= 0 && cs_nload < pt / 64) ? cs_nload : 0;
            const size_t tb = (tile_bytes + 15) & ~(size_t)15;
            const size_t ldsp = (nlp > 0 ? 2 : 1) * tb + (size_t)(pt / 64) * 512;
            // k_sca_fwd_cs addresses slots / offsets / logits / uv with 32-bit byte offsets (`__umul24` row products):
            // every one of them must stay below 2^32 and the 24-bit multiplicands below 2^24, else the generic
            // kernel (64-bit addressing) takes the launch
            const size_t row_b = (size_t)heads * head_dim * 4;
            const bool cs_addr_ok = (size_t)Nq * row_b < ((size_t)1 << 32) &&
                                    (size_t)Nq * heads * points * 8 < ((size_t)1 << 32) &&
                                    (size_t)Nq * D * 8 < ((size_t)1 << 32) && row_b < ((size_t)1 << 24) &&
                                    (size_t)heads * points * 8 < ((size_t)1 << 24) && Nq < (1 << 24) &&
                                    // unit indices go through 32-bit magic divisions (cs_div): exact below 2^32 / divisor
                                    (size_t)B * Ncam * nchunks * heads < ((size_t)1 << 24);
            if (use_cs && ldsp <= kMaxLds && cs_addr_ok) {
                int hsp = 1;
                while (hsp < cs_hsplit && hsp < heads && heads % (hsp * 2) == 0) hsp *= 2;
                const int units = B * Ncam * nchunks * hsp;
                int upw = cs_upw;
                if (upw <= 0) {
                    int dev = 0, cus = 256;
                    if (hipGetDevice(&dev) == hipSuccess)
                        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
                    if (cus <= 0) cus = 256;
                    const int slots_wg = cus * (cs_wgs_per_cu > 0 ? cs_wgs_per_cu : 1);
                    upw = (units + slots_wg - 1) / slots_wg;
                }
                const int grid = ((units + upw - 1) / upw + 7) & ~7;     // multiple of 8: dealt to the XCDs in blocks
                const CsMagic mg = {cs_magic(hsp), cs_magic(nchunks), cs_magic(Ncam), cs_magic(heads / hsp),
                                    cs_magic(pt / 64 - nlp)};
                auto launch_cs = [&](auto kern, auto vptr) {
                    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsp);
                    if (e != hipSuccess)
                        return ver_fail(VER_ELAUNCH, "ver_sca_forward: LDS attribute: %s", hipGetErrorString(e));
                    hipLaunchKernelGGL(kern, dim3(grid), dim3(pt), ldsp, st, vptr, offsets, logits, uv, vis, fwd_list,
                                       fwd_cnt, slots, Ncam, Nq, D, heads, map_h, map_w, nchunks, kFwdChunk, hsp, units, upw,
                                       nlp, head_major ? B : 0, cs_reverse, heads / hsp, mg);
                    return ver_check_launch("ver_sca_forward");
                };
                const bool k196 = map_h * map_w == 196;                // plane offsets become immediates
                if (f32) {
                    if (k196) return launch_cs(k_sca_fwd_cs<HD, float, 196>, (const float*)value);
                    return launch_cs(k_sca_fwd_cs<HD, float, 0>, (const float*)value);
                }
                // VER_SCA_FWD_MATH: 0 bf16 tile + fp32 unpack and accumulation (exact), 2 fp16 tile + packed fp16
                // accumulation over a voxel's points (default: DESIGN.md section 3.1)
                static const int cs_math = env_int("VER_SCA_FWD_MATH", 2);
                static const int cs_one = env_int("VER_SCA_CS_ONE", 1);          // 0: the general form for every launch shape
                const bool one = cs_one && nlp == 0 && upw == 1 && hsp == heads && nchunks == 1;
                if (head_major && one) return launch_cs(k_sca_fwd_cs<HD, uint16_t, 196, 2, true, true>, (const uint16_t*)value);
                if (head_major) return launch_cs(k_sca_fwd_cs<HD, uint16_t, 196, 2, true>, (const uint16_t*)value);
                if (k196 && cs_math == 2 && one) return launch_cs(k_sca_fwd_cs<HD, uint16_t, 196, 2, false, true>, (const uint16_t*)value);
                if (k196 && cs_math == 2) return launch_cs(k_sca_fwd_cs<HD, uint16_t, 196, 2>, (const uint16_t*)value);
                if (k196) return launch_cs(k_sca_fwd_cs<HD, uint16_t, 196>, (const uint16_t*)value);
                if (cs_math == 2) return launch_cs(k_sca_fwd_cs<HD, uint16_t, 0, 2>, (const uint16_t*)value);
                return launch_cs(k_sca_fwd_cs<HD, uint16_t, 0>, (const uint16_t*)value);
            }
        }
        if (head_major)
            return ver_fail(VER_EUNSUPPORTED, "ver_sca_forward: head-major value needs the corner-slot kernel for this shape");
        if constexpr (HD % 8 == 0) {
            if (value_dtype == VER_BF16) return launch(k_sca_fwd<HD, G, P, uint16_t>, (const uint16_t*)value);
        }
        return launch(k_sca_fwd<HD, G, P, float>, (const float*)value);
    });
}

// dtype ver_sca_backward writes d(value) in most cheaply for this problem: VER_BF16 on the matrix-core path (bf16 value
// tiles, 8 points, head_dim % 32 == 0, tile fits), else VER_F32.  VER_F32 is always accepted.
extern "C" int ver_sca_backward_grad_dtype(int value_dtype, int head_dim, int points, int map_h, int map_w) {
    if (value_dtype != VER_BF16 || points != 8 || head_dim % 32 != 0 || head_dim > 128) return VER_F32;
    if (!sca_bwd_use_mm()) return VER_F32;
    const int nk = map_h * map_w, mt = (nk + 15) / 16;
    const size_t tile_b = ((size_t)nk * head_dim * 2 + 15) & ~(size_t)15, ds_b = (size_t)nk * kMmDss * 4,
                 g_b = (size_t)2 * 32 * head_dim * 2;
    const bool slack_ok = (size_t)(16 * mt - nk) * head_dim * 2 <= ds_b && (size_t)(16 * mt - nk) * kMmDss * 4 <= g_b;
    return (mt <= 16 && slack_ok && tile_b + ds_b + g_b <= kMaxLds) ? VER_BF16 : VER_F32;
}

extern "C" int ver_sca_backward(const void* value, int value_dtype, const float* offsets, const float* logits,
                                const float* uv, const uint8_t* vis, const int32_t* vis_list,
                                const int32_t* vis_cnt, const int32_t* fwd_list, const int32_t* fwd_cnt,
                                const void* grad_slots_v, void* grad_value, int grad_value_dtype,
                                float* grad_offsets, float* grad_logits, int B, int Ncam, int Nq, int D,
                                int heads, int head_dim, int points, int map_h, int map_w, int flags, void* stream) {
    const float* grad_slots = reinterpret_cast<const float*>(grad_slots_v);     // (bf16 with VER_SCA_GRAD_SLOTS_BF16)
    int rc = check_sca(value, value_dtype, offsets, logits, uv, vis, vis_list, vis_cnt, B, Ncam, Nq, D, heads,
                       head_dim, points, map_h, map_w);
    if (rc) return rc;
    VER_REQUIRE(grad_slots && grad_value && grad_offsets && grad_logits && fwd_list && fwd_cnt, VER_EINVAL,
                "ver_sca_backward: null pointer argument");
    VER_REQUIRE((flags & ~(VER_SCA_VALUE_HEAD_MAJOR | VER_SCA_GRAD_SLOTS_BF16)) == 0, VER_EINVAL,
                "ver_sca_backward: unknown flags 0x%x", flags);
    const bool head_major = flags & VER_SCA_VALUE_HEAD_MAJOR, grad_bf16 = flags & VER_SCA_GRAD_SLOTS_BF16;
    VER_REQUIRE(!grad_bf16 || ver_sca_backward_grad_dtype(value_dtype, head_dim, points, map_h, map_w) == VER_BF16,
                VER_EUNSUPPORTED, "ver_sca_backward: bf16 grad_slots are read by the matrix-core kernel only");
    VER_REQUIRE(!head_major || (ver_sca_head_major_supported(value_dtype, head_dim, points, map_h, map_w) &&
                                ver_sca_backward_grad_dtype(value_dtype, head_dim, points, map_h, map_w) == VER_BF16),
                VER_EUNSUPPORTED, "ver_sca_backward: the head-major value layout is read by the matrix-core kernel only");
    VER_REQUIRE(grad_value_dtype == ver_sca_backward_grad_dtype(value_dtype, head_dim, points, map_h, map_w) ||
                    grad_value_dtype == VER_F32,
                VER_EUNSUPPORTED, "ver_sca_backward: grad_value_dtype %d not available for this shape (ask "
                "ver_sca_backward_grad_dtype)", grad_value_dtype);
    if (B == 0 || Nq == 0) return VER_OK;
    const size_t esz = value_dtype == VER_BF16 ? 2 : 4;
    const size_t lds = (size_t)map_h * map_w * head_dim * (sizeof(float) + esz);
    const int nchunks = (Nq + kBwdChunk - 1) / kBwdChunk;
    hipStream_t st = (hipStream_t)stream;
    const size_t nsmall = (size_t)B * Nq * heads * points;
    // (zero fills by kernel, not hipMemsetAsync: see ver_zero_async -- this launcher runs inside replayed hipGraphs)
    rc = ver_zero_async(grad_offsets, nsmall * 2 * sizeof(float), st);
    if (!rc) rc = ver_zero_async(grad_logits, nsmall * sizeof(float), st);
    if (!rc && nchunks > 1 && grad_value_dtype == VER_F32)                  // (the chunked two-kernel path flushes with atomics)
        rc = ver_zero_async(grad_value, (size_t)B * Ncam * map_h * map_w * heads * head_dim * sizeof(float), st);
    float* grad_value_f32 = reinterpret_cast<float*>(grad_value);          // (the two-kernel paths write fp32 only)
    if (rc) return rc;
    return dispatch_shape(head_dim, points, [&](auto hd, auto g, auto pp) {
        constexpr int HD = decltype(hd)::value, G = decltype(g)::value, P = decltype(pp)::value;
        if constexpr (P == 8 && HD % 32 == 0) {
            // bf16 value tiles: everything on the matrix cores, one kernel (VER_SCA_BWD_MM=0: the two-kernel path below)
            const bool use_mm = sca_bwd_use_mm();
            static const int bwd_reverse = env_int("VER_SCA_BWD_REVERSE", 1);
            const int nk = map_h * map_w, mt = (nk + 15) / 16;
            const size_t tile_b = ((size_t)nk * HD * 2 + 15) & ~(size_t)15, ds_b = (size_t)nk * kMmDss * 4, g_b = (size_t)2 * 32 * HD * 2;
            // the last tile-row tile reads (16 mt - nk) rows past the tile / past DS: they must stay inside the allocation
            const bool slack_ok = (size_t)(16 * mt - nk) * HD * 2 <= ds_b && (size_t)(16 * mt - nk) * kMmDss * 4 <= g_b;
            if (use_mm && value_dtype == VER_BF16 && mt <= 16 && slack_ok && tile_b + ds_b + g_b <= kMaxLds) {
                const size_t lds_mm = tile_b + ds_b + g_b;
                auto launch_mm = [&](auto kern, auto gptr, auto sptr) {
                    hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_mm);
                    if (e2 != hipSuccess)
                        return ver_fail(VER_ELAUNCH, "ver_sca_backward: LDS attribute: %s", hipGetErrorString(e2));
                    typedef std::remove_pointer_t<decltype(gptr)> gv_t;
                    typedef std::remove_cv_t<std::remove_pointer_t<decltype(sptr)>> gs_t;
                    const int wgs = B * Ncam * heads;
                    hipLaunchKernelGGL(kern, dim3((unsigned)((wgs + 7) & ~7)), dim3(kMmWaves * 64), lds_mm, st, (const uint16_t*)value,
                                       offsets, logits, uv, vis, fwd_list, fwd_cnt, (const gs_t*)(const void*)grad_slots,
                                       (gv_t*)grad_value, grad_offsets, grad_logits, Ncam, Nq, D, heads, map_h, map_w, wgs,
                                       head_major ? B : 0, bwd_reverse);
                    return ver_check_launch("ver_sca_backward/k_sca_bwd_mm");
                };
                if (grad_bf16) {
                    VER_REQUIRE(grad_value_dtype == VER_BF16, VER_EUNSUPPORTED, "ver_sca_backward: bf16 grad_slots go with bf16 grad_value");
                    if (nk == 196) return launch_mm(k_sca_bwd_mm<HD, 196, uint16_t, kMmWaves, uint16_t>, (uint16_t*)nullptr, (const uint16_t*)nullptr);
                    return launch_mm(k_sca_bwd_mm<HD, 0, uint16_t, kMmWaves, uint16_t>, (uint16_t*)nullptr, (const uint16_t*)nullptr);
                }
                if (grad_value_dtype == VER_BF16) {
                    if (nk == 196) return launch_mm(k_sca_bwd_mm<HD, 196, uint16_t, kMmWaves>, (uint16_t*)nullptr, (const float*)nullptr);
                    return launch_mm(k_sca_bwd_mm<HD, 0, uint16_t, kMmWaves>, (uint16_t*)nullptr, (const float*)nullptr);
                }
                if (nk == 196) return launch_mm(k_sca_bwd_mm<HD, 196, float, kMmWaves>, (float*)nullptr, (const float*)nullptr);
                return launch_mm(k_sca_bwd_mm<HD, 0, float, kMmWaves>, (float*)nullptr, (const float*)nullptr);
            }
        }
        // everything below writes d(value) as fp32 (twice the bytes of a bf16 buffer): never fall through with one
        VER_REQUIRE(grad_value_dtype == VER_F32, VER_EUNSUPPORTED,
                    "ver_sca_backward: bf16 d(value) is only written by the matrix-core kernel; this shape runs the fp32 paths");
        if constexpr (G == 16) {
            if (map_h * map_w <= kValMaxRows) {
                // ---- d(offsets), d(logits): forward-shaped kernel
                const size_t tile_bytes = (size_t)map_h * map_w * head_dim * esz;
                static const int off_threads = [] {
                    const char* ev = getenv("VER_SCA_BWD_THREADS");
                    const int t = ev ? atoi(ev) : 512;
                    return (t == 256 || t == 512 || t == 1024) ? t : 512;
                }();
                static const long off_min_wgs = [] {
                    const char* ev = getenv("VER_SCA_BWD_MIN_WGS");
                    return ev ? atol(ev) : 12288L;      // one head per workgroup up to B*Ncam = 1536: 4-7 % faster than 768
                }();
                const int nbuf = (off_threads == kFwdThreads && 2 * tile_bytes <= kMaxLds) ? 2 : 1;
                const size_t lds_off = tile_bytes * nbuf;
                const int nch = (Nq + kFwdChunk - 1) / kFwdChunk;
                int hsplit = 1;
                while (hsplit < heads && heads % (hsplit * 2) == 0 && (long)B * Ncam * hsplit * nch < off_min_wgs)
                    hsplit *= 2;
                auto launch_off = [&](auto kern, auto vptr) {
                    hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_off);
                    if (e2 != hipSuccess)
                        return ver_fail(VER_ELAUNCH, "ver_sca_backward: LDS attribute: %s", hipGetErrorString(e2));
                    hipLaunchKernelGGL(kern, dim3((unsigned)B * Ncam * hsplit * nch), dim3(off_threads), lds_off, st, vptr,
                                       offsets, logits, uv, vis, vis_list, vis_cnt, grad_slots, grad_offsets,
                                       grad_logits, Ncam, Nq, D, heads, map_h, map_w, nch, kFwdChunk, hsplit, nbuf);
                    return ver_check_launch("ver_sca_backward/k_sca_bwd_off");
                };
                int r2;
                if (value_dtype == VER_BF16) {
                    if constexpr (HD % 8 == 0) r2 = launch_off(k_sca_bwd_off<HD, P, uint16_t>, (const uint16_t*)value);
                    else r2 = ver_fail(VER_EUNSUPPORTED, "bf16 value needs head_dim %% 8 == 0");
                } else {
                    r2 = launch_off(k_sca_bwd_off<HD, P, float>, (const float*)value);
                }
                if (r2) return r2;
                // ---- d(value): event sort, no value tile
                const size_t lds_val = (size_t)kValSub * HD * 4 + (size_t)kValSub * P * 4 * (4 + 2 + 2) +
                                       3 * kValMaxRows * sizeof(int);
                auto kv = k_sca_bwd_val<HD, P>;
                hipError_t e3 = hipFuncSetAttribute(reinterpret_cast<const void*>(kv),
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_val);
                if (e3 != hipSuccess)
                    return ver_fail(VER_ELAUNCH, "ver_sca_backward: LDS attribute: %s", hipGetErrorString(e3));
                hipLaunchKernelGGL(kv, dim3((unsigned)B * Ncam * heads * nchunks), dim3(kValThreads), lds_val, st, offsets,
                                   logits, uv, vis, vis_list, vis_cnt, grad_slots, grad_value_f32, Ncam, Nq, D, heads,
                                   map_h, map_w, nchunks, kBwdChunk);
                return ver_check_launch("ver_sca_backward/k_sca_bwd_val");
            }
        }
        // narrow heads / large maps: single-kernel LDS-atomic path
        VER_REQUIRE(lds <= kMaxLds, VER_EUNSUPPORTED, "ver_sca_backward: %dx%dx%d tiles (%zu B) exceed LDS", map_h,
                    map_w, head_dim, lds);
        auto launch = [&](auto kern, auto vptr) {
            hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e2 != hipSuccess)
                return ver_fail(VER_ELAUNCH, "ver_sca_backward: LDS attribute: %s", hipGetErrorString(e2));
            const unsigned blocks = (unsigned)B * Ncam * heads * nchunks;
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), lds, st, vptr, offsets, logits, uv, vis, vis_list, vis_cnt,
                               grad_slots, grad_value_f32, grad_offsets, grad_logits, Ncam, Nq, D, heads, map_h, map_w,
                               nchunks, kBwdChunk);
            return ver_check_launch("ver_sca_backward");
        };
        if constexpr (HD % 8 == 0) {
            if (value_dtype == VER_BF16) return launch(k_sca_bwd<HD, G, P, uint16_t>, (const uint16_t*)value);
        }
        return launch(k_sca_bwd<HD, G, P, float>, (const float*)value);
    });
}
