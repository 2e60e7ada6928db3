// Layout changes around the coarse-to-fine head (include/ver_ops.h: ver_convt_weight_*, ver_lattice_transpose):
// LDS-tiled transposes, fully coalesced on the wide side.
//
//  * ConvTranspose3d weight [Ci,Co,3,5,5] fp32 (reference: dense_heads/voxelformer_occupancy_head.py
//    :251-258) <-> correlation taps [75,Ci,Co] in the compute dtype, tap (a,b,c) = W[.., 2-a, 4-b, 4-c]
//    (a transposed convolution is a correlation with the flipped kernel).  As torch ops this is a
//    cast + flip + permuted copy per layer and step (0.6 ms each way for 44 M elements).
//  * even lattice, channels-last (plain [B,Z,H,W,C] or planar 4x[B,Z,H/2,W/2,C]) <-> channel-first
//    rows dst[b, ((c*Z + z)*H + y)*W + x] (row stride given): the order in which the reference's raw
//    .view (head:564) reads the volume, consumed by the gathered occ_proj GEMMs.
#include "ver_common.h"

namespace {
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {
    const uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
constexpr int kTaps = 75;
constexpr int kPairs = 64;       // (ci, co) pairs per workgroup
}  // namespace
#define VER_BLOCKS_VEC_SLICES 8  // ci slices of ver_blocks_vec_forward (= rows of its partial result, include/ver_ops.h)

// w [P][75] fp32 -> k [75][P] (bf16 or fp32), P = Ci*Co, tap order flipped
template <bool BF16>
__global__ __launch_bounds__(256) void k_convt_weight_fwd(const float* __restrict__ w, void* __restrict__ k, long P) {
    __shared__ float tile[kPairs * kTaps + 1];
    const long p0 = (long)blockIdx.x * kPairs;
    const int np = (int)((P - p0) < kPairs ? (P - p0) : kPairs);
    for (int i = threadIdx.x; i < np * kTaps; i += 256) tile[i] = w[p0 * kTaps + i];
    __syncthreads();
    for (int i = threadIdx.x; i < kTaps * kPairs; i += 256) {
        const int t = i / kPairs, p = i % kPairs;
        if (p >= np) continue;
        const int a = t / 25, b = (t / 5) % 5, c = t % 5;
        const float v = tile[p * kTaps + ((2 - a) * 5 + (4 - b)) * 5 + (4 - c)];
        if (BF16)
            reinterpret_cast<uint16_t*>(k)[(long)t * P + p0 + p] = f32_to_bf16(v);
        else
            reinterpret_cast<float*>(k)[(long)t * P + p0 + p] = v;
    }
}

// adjoint: dk [75][P] -> dw [P][75] fp32
template <bool BF16>
__global__ __launch_bounds__(256) void k_convt_weight_bwd(const void* __restrict__ dk, float* __restrict__ dw, long P) {
    __shared__ float tile[kPairs * kTaps + 1];
    const long p0 = (long)blockIdx.x * kPairs;
    const int np = (int)((P - p0) < kPairs ? (P - p0) : kPairs);
    for (int i = threadIdx.x; i < kTaps * kPairs; i += 256) {
        const int t = i / kPairs, p = i % kPairs;
        if (p >= np) continue;
        const int a = t / 25, b = (t / 5) % 5, c = t % 5;
        float v;
        if (BF16)
            v = __uint_as_float((uint32_t)reinterpret_cast<const uint16_t*>(dk)[(long)t * P + p0 + p] << 16);
        else
            v = reinterpret_cast<const float*>(dk)[(long)t * P + p0 + p];
        tile[p * kTaps + ((2 - a) * 5 + (4 - b)) * 5 + (4 - c)] = v;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < np * kTaps; i += 256) dw[p0 * kTaps + i] = tile[i];
}

// adjoint from the class gradients of a z-split lattice layer: every tap t gathers up to two [Ci x Co] blocks of ONE
// row-major buffer (element offsets off[t][0..1], -1 = none; row pitch ld) -- the "lower half" and "upper half" weight
// gradients of the class GEMMs (dense_heads/upsample.py::_LatticeLayerZ4) -- plus prev_bias[ci] * d_v[t][co] (the constant
// rows' gradient through v = prev_bias^T K[t]); fp32 sums -> dw [P][75] fp32, tap order flipped.  Replaces two zero-filled
// [75 Ci, Co] buffers, eight indexed copies, an add, an addcmul and k_convt_weight_bwd.
template <bool BF16>
__global__ __launch_bounds__(256) void k_convt_weight_bwd_blocks(const void* __restrict__ src, const long* __restrict__ off,
                                                                 long ld, const void* __restrict__ pb,
                                                                 const void* __restrict__ dv, float* __restrict__ dw,
                                                                 long P, int Co) {
    __shared__ float tile[kPairs * kTaps + 1];
    __shared__ long offs[2 * kTaps];
    const long p0 = (long)blockIdx.x * kPairs;
    const int np = (int)((P - p0) < kPairs ? (P - p0) : kPairs);
    if (threadIdx.x < 2 * kTaps) offs[threadIdx.x] = off[threadIdx.x];
    __syncthreads();
    auto ld_elem = [&](const void* base, long i) -> float {
        if (BF16) return __uint_as_float((uint32_t)reinterpret_cast<const uint16_t*>(base)[i] << 16);
        return reinterpret_cast<const float*>(base)[i];
    };
    for (int i = threadIdx.x; i < kTaps * kPairs; i += 256) {
        const int t = i / kPairs, p = i % kPairs;
        if (p >= np) continue;
        const long pair = p0 + p;
        const long ci = pair / Co, co = pair - ci * Co;
        float v = 0.f;
        const long o0 = offs[2 * t], o1 = offs[2 * t + 1];
        if (o0 >= 0) v += ld_elem(src, o0 + ci * ld + co);
        if (o1 >= 0) v += ld_elem(src, o1 + ci * ld + co);
        if (pb) v += ld_elem(pb, ci) * ld_elem(dv, (long)t * Co + co);
        const int a = t / 25, b = (t / 5) % 5, c = t % 5;
        tile[p * kTaps + ((2 - a) * 5 + (4 - b)) * 5 + (4 - c)] = v;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < np * kTaps; i += 256) dw[p0 * kTaps + i] = tile[i];
}

// bf16 form for even Co / ld: a lane owns TWO adjacent (ci, co) pairs (4-byte loads, 256 B per wavefront and tap), a
// workgroup 128 pairs; the pair -> (ci, co) division is made once per lane.
__global__ __launch_bounds__(256) void k_convt_weight_bwd_blocks2(const uint16_t* __restrict__ src, const long* __restrict__ off,
                                                                  long ld, const uint16_t* __restrict__ pb,
                                                                  const uint16_t* __restrict__ dv, float* __restrict__ dw,
                                                                  long P, int Co) {
    constexpr int kP2 = 2 * kPairs;
    __shared__ float tile[kP2 * kTaps + 1];
    __shared__ long offs[2 * kTaps];
    const long p0 = (long)blockIdx.x * kP2;
    const int np = (int)((P - p0) < kP2 ? (P - p0) : kP2);
    if (threadIdx.x < 2 * kTaps) offs[threadIdx.x] = off[threadIdx.x];
    __syncthreads();
    const int p = 2 * (threadIdx.x & 63);
    if (p < np) {                                         // (P is even: both pairs of a lane exist together)
        const long pair = p0 + p;
        const long ci = pair / Co, co = pair - ci * Co;   // co even, co + 1 < Co
        const long row = ci * ld + co;
        const float b = pb ? __uint_as_float((uint32_t)pb[ci] << 16) : 0.f;
        for (int t = threadIdx.x >> 6; t < kTaps; t += 4) {
            float v0 = 0.f, v1 = 0.f;
            const long o0 = offs[2 * t], o1 = offs[2 * t + 1];
            if (o0 >= 0) {
                const uint32_t u = *reinterpret_cast<const uint32_t*>(src + o0 + row);
                v0 += __uint_as_float(u << 16), v1 += __uint_as_float(u & 0xffff0000u);
            }
            if (o1 >= 0) {
                const uint32_t u = *reinterpret_cast<const uint32_t*>(src + o1 + row);
                v0 += __uint_as_float(u << 16), v1 += __uint_as_float(u & 0xffff0000u);
            }
            if (pb) {
                const uint32_t u = *reinterpret_cast<const uint32_t*>(dv + (long)t * Co + co);
                v0 += b * __uint_as_float(u << 16), v1 += b * __uint_as_float(u & 0xffff0000u);
            }
            const int a = t / 25, bb = (t / 5) % 5, c = t % 5;
            const int f = ((2 - a) * 5 + (4 - bb)) * 5 + (4 - c);
            tile[p * kTaps + f] = v0;
            tile[(p + 1) * kTaps + f] = v1;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < np * kTaps; i += 256) dw[p0 * kTaps + i] = tile[i];
}

// The forward twin of k_convt_weight_bwd_blocks: the ConvTranspose3d weight w [P][75] fp32 goes STRAIGHT into the weight
// matrices the class GEMMs of a z-split lattice layer read -- tap t's [Ci x Co] block is written at up to two places of one
// row-major buffer (element offsets off[t][0..1], -1 = none; row pitch ld): its "lower half" and "upper half" slots in the
// class-stacked [sum K_c, 2 Co] matrix (the same table the backward reads its gradient from).  Replaces the tap tensor
// [75, Ci, Co], a concatenation of it with the constant rows (88 MB each way) and one row gather per parity class.
template <bool BF16>
__global__ __launch_bounds__(256) void k_convt_weight_fwd_blocks(const float* __restrict__ w, const long* __restrict__ off, long ld,
                                                                 void* __restrict__ dst, long P, int Co) {
    constexpr int kP2 = 2 * kPairs;
    __shared__ float tile[kP2 * kTaps + 1];
    __shared__ long offs[2 * kTaps];
    const long p0 = (long)blockIdx.x * kP2;
    const int np = (int)((P - p0) < kP2 ? (P - p0) : kP2);
    if (threadIdx.x < 2 * kTaps) offs[threadIdx.x] = off[threadIdx.x];
    for (int i = threadIdx.x; i < np * kTaps; i += 256) tile[i] = w[p0 * kTaps + i];
    __syncthreads();
    const int p = 2 * (threadIdx.x & 63);
    if (p >= np) return;                                  // (P is even: both pairs of a lane exist together)
    const long pair = p0 + p;
    const long ci = pair / Co, co = pair - ci * Co;       // co even, co + 1 < Co
    const long row = ci * ld + co;
    for (int t = threadIdx.x >> 6; t < kTaps; t += 4) {
        const int a = t / 25, bb = (t / 5) % 5, c = t % 5;
        const int f = ((2 - a) * 5 + (4 - bb)) * 5 + (4 - c);
        const float v0 = tile[p * kTaps + f], v1 = tile[(p + 1) * kTaps + f];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const long o = offs[2 * t + h];
            if (o < 0) continue;
            if (BF16)
                *reinterpret_cast<uint32_t*>(reinterpret_cast<uint16_t*>(dst) + o + row) =
                    (uint32_t)f32_to_bf16(v0) | ((uint32_t)f32_to_bf16(v1) << 16);
            else
                *reinterpret_cast<float2*>(reinterpret_cast<float*>(dst) + o + row) = make_float2(v0, v1);
        }
    }
}

// part[z][b][col] = sum_{ci in slice z} x[ci] * S[rows[b] + ci][col]: a row vector through every [Ci x ncols] block of a
// stacked weight matrix (the previous layer's bias seen through every tap, v = b_prev^T K[t], for all taps at once).
// grid (ncols / 512, blocks, ci slices); a lane owns two adjacent columns; the caller adds the slices up in order (no
// atomics: the result is bitwise reproducible).
template <bool BF16>
__global__ __launch_bounds__(256) void k_blocks_vec_fwd(const void* __restrict__ S, const long* __restrict__ rows, long ld,
                                                        int Ci, int ncols, const float* __restrict__ x, float* __restrict__ vec) {
    const int col = (blockIdx.x * 256 + threadIdx.x) * 2;
    if (col >= ncols) return;
    const int b = blockIdx.y;
    const int per = (Ci + gridDim.z - 1) / gridDim.z;
    const int c0 = blockIdx.z * per, c1 = min(Ci, c0 + per);
    float a0 = 0.f, a1 = 0.f;
    const long base = rows[b] * ld + col;
    for (int ci = c0; ci < c1; ++ci) {
        const float xv = x[ci];
        if (BF16) {
            const uint32_t u = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint16_t*>(S) + base + (long)ci * ld);
            a0 += xv * __uint_as_float(u << 16), a1 += xv * __uint_as_float(u & 0xffff0000u);
        } else {
            const float2 u = *reinterpret_cast<const float2*>(reinterpret_cast<const float*>(S) + base + (long)ci * ld);
            a0 += xv * u.x, a1 += xv * u.y;
        }
    }
    *reinterpret_cast<float2*>(vec + ((long)blockIdx.z * gridDim.y + b) * ncols + col) = make_float2(a0, a1);
}

// adjoint in x: part[b][ci] = sum_col S[rows[b] + ci][col] * dvec[b][col] (the caller adds the blocks up in order).  One
// wavefront per (block, ci) row.
template <bool BF16>
__global__ __launch_bounds__(256) void k_blocks_vec_bwd(const void* __restrict__ S, const long* __restrict__ rows, long ld,
                                                        int Ci, int ncols, int nblocks, const float* __restrict__ dvec,
                                                        float* __restrict__ dx) {
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= (long)nblocks * Ci) return;
    const int b = (int)(r / Ci), ci = (int)(r - (long)b * Ci);
    const long base = (rows[b] + ci) * ld;
    const float* dv = dvec + (long)b * ncols;
    float acc = 0.f;
    for (int col = 2 * (threadIdx.x & 63); col < ncols; col += 128) {
        if (BF16) {
            const uint32_t u = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint16_t*>(S) + base + col);
            acc += __uint_as_float(u << 16) * dv[col] + __uint_as_float(u & 0xffff0000u) * dv[col + 1];
        } else {
            const float2 u = *reinterpret_cast<const float2*>(reinterpret_cast<const float*>(S) + base + col);
            acc += u.x * dv[col] + u.y * dv[col + 1];
        }
    }
#pragma unroll
    for (int o = 32; o; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) dx[r] = acc;
}

// ---- lattice <-> channel-first rows.  One workgroup = one (b, z, y) row of W positions x 128 channels.
namespace {
constexpr int kCh = 128;
// same source layouts as ver_lattice_gather (0 plain, 1 planar, 2 z-split, 3 planar z-split)
template <int LAYOUT>
__device__ __forceinline__ long cl_index(int b, int z, int y, int x, int B, int Z, int H, int W) {
    if (LAYOUT == 1) {
        const int plane = ((y & 1) << 1) | (x & 1);
        return ((((long)plane * B + b) * Z + z) * (H >> 1) + (y >> 1)) * (W >> 1) + (x >> 1);
    }
    if (LAYOUT == 2) return (((((long)b * 2 + (z & 1)) * H + y) * W + x) << 1) + (z >> 1);
    if (LAYOUT == 3) {
        const int plane = ((y & 1) << 1) | (x & 1);
        return ((((((long)plane * B + b) * 2 + (z & 1)) * (H >> 1) + (y >> 1)) * (W >> 1) + (x >> 1)) << 1) + (z >> 1);
    }
    return (((long)b * Z + z) * H + y) * W + x;
}
}  // namespace

// T = 2-byte or 4-byte element.  TO_CF: channels-last -> channel-first rows; else the reverse.
template <typename T, int LAYOUT, bool TO_CF>
__global__ __launch_bounds__(256) void k_lattice_transpose(T* __restrict__ cl, T* __restrict__ cf, long cf_stride,
                                                           int B, int Z, int H, int W, int C) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T* tile = reinterpret_cast<T*>(smem);                  // [kCh][W + 1]
    const int wp = W + 1;
    int r = blockIdx.x;
    const int y = r % H;
    r /= H;
    const int z = r % Z;
    const int b = r / Z;
    const int c0 = blockIdx.y * kCh;
    const int nc = (C - c0) < kCh ? (C - c0) : kCh;
    if (TO_CF) {
        for (int i = threadIdx.x; i < W * kCh; i += 256) {
            const int x = i / kCh, c = i % kCh;
            if (c < nc) tile[c * wp + x] = cl[cl_index<LAYOUT>(b, z, y, x, B, Z, H, W) * C + c0 + c];
        }
        __syncthreads();
        for (int i = threadIdx.x; i < kCh * W; i += 256) {
            const int c = i / W, x = i % W;
            if (c < nc) cf[(long)b * cf_stride + (((long)(c0 + c) * Z + z) * H + y) * W + x] = tile[c * wp + x];
        }
    } else {
        for (int i = threadIdx.x; i < kCh * W; i += 256) {
            const int c = i / W, x = i % W;
            if (c < nc) tile[c * wp + x] = cf[(long)b * cf_stride + (((long)(c0 + c) * Z + z) * H + y) * W + x];
        }
        __syncthreads();
        for (int i = threadIdx.x; i < W * kCh; i += 256) {
            const int x = i / kCh, c = i % kCh;
            if (c < nc) cl[cl_index<LAYOUT>(b, z, y, x, B, Z, H, W) * C + c0 + c] = tile[c * wp + x];
        }
    }
}

// The same with 16-byte accesses on the channels-last side and 8-byte accesses on the channel-first side (W a multiple of
// 8 / sizeof(T) elements, C a multiple of 16 / sizeof(T)).  A tile is CH channels x R rows of W positions: on the
// channel-first side a channel's R W positions are ONE contiguous run (R = 10 rows of 60: 1 200 bytes), so the 8-byte
// stores of a wave fill whole cache lines -- with single rows (120-byte runs, one per channel and workgroup) the
// transposes of the 1.06-GB lattice of the vocc.py step took 3.6 ms / 2.1 ms; element-wise (2 bytes per lane) 4.0 / 3.6 ms.
// Round 3: 16-bit tiles hold dwords of two neighbouring positions (v_perm_b32 instead of 2-byte LDS accesses) and the
// grid runs the channel blocks of one spatial tile next to each other: 2.73 -> 1.90 ms to channel-first, 2.25 -> 1.79 back.
// grid = B * Z * H / R spatial tiles x C / CH channel blocks, channel block fastest.
template <typename T, int LAYOUT, bool TO_CF, int CH>
__global__ __launch_bounds__(256) void k_lattice_transpose_v(T* __restrict__ cl, T* __restrict__ cf, long cf_stride,
                                                             int B, int Z, int H, int W, int C, int R) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int VC = 16 / sizeof(T), VX = 8 / sizeof(T);
    const int P = R * W;
    // channel block fastest: the workgroups that read (write) the pieces of one 128-byte line of a position run next to
    // each other -- with the spatial tile fastest every line of the lattice was fetched from HBM once per piece
    const int ncb = (C + CH - 1) / CH;
    int r = (int)(blockIdx.x / ncb);
    const int hr = H / R;
    const int y0 = (r % hr) * R;
    r /= hr;
    const int z = r % Z;
    const int b = r / Z;
    const int c0 = (int)(blockIdx.x % ncb) * CH;
    const int nc = (C - c0) < CH ? (C - c0) : CH;          // a multiple of VC
    const int ncv = nc / VC, npv = P / VX;
    if constexpr (sizeof(T) == 2) {
        // 16-bit elements, W even: the tile holds DWORDS of two neighbouring positions of one channel, [CH][P / 2 (+ pad)].
        // A thread on the channels-last side handles a PAIR of positions: two 16-byte vectors, eight v_perm_b32, eight
        // 4-byte LDS accesses; the channel-first side moves 8 bytes (four positions) per LDS access.  (Element-wise --
        // one 2-byte LDS access per element, two lanes per bank word -- the 1.06-GB lattice of the vocc.py step took
        // 2.7 ms / 2.25 ms per direction, 0.8-0.9 TB/s.)
        unsigned* tile32 = reinterpret_cast<unsigned*>(smem);
        const int P2 = P >> 1, wp32 = (P2 + 2) & ~1;             // even row pitch: 8-byte aligned rows
        if (TO_CF) {
            for (int i = threadIdx.x; i < P2 * ncv; i += 256) {
                const int p2 = i / ncv, cv = i - p2 * ncv;
                const int p = 2 * p2, yy = p / W, x = p - yy * W;
                const uint4 a = *reinterpret_cast<const uint4*>(cl + cl_index<LAYOUT>(b, z, y0 + yy, x, B, Z, H, W) * C + c0 + cv * VC);
                const uint4 q = *reinterpret_cast<const uint4*>(cl + cl_index<LAYOUT>(b, z, y0 + yy, x + 1, B, Z, H, W) * C + c0 + cv * VC);
                const unsigned aw[4] = {a.x, a.y, a.z, a.w}, qw[4] = {q.x, q.y, q.z, q.w};
                unsigned* dst = tile32 + (cv * VC) * wp32 + p2;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    dst[(2 * k) * wp32] = __builtin_amdgcn_perm(qw[k], aw[k], 0x05040100u);        // channel 2k:   (pos p, pos p + 1)
                    dst[(2 * k + 1) * wp32] = __builtin_amdgcn_perm(qw[k], aw[k], 0x07060302u);    // channel 2k+1
                }
            }
            __syncthreads();
            for (int i = threadIdx.x; i < nc * npv; i += 256) {
                const int c = i / npv, pv = i - c * npv;
                const uint2 u = *reinterpret_cast<const uint2*>(tile32 + c * wp32 + 2 * pv);
                *reinterpret_cast<uint2*>(cf + (long)b * cf_stride + (((long)(c0 + c) * Z + z) * H + y0) * W + pv * VX) = u;
            }
        } else {
            for (int i = threadIdx.x; i < nc * npv; i += 256) {
                const int c = i / npv, pv = i - c * npv;
                *reinterpret_cast<uint2*>(tile32 + c * wp32 + 2 * pv) =
                    *reinterpret_cast<const uint2*>(cf + (long)b * cf_stride + (((long)(c0 + c) * Z + z) * H + y0) * W + pv * VX);
            }
            __syncthreads();
            for (int i = threadIdx.x; i < P2 * ncv; i += 256) {
                const int p2 = i / ncv, cv = i - p2 * ncv;
                const int p = 2 * p2, yy = p / W, x = p - yy * W;
                const unsigned* srcw = tile32 + (cv * VC) * wp32 + p2;
                unsigned d[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) d[j] = srcw[j * wp32];
                uint4 a, q;
                a.x = __builtin_amdgcn_perm(d[1], d[0], 0x05040100u); q.x = __builtin_amdgcn_perm(d[1], d[0], 0x07060302u);
                a.y = __builtin_amdgcn_perm(d[3], d[2], 0x05040100u); q.y = __builtin_amdgcn_perm(d[3], d[2], 0x07060302u);
                a.z = __builtin_amdgcn_perm(d[5], d[4], 0x05040100u); q.z = __builtin_amdgcn_perm(d[5], d[4], 0x07060302u);
                a.w = __builtin_amdgcn_perm(d[7], d[6], 0x05040100u); q.w = __builtin_amdgcn_perm(d[7], d[6], 0x07060302u);
                *reinterpret_cast<uint4*>(cl + cl_index<LAYOUT>(b, z, y0 + yy, x, B, Z, H, W) * C + c0 + cv * VC) = a;
                *reinterpret_cast<uint4*>(cl + cl_index<LAYOUT>(b, z, y0 + yy, x + 1, B, Z, H, W) * C + c0 + cv * VC) = q;
            }
        }
        return;
    }
    T* tile = reinterpret_cast<T*>(smem);                  // [CH][R * W + 1]
    const int wp = P + 1;
    union V16 { uint4 v; T e[VC]; };
    union V8 { uint2 v; T e[VX]; };
    if (TO_CF) {
        for (int i = threadIdx.x; i < P * ncv; i += 256) {
            const int p = i / ncv, cv = i - p * ncv;
            const int yy = p / W, x = p - yy * W;
            V16 u;
            u.v = *reinterpret_cast<const uint4*>(cl + cl_index<LAYOUT>(b, z, y0 + yy, x, B, Z, H, W) * C + c0 + cv * VC);
#pragma unroll
            for (int j = 0; j < VC; ++j) tile[(cv * VC + j) * wp + p] = u.e[j];
        }
        __syncthreads();
        for (int i = threadIdx.x; i < nc * npv; i += 256) {
            const int c = i / npv, pv = i - c * npv;
            V8 u;
#pragma unroll
            for (int k = 0; k < VX; ++k) u.e[k] = tile[c * wp + pv * VX + k];
            *reinterpret_cast<uint2*>(cf + (long)b * cf_stride + (((long)(c0 + c) * Z + z) * H + y0) * W + pv * VX) = u.v;
        }
    } else {
        for (int i = threadIdx.x; i < nc * npv; i += 256) {
            const int c = i / npv, pv = i - c * npv;
            V8 u;
            u.v = *reinterpret_cast<const uint2*>(cf + (long)b * cf_stride + (((long)(c0 + c) * Z + z) * H + y0) * W + pv * VX);
#pragma unroll
            for (int k = 0; k < VX; ++k) tile[c * wp + pv * VX + k] = u.e[k];
        }
        __syncthreads();
        for (int i = threadIdx.x; i < P * ncv; i += 256) {
            const int p = i / ncv, cv = i - p * ncv;
            const int yy = p / W, x = p - yy * W;
            V16 u;
#pragma unroll
            for (int j = 0; j < VC; ++j) u.e[j] = tile[(cv * VC + j) * wp + p];
            *reinterpret_cast<uint4*>(cl + cl_index<LAYOUT>(b, z, y0 + yy, x, B, Z, H, W) * C + c0 + cv * VC) = u.v;
        }
    }
}

// ---- lattice (channels-last) <-> gathered operand rows of occ_proj, in ONE pass (bf16).
// The raw .view of the reference (head:564) makes every operand row R runs of the channel-first lattice, and for the
// geometries of interest the runs tile the flat lattice PERIODICALLY: flat index i = k * quarter + row * period + off, the
// segment [seg_off, seg_off + seg_len) that holds `off` names the pattern group, and the element sits at column
// k * seg_len + (off - seg_off) of that group's row (b * seg_rows + row).  k_lattice_transpose_v's channel-first side is a
// contiguous run per channel; here the same 8-byte pieces go straight to / come straight from the operand rows, so the
// channel-first copy of the lattice (1.06 GB per direction at 192 viewpoints) is never written or read.
namespace {
struct RowMap {
    long quarter;
    int period, nseg;
    int seg_off[8], seg_len[8], seg_pitch[8], seg_rows[8];
    long seg_base[8];
};
// Per workgroup (one sample b, one (z, y0) spatial tile, CH channels): the flat index of a channel's first piece is split
// into (k, row, off) ONCE per channel (two divisions by the first CH lanes, kept in LDS); a piece then adds its position,
// wraps over at most a few periods and looks its segment up in a byte table over the period (4-element slots) -- ~20 VALU
// operations per 8-byte piece.  (The first form divided and walked the segment table per piece: ~100 operations, and
// both directions ran VALU-bound at 3.1 TB/s where ver_lattice_transpose moves 4.5.)
struct RowMapLds {
    int4 ch[128];                         // per channel of the tile: (k, row, off, -) of its first piece
    int4 seg[8];                          // per segment: (seg_off, seg_len, seg_pitch, -)
    long seg_base[8];                     // + b * seg_rows * seg_pitch
    unsigned char slot_seg[1024];         // period / 4 slots -> segment
};
__device__ __forceinline__ void row_map_setup(RowMapLds& t, const RowMap& m, int b, int nc, long flat0, long flat_per_channel) {
    const int tid = threadIdx.x;
    if (tid < nc) {
        const unsigned flat = (unsigned)(flat0 + tid * flat_per_channel);
        const unsigned k = flat / (unsigned)m.quarter;
        const unsigned rem = flat - k * (unsigned)m.quarter;
        const unsigned row = rem / (unsigned)m.period;
        t.ch[tid] = make_int4((int)k, (int)row, (int)(rem - row * (unsigned)m.period), 0);
    }
    if (tid >= 128 && tid < 136) {
        const int j = tid - 128;
        int so = 0, sl = 0, sp = 0, sr = 0;
        long sb = 0;
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (q == j) so = m.seg_off[q], sl = m.seg_len[q], sp = m.seg_pitch[q], sr = m.seg_rows[q], sb = m.seg_base[q];
        t.seg[j] = make_int4(so, sl, sp, 0);
        t.seg_base[j] = sb + (long)b * sr * sp;
    }
    for (int sl4 = tid; sl4 < (m.period >> 2); sl4 += 256) {
        const int off = sl4 << 2;
        int sg = 0;
#pragma unroll
        for (int q = 1; q < 8; ++q)
            if (q < m.nseg && off >= m.seg_off[q]) sg = q;
        t.slot_seg[sl4] = (unsigned char)sg;
    }
}
// four LDS reads per 8-byte piece (channel record, slot byte, segment record, segment base)
__device__ __forceinline__ long row_map_piece(const RowMapLds& t, int period, int c, int piece_off) {
    const int4 ch = t.ch[c];
    int off = ch.z + piece_off, row = ch.y;
    while (off >= period) off -= period, ++row;
    const int sg = t.slot_seg[off >> 2];
    const int4 sp = t.seg[sg];
    return t.seg_base[sg] + (long)row * sp.z + ch.x * sp.y + (off - sp.x);
}
}  // namespace

template <int LAYOUT, bool TO_ROWS, int CH>
__global__ __launch_bounds__(256) void k_lattice_rows(uint16_t* __restrict__ cl, uint16_t* __restrict__ rows, RowMap map,
                                                      int B, int Z, int H, int W, int C, int R) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int VC = 8, VX = 4;
    const int P = R * W;
    const int ncb = (C + CH - 1) / CH;
    int r = (int)(blockIdx.x / ncb);
    const int hr = H / R;
    const int y0 = (r % hr) * R;
    r /= hr;
    const int z = r % Z;
    const int b = r / Z;
    const int c0 = (int)(blockIdx.x % ncb) * CH;
    const int nc = (C - c0) < CH ? (C - c0) : CH;
    const int ncv = nc / VC, npv = P / VX;
    unsigned* tile32 = reinterpret_cast<unsigned*>(smem);        // [CH][P / 2 (+ pad)] dwords of two neighbouring positions
    const int P2 = P >> 1, wp32 = (P2 + 2) & ~1;
    __shared__ RowMapLds lmap;
    row_map_setup(lmap, map, b, nc, (((long)c0 * Z + z) * H + y0) * W, (long)Z * H * W);
    if (TO_ROWS) {
        for (int i = threadIdx.x; i < P2 * ncv; i += 256) {
            const int p2 = i / ncv, cv = i - p2 * ncv;
            const int p = 2 * p2, yy = p / W, x = p - yy * W;
            const uint4 a = *reinterpret_cast<const uint4*>(cl + cl_index<LAYOUT>(b, z, y0 + yy, x, B, Z, H, W) * C + c0 + cv * VC);
            const uint4 q = *reinterpret_cast<const uint4*>(cl + cl_index<LAYOUT>(b, z, y0 + yy, x + 1, B, Z, H, W) * C + c0 + cv * VC);
            const unsigned aw[4] = {a.x, a.y, a.z, a.w}, qw[4] = {q.x, q.y, q.z, q.w};
            unsigned* dst = tile32 + (cv * VC) * wp32 + p2;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                dst[(2 * k) * wp32] = __builtin_amdgcn_perm(qw[k], aw[k], 0x05040100u);
                dst[(2 * k + 1) * wp32] = __builtin_amdgcn_perm(qw[k], aw[k], 0x07060302u);
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < nc * npv; i += 256) {
            const int c = i / npv, pv = i - c * npv;
            const uint2 u = *reinterpret_cast<const uint2*>(tile32 + c * wp32 + 2 * pv);
            *reinterpret_cast<uint2*>(rows + row_map_piece(lmap, map.period, c, pv * VX)) = u;
        }
    } else {
        __syncthreads();                                         // (the row map)
        for (int i = threadIdx.x; i < nc * npv; i += 256) {
            const int c = i / npv, pv = i - c * npv;
            *reinterpret_cast<uint2*>(tile32 + c * wp32 + 2 * pv) = *reinterpret_cast<const uint2*>(rows + row_map_piece(lmap, map.period, c, pv * VX));
        }
        __syncthreads();
        for (int i = threadIdx.x; i < P2 * ncv; i += 256) {
            const int p2 = i / ncv, cv = i - p2 * ncv;
            const int p = 2 * p2, yy = p / W, x = p - yy * W;
            const unsigned* srcw = tile32 + (cv * VC) * wp32 + p2;
            unsigned d[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) d[j] = srcw[j * wp32];
            uint4 a, q;
            a.x = __builtin_amdgcn_perm(d[1], d[0], 0x05040100u); q.x = __builtin_amdgcn_perm(d[1], d[0], 0x07060302u);
            a.y = __builtin_amdgcn_perm(d[3], d[2], 0x05040100u); q.y = __builtin_amdgcn_perm(d[3], d[2], 0x07060302u);
            a.z = __builtin_amdgcn_perm(d[5], d[4], 0x05040100u); q.z = __builtin_amdgcn_perm(d[5], d[4], 0x07060302u);
            a.w = __builtin_amdgcn_perm(d[7], d[6], 0x05040100u); q.w = __builtin_amdgcn_perm(d[7], d[6], 0x07060302u);
            *reinterpret_cast<uint4*>(cl + cl_index<LAYOUT>(b, z, y0 + yy, x, B, Z, H, W) * C + c0 + cv * VC) = a;
            *reinterpret_cast<uint4*>(cl + cl_index<LAYOUT>(b, z, y0 + yy, x + 1, B, Z, H, W) * C + c0 + cv * VC) = q;
        }
    }
}

extern "C" int ver_convt_weight_forward(const float* weight, void* taps, long pairs, int dtype, void* stream) {
    VER_REQUIRE(pairs >= 0, VER_EINVAL, "ver_convt_weight_forward: negative size");
    VER_REQUIRE(dtype == VER_F32 || dtype == VER_BF16, VER_EINVAL, "ver_convt_weight_forward: dtype %d", dtype);
    if (pairs == 0) return VER_OK;
    VER_REQUIRE(weight && taps, VER_EINVAL, "ver_convt_weight_forward: null pointer argument");
    const unsigned blocks = (unsigned)((pairs + kPairs - 1) / kPairs);
    if (dtype == VER_BF16)
        hipLaunchKernelGGL(k_convt_weight_fwd<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, weight, taps, pairs);
    else
        hipLaunchKernelGGL(k_convt_weight_fwd<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, weight, taps, pairs);
    return ver_check_launch("ver_convt_weight_forward");
}

extern "C" int ver_convt_weight_backward(const void* grad_taps, float* grad_weight, long pairs, int dtype, void* stream) {
    VER_REQUIRE(pairs >= 0, VER_EINVAL, "ver_convt_weight_backward: negative size");
    VER_REQUIRE(dtype == VER_F32 || dtype == VER_BF16, VER_EINVAL, "ver_convt_weight_backward: dtype %d", dtype);
    if (pairs == 0) return VER_OK;
    VER_REQUIRE(grad_taps && grad_weight, VER_EINVAL, "ver_convt_weight_backward: null pointer argument");
    const unsigned blocks = (unsigned)((pairs + kPairs - 1) / kPairs);
    if (dtype == VER_BF16)
        hipLaunchKernelGGL(k_convt_weight_bwd<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, grad_taps,
                           grad_weight, pairs);
    else
        hipLaunchKernelGGL(k_convt_weight_bwd<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, grad_taps,
                           grad_weight, pairs);
    return ver_check_launch("ver_convt_weight_backward");
}

extern "C" int ver_convt_weight_backward_blocks(const void* blocks, const long* block_offsets, long ld, const void* prev_bias,
                                               const void* grad_v, float* grad_weight, int ci, int co, int dtype,
                                               void* stream) {
    VER_REQUIRE(ci >= 0 && co >= 0 && ld >= co, VER_EINVAL, "ver_convt_weight_backward_blocks: bad sizes (ci %d co %d ld %ld)",
                ci, co, ld);
    VER_REQUIRE(dtype == VER_F32 || dtype == VER_BF16, VER_EINVAL, "ver_convt_weight_backward_blocks: dtype %d", dtype);
    VER_REQUIRE((prev_bias == nullptr) == (grad_v == nullptr), VER_EINVAL,
                "ver_convt_weight_backward_blocks: prev_bias and grad_v come together");
    const long pairs = (long)ci * co;
    if (pairs == 0) return VER_OK;
    VER_REQUIRE(blocks && block_offsets && grad_weight, VER_EINVAL, "ver_convt_weight_backward_blocks: null pointer argument");
    const unsigned nb = (unsigned)((pairs + kPairs - 1) / kPairs);
    if (dtype == VER_BF16 && co % 2 == 0 && ld % 2 == 0 && ((uintptr_t)blocks & 3) == 0 && ((uintptr_t)grad_v & 3) == 0)
        // (even Co and pitch: every block offset the layers hand over -- row * ld + {0, co} -- is even as well)
        hipLaunchKernelGGL(k_convt_weight_bwd_blocks2, dim3((unsigned)((pairs + 2 * kPairs - 1) / (2 * kPairs))), dim3(256), 0,
                           (hipStream_t)stream, (const uint16_t*)blocks, block_offsets, ld, (const uint16_t*)prev_bias,
                           (const uint16_t*)grad_v, grad_weight, pairs, co);
    else if (dtype == VER_BF16)
        hipLaunchKernelGGL(k_convt_weight_bwd_blocks<true>, dim3(nb), dim3(256), 0, (hipStream_t)stream, blocks, block_offsets,
                           ld, prev_bias, grad_v, grad_weight, pairs, co);
    else
        hipLaunchKernelGGL(k_convt_weight_bwd_blocks<false>, dim3(nb), dim3(256), 0, (hipStream_t)stream, blocks, block_offsets,
                           ld, prev_bias, grad_v, grad_weight, pairs, co);
    return ver_check_launch("ver_convt_weight_backward_blocks");
}

extern "C" int ver_convt_weight_forward_blocks(const float* weight, const long* block_offsets, long ld, void* blocks, int ci,
                                              int co, int dtype, void* stream) {
    VER_REQUIRE(ci >= 0 && co >= 0 && ld >= co, VER_EINVAL, "ver_convt_weight_forward_blocks: bad sizes (ci %d co %d ld %ld)", ci, co, ld);
    VER_REQUIRE(dtype == VER_F32 || dtype == VER_BF16, VER_EINVAL, "ver_convt_weight_forward_blocks: dtype %d", dtype);
    const long pairs = (long)ci * co;
    if (pairs == 0) return VER_OK;
    VER_REQUIRE(weight && block_offsets && blocks, VER_EINVAL, "ver_convt_weight_forward_blocks: null pointer argument");
    VER_REQUIRE(co % 2 == 0 && ld % 2 == 0 && ((uintptr_t)blocks & 7) == 0, VER_EUNSUPPORTED,
                "ver_convt_weight_forward_blocks: even co and ld and an 8-byte aligned destination (pairs of adjacent columns are "
                "written together; block offsets must be even too)");
    const unsigned nb = (unsigned)((pairs + 2 * kPairs - 1) / (2 * kPairs));
    if (dtype == VER_BF16)
        hipLaunchKernelGGL(k_convt_weight_fwd_blocks<true>, dim3(nb), dim3(256), 0, (hipStream_t)stream, weight, block_offsets, ld,
                           blocks, pairs, co);
    else
        hipLaunchKernelGGL(k_convt_weight_fwd_blocks<false>, dim3(nb), dim3(256), 0, (hipStream_t)stream, weight, block_offsets, ld,
                           blocks, pairs, co);
    return ver_check_launch("ver_convt_weight_forward_blocks");
}

extern "C" int ver_blocks_vec_forward(const void* blocks, const long* block_rows, int nblocks, long ld, int ci, int ncols,
                                      const float* x, float* vec, int dtype, void* stream) {
    VER_REQUIRE(nblocks >= 0 && ci >= 0 && ncols >= 0 && ld >= ncols, VER_EINVAL, "ver_blocks_vec_forward: bad sizes");
    VER_REQUIRE(dtype == VER_F32 || dtype == VER_BF16, VER_EINVAL, "ver_blocks_vec_forward: dtype %d", dtype);
    if (nblocks == 0 || ncols == 0) return VER_OK;
    VER_REQUIRE(blocks && block_rows && x && vec, VER_EINVAL, "ver_blocks_vec_forward: null pointer argument");
    VER_REQUIRE(ncols % 2 == 0 && ld % 2 == 0 && ((uintptr_t)blocks & 7) == 0 && nblocks <= 65535, VER_EUNSUPPORTED,
                "ver_blocks_vec_forward: even ncols / ld, an 8-byte aligned matrix and at most 65 535 blocks");
    hipStream_t st = (hipStream_t)stream;
    const int slices = VER_BLOCKS_VEC_SLICES;
    const dim3 grid((unsigned)((ncols / 2 + 255) / 256), (unsigned)nblocks, (unsigned)slices);
    if (dtype == VER_BF16)
        hipLaunchKernelGGL(k_blocks_vec_fwd<true>, grid, dim3(256), 0, st, blocks, block_rows, ld, ci, ncols, x, vec);
    else
        hipLaunchKernelGGL(k_blocks_vec_fwd<false>, grid, dim3(256), 0, st, blocks, block_rows, ld, ci, ncols, x, vec);
    return ver_check_launch("ver_blocks_vec_forward");
}

extern "C" int ver_blocks_vec_backward(const void* blocks, const long* block_rows, int nblocks, long ld, int ci, int ncols,
                                       const float* grad_vec, float* grad_x, int dtype, void* stream) {
    VER_REQUIRE(nblocks >= 0 && ci >= 0 && ncols >= 0 && ld >= ncols, VER_EINVAL, "ver_blocks_vec_backward: bad sizes");
    VER_REQUIRE(dtype == VER_F32 || dtype == VER_BF16, VER_EINVAL, "ver_blocks_vec_backward: dtype %d", dtype);
    if (ci == 0 || nblocks == 0) return VER_OK;
    VER_REQUIRE(blocks && block_rows && grad_vec && grad_x, VER_EINVAL, "ver_blocks_vec_backward: null pointer argument");
    hipStream_t st = (hipStream_t)stream;
    VER_REQUIRE(ncols % 2 == 0 && ld % 2 == 0 && ((uintptr_t)blocks & 7) == 0, VER_EUNSUPPORTED,
                "ver_blocks_vec_backward: even ncols / ld and an 8-byte aligned matrix");
    const long rows = (long)nblocks * ci;
    const unsigned grid = (unsigned)((rows + 3) / 4);
    if (dtype == VER_BF16)
        hipLaunchKernelGGL(k_blocks_vec_bwd<true>, dim3(grid), dim3(256), 0, st, blocks, block_rows, ld, ci, ncols, nblocks, grad_vec, grad_x);
    else
        hipLaunchKernelGGL(k_blocks_vec_bwd<false>, dim3(grid), dim3(256), 0, st, blocks, block_rows, ld, ci, ncols, nblocks, grad_vec, grad_x);
    return ver_check_launch("ver_blocks_vec_backward");
}

extern "C" int ver_lattice_transpose(void* channels_last, void* channel_first, long cf_stride, int B, int Z, int H, int W,
                                     int C, int layout, int to_channel_first, int dtype, void* stream) {
    VER_REQUIRE(B >= 0 && Z > 0 && H > 0 && W > 0 && C > 0, VER_EINVAL, "ver_lattice_transpose: bad sizes");
    VER_REQUIRE(dtype == VER_F32 || dtype == VER_BF16, VER_EINVAL, "ver_lattice_transpose: dtype %d", dtype);
    VER_REQUIRE(layout >= 0 && layout <= 3, VER_EINVAL, "ver_lattice_transpose: layout %d", layout);
    VER_REQUIRE(!(layout & 1) || (H % 2 == 0 && W % 2 == 0), VER_EINVAL, "ver_lattice_transpose: planar needs even H, W");
    VER_REQUIRE(layout < 2 || Z == 4, VER_EUNSUPPORTED, "ver_lattice_transpose: the z-split layouts are built for 4 z-layers");
    VER_REQUIRE(cf_stride >= (long)C * Z * H * W, VER_EINVAL, "ver_lattice_transpose: row stride too small");
    if (B == 0) return VER_OK;
    VER_REQUIRE(channels_last && channel_first, VER_EINVAL, "ver_lattice_transpose: null pointer argument");
    const size_t esize = dtype == VER_BF16 ? 2 : 4;
    hipStream_t st = (hipStream_t)stream;
    // vector form: whole 16-byte channel vectors / 8-byte position vectors and buffers aligned for them; tiles of
    // 32 channels x R rows, R the largest divisor of H whose tile stays under 48 KB (and under ~1 200 positions)
    constexpr int kChV = 32;
    const int vc = 16 / (int)esize, vx = 8 / (int)esize;
    const bool vec = W % vx == 0 && C % vc == 0 && cf_stride % vx == 0 && ((uintptr_t)channels_last & 15) == 0 &&
                     ((uintptr_t)channel_first & 7) == 0 && ((long)Z * H * W) % vx == 0;
    // (the guard is the vector kernels' own LDS request with R = 1: kCh x (W + 4) elements, under the 64 KB a launch gets
    // without hipFuncSetAttribute; wider rows take the scalar kernel)
    if (vec && (size_t)kCh * ((size_t)W + 4) * esize <= 64 * 1024) {
        // the WRITE side wants long runs: to channel-first 32 channels x R rows (R W contiguous positions per channel),
        // to channels-last 128 channels x one row (256 contiguous bytes per position); measured the other way round
        // each direction loses a third (2.7 vs 3.6 ms to channel-first, 2.1 vs 3.0 ms back)
        int R = 1;       // (to channels-last with 3 rows of 128 channels: 2.64 ms against 1.79 -- fewer workgroups per CU)
        if (to_channel_first)
            for (int cand = 1; cand <= H; ++cand)
                if (H % cand == 0 && (size_t)kChV * ((size_t)cand * W + 4) * esize <= 48 * 1024 && cand * W <= 1200) R = cand;
        const int chv = to_channel_first ? kChV : kCh;
        const size_t ldsv = (size_t)chv * ((size_t)R * W + 4) * esize;     // (+ row padding: 16-bit tiles use an even dword pitch)
        const dim3 gridv((unsigned)((long)B * Z * (H / R) * ((C + chv - 1) / chv)));
#define VER_TRV(T, L, CF)                                                                                              \
    hipLaunchKernelGGL((k_lattice_transpose_v<T, L, CF, (CF ? kChV : kCh)>), gridv, dim3(256), ldsv, st,                \
                       (T*)channels_last, (T*)channel_first, cf_stride, B, Z, H, W, C, R)
#define VER_TRV_L(T, CF)                          \
    do {                                          \
        if (layout == 0) VER_TRV(T, 0, CF);       \
        else if (layout == 1) VER_TRV(T, 1, CF);  \
        else if (layout == 2) VER_TRV(T, 2, CF);  \
        else VER_TRV(T, 3, CF);                   \
    } while (0)
        if (dtype == VER_BF16) {
            if (to_channel_first) VER_TRV_L(uint16_t, true); else VER_TRV_L(uint16_t, false);
        } else {
            if (to_channel_first) VER_TRV_L(float, true); else VER_TRV_L(float, false);
        }
#undef VER_TRV_L
#undef VER_TRV
        return ver_check_launch("ver_lattice_transpose");
    }
    const size_t lds = (size_t)kCh * (W + 1) * esize;
    VER_REQUIRE(lds <= 64 * 1024, VER_EUNSUPPORTED, "ver_lattice_transpose: W = %d too wide", W);
    const dim3 grid((unsigned)((long)B * Z * H), (unsigned)((C + kCh - 1) / kCh));
#define VER_TR(T, L, CF)                                                                                             \
    hipLaunchKernelGGL((k_lattice_transpose<T, L, CF>), grid, dim3(256), lds, st, (T*)channels_last, (T*)channel_first, \
                       cf_stride, B, Z, H, W, C)
#define VER_TR_L(T, CF)                  \
    do {                                 \
        if (layout == 0) VER_TR(T, 0, CF);      \
        else if (layout == 1) VER_TR(T, 1, CF); \
        else if (layout == 2) VER_TR(T, 2, CF); \
        else VER_TR(T, 3, CF);                  \
    } while (0)
    if (dtype == VER_BF16) {
        if (to_channel_first) VER_TR_L(uint16_t, true); else VER_TR_L(uint16_t, false);
    } else {
        if (to_channel_first) VER_TR_L(float, true); else VER_TR_L(float, false);
    }
#undef VER_TR_L
#undef VER_TR
    return ver_check_launch("ver_lattice_transpose");
}

extern "C" int ver_lattice_rows(void* channels_last, void* rows, long quarter, int period, int nseg, const int* seg_off,
                                const int* seg_len, const long* seg_base, const int* seg_pitch, const int* seg_rows, int B,
                                int Z, int H, int W, int C, int layout, int to_rows, int dtype, void* stream) {
    VER_REQUIRE(B >= 0 && Z > 0 && H > 0 && W > 0 && C > 0, VER_EINVAL, "ver_lattice_rows: bad sizes");
    VER_REQUIRE(dtype == VER_BF16, VER_EUNSUPPORTED, "ver_lattice_rows: bf16 lattices only (dtype %d)", dtype);
    VER_REQUIRE(layout >= 0 && layout <= 3, VER_EINVAL, "ver_lattice_rows: layout %d", layout);
    VER_REQUIRE(!(layout & 1) || (H % 2 == 0 && W % 2 == 0), VER_EINVAL, "ver_lattice_rows: planar needs even H, W");
    VER_REQUIRE(layout < 2 || Z == 4, VER_EUNSUPPORTED, "ver_lattice_rows: the z-split layouts are built for 4 z-layers");
    VER_REQUIRE(nseg >= 1 && nseg <= 8 && seg_off && seg_len && seg_base && seg_pitch && seg_rows, VER_EINVAL,
                "ver_lattice_rows: 1..8 segments with their tables");
    const long L = (long)C * Z * H * W;
    VER_REQUIRE(L < (1L << 31) && quarter > 0 && period > 0 && L % quarter == 0 && quarter % period == 0, VER_EINVAL,
                "ver_lattice_rows: lattice of %ld elements, quarter %ld, period %d", L, quarter, period);
    VER_REQUIRE(W % 4 == 0 && C % 8 == 0 && ((long)Z * H * W) % 4 == 0, VER_EUNSUPPORTED,
                "ver_lattice_rows: W %% 4, C %% 8 (8-byte pieces along W, 16-byte channel vectors)");
    VER_REQUIRE(period <= 4096 && quarter % ((long)H * W) == 0, VER_EUNSUPPORTED,
                "ver_lattice_rows: period %d > 4096 or a (c, z) plane of %d elements straddles two quarters", period, H * W);
    RowMap m;
    m.quarter = quarter, m.period = period, m.nseg = nseg;
    int covered = 0;
    for (int j = 0; j < 8; ++j) {
        const bool on = j < nseg;
        m.seg_off[j] = on ? seg_off[j] : 0, m.seg_len[j] = on ? seg_len[j] : 0, m.seg_pitch[j] = on ? seg_pitch[j] : 0;
        m.seg_rows[j] = on ? seg_rows[j] : 0, m.seg_base[j] = on ? seg_base[j] : 0;
        if (!on) continue;
        VER_REQUIRE(seg_off[j] == covered && seg_len[j] > 0 && seg_len[j] % 4 == 0 && seg_pitch[j] % 4 == 0 && seg_base[j] % 4 == 0 &&
                        seg_base[j] >= 0 && (long)seg_rows[j] * period == quarter && (L / quarter) * seg_len[j] <= seg_pitch[j],
                    VER_EINVAL, "ver_lattice_rows: segment %d (offset %d, length %d, pitch %d, rows %d)", j, seg_off[j], seg_len[j],
                    seg_pitch[j], seg_rows[j]);
        covered += seg_len[j];
    }
    VER_REQUIRE(covered == period, VER_EINVAL, "ver_lattice_rows: the segments cover %d of the period %d", covered, period);
    if (B == 0) return VER_OK;
    VER_REQUIRE(channels_last && rows, VER_EINVAL, "ver_lattice_rows: null pointer argument");
    VER_REQUIRE(((uintptr_t)channels_last & 15) == 0 && ((uintptr_t)rows & 7) == 0, VER_EINVAL, "ver_lattice_rows: alignment");
    constexpr int kChV = 32;
    VER_REQUIRE((size_t)kCh * ((size_t)W + 4) * 2 <= 64 * 1024, VER_EUNSUPPORTED, "ver_lattice_rows: W = %d too wide", W);
    int R = 1;                                   // (tile shapes as in ver_lattice_transpose: long runs on the WRITE side)
    if (to_rows)
        for (int cand = 1; cand <= H; ++cand)
            if (H % cand == 0 && (size_t)kChV * ((size_t)cand * W + 4) * 2 <= 48 * 1024 && cand * W <= 1200) R = cand;
    const int chv = to_rows ? kChV : kCh;
    const size_t lds = (size_t)chv * ((size_t)R * W + 4) * 2;
    const dim3 grid((unsigned)((long)B * Z * (H / R) * ((C + chv - 1) / chv)));
    hipStream_t st = (hipStream_t)stream;
#define VER_LR(L_, TR)                                                                                                       \
    hipLaunchKernelGGL((k_lattice_rows<L_, TR, (TR ? kChV : kCh)>), grid, dim3(256), lds, st, (uint16_t*)channels_last,       \
                       (uint16_t*)rows, m, B, Z, H, W, C, R)
#define VER_LR_L(TR)                         \
    do {                                     \
        if (layout == 0) VER_LR(0, TR);      \
        else if (layout == 1) VER_LR(1, TR); \
        else if (layout == 2) VER_LR(2, TR); \
        else VER_LR(3, TR);                  \
    } while (0)
    if (to_rows) VER_LR_L(true); else VER_LR_L(false);
#undef VER_LR_L
#undef VER_LR
    return ver_check_launch("ver_lattice_rows");
}


// ------------------------------------------------------------------------------------------
// Run copies for the gathered `occ_proj` operand (dense_heads/occ_proj_lattice.py).  A row of the operand is R
// contiguous runs of the channel-first lattice (one per token of the reference's raw .view, head:564) plus a few
// augmentation columns: gathered element by element through an index table (torch.index_select) this was 10.7 ms
// per step forward and 5.8 ms backward (index_copy_); as run copies of 8 bytes per lane it is two streaming passes.
//   gather : dst[(b*n_rows + i)*row_elems + k*run_len + o] = src[b*src_stride + run_start[i*R + k] + o]
//            dst[(b*n_rows + i)*row_elems + R*run_len + j] = src[b*src_stride + aug_idx[i*n_aug + j]]
//   scatter: dst[b*dst_stride + run_start[i*R + k] + o]    = src[(b*n_rows + i)*row_elems + k*run_len + o]
// V = 8-byte vector: 4 bf16 or 2 fp32 elements (VE per vector); run_len, run starts and strides are multiples of VE.
template <bool GATHER, typename E>
__global__ __launch_bounds__(256) void k_run_copy(const E* __restrict__ src, E* __restrict__ dst, long img_stride,
                                                  const int* __restrict__ run_start, const int* __restrict__ aug_idx,
                                                  long rows_total, int n_rows, int R, int run_len, int n_aug,
                                                  int row_elems) {
    constexpr int VE = 8 / sizeof(E);
    const int vec_per_run = run_len / VE;
    const int n_vec = R * vec_per_run;
    const int per_row = n_vec + (GATHER ? n_aug : 0);
    typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
    if (per_row <= 256) {
        // a thread owns ONE vector position of the row (its run k and offset o never change) and walks rows: per copy one
        // 4-byte table load and a few adds.  (With the (row, vector) pairs as one flat index space every 8-byte copy paid
        // two 64-bit divisions and a 32-bit one: 0.55 ms per 0.9-GB operand.)  The gathered operand is written once and
        // read by a GEMM much later: nontemporal stores.
        const int v = threadIdx.x;
        if (v >= per_row) return;
        const bool is_vec = v < n_vec;
        const int k = is_vec ? v / vec_per_run : 0, o = is_vec ? v - k * vec_per_run : 0;
        const int j = v - n_vec;
        // four rows per trip: table loads, source loads and stores of four independent rows in flight together
        constexpr int U = 4;
        for (long row0 = (long)blockIdx.x * U; row0 < rows_total; row0 += (long)gridDim.x * U) {
            long img[U], buf[U];
            int tab[U];
            bool live[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const long row = min(row0 + u, rows_total - 1);
                live[u] = row0 + u < rows_total;
                const long b = row / n_rows;
                const int i = (int)(row - b * n_rows);
                tab[u] = is_vec ? run_start[i * R + k] : aug_idx[i * n_aug + j];
                img[u] = b * img_stride + (is_vec ? (long)o * VE : 0);
                buf[u] = row * row_elems + (is_vec ? (long)v * VE : (long)n_vec * VE + j);
            }
            if (is_vec) {
                uint2 val[U];
#pragma unroll
                for (int u = 0; u < U; ++u)
                    val[u] = *reinterpret_cast<const uint2*>(GATHER ? src + img[u] + tab[u] : src + buf[u]);
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (!live[u]) continue;
                    if (GATHER)
                        __builtin_nontemporal_store(u32x2_t{val[u].x, val[u].y}, reinterpret_cast<u32x2_t*>(dst + buf[u]));
                    else
                        *reinterpret_cast<uint2*>(dst + img[u] + tab[u]) = val[u];
                }
            } else {
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (live[u]) dst[buf[u]] = src[img[u] + tab[u]];
            }
        }
        return;
    }
    // rows wider than a workgroup: the (row, vector) pairs as one index space
    const long total = rows_total * per_row;
#pragma unroll 4
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const long row = idx / per_row;
        const int v = (int)(idx - row * per_row);
        const long b = row / n_rows;
        const int i = (int)(row - b * n_rows);
        if (v < n_vec) {
            const int k = v / vec_per_run, o = v - k * vec_per_run;
            const long img = b * img_stride + run_start[i * R + k] + (long)o * VE;
            const long buf = row * row_elems + (long)v * VE;
            if (GATHER)
                *reinterpret_cast<uint2*>(dst + buf) = *reinterpret_cast<const uint2*>(src + img);
            else
                *reinterpret_cast<uint2*>(dst + img) = *reinterpret_cast<const uint2*>(src + buf);
        } else {
            const int j = v - n_vec;
            dst[row * row_elems + (long)n_vec * VE + j] = src[b * img_stride + aug_idx[i * n_aug + j]];
        }
    }
}

namespace {
int run_copy(bool gather, const void* src, void* dst, long img_stride, const int32_t* run_start, const int32_t* aug_idx,
             int B, int n_rows, int R, int run_len, int n_aug, int row_elems, int dtype, void* stream, const char* who) {
    VER_REQUIRE(dtype == VER_F32 || dtype == VER_BF16, VER_EINVAL, "%s: dtype %d is neither VER_F32 nor VER_BF16", who, dtype);
    VER_REQUIRE(B >= 0 && n_rows >= 0 && R > 0 && run_len > 0 && n_aug >= 0, VER_EINVAL, "%s: bad sizes", who);
    const int ve = dtype == VER_BF16 ? 4 : 2;
    VER_REQUIRE(run_len % ve == 0 && img_stride % ve == 0 && row_elems % ve == 0, VER_EUNSUPPORTED,
                "%s: run length / strides must be multiples of %d elements (8 bytes)", who, ve);
    VER_REQUIRE(row_elems >= R * run_len + (gather ? n_aug : 0), VER_EINVAL, "%s: buffer row shorter than its columns", who);
    if (B == 0 || n_rows == 0) return VER_OK;
    VER_REQUIRE(src && dst && run_start && (aug_idx || !gather || n_aug == 0), VER_EINVAL, "%s: null pointer argument", who);
    VER_REQUIRE(((uintptr_t)src & 7) == 0 && ((uintptr_t)dst & 7) == 0, VER_EINVAL, "%s: buffers must be 8-byte aligned", who);
    const long rows_total = (long)B * n_rows;
    const int per_row = R * (run_len / ve) + (gather ? n_aug : 0);
    const long want = per_row <= 256 ? (rows_total + 3) / 4 : (rows_total * per_row + 1023) / 1024;
    const unsigned grid = (unsigned)(want < 16384 ? (want > 0 ? want : 1) : 16384);
    hipStream_t st = (hipStream_t)stream;
#define VER_RUN(G, E)                                                                                         \
    hipLaunchKernelGGL((k_run_copy<G, E>), dim3(grid), dim3(256), 0, st, (const E*)src, (E*)dst, img_stride,   \
                       run_start, aug_idx, rows_total, n_rows, R, run_len, n_aug, row_elems)
    if (dtype == VER_BF16) {
        if (gather) VER_RUN(true, uint16_t); else VER_RUN(false, uint16_t);
    } else {
        if (gather) VER_RUN(true, float); else VER_RUN(false, float);
    }
#undef VER_RUN
    return ver_check_launch(who);
}
}  // namespace

extern "C" int ver_run_gather(const void* image, long image_stride, const int32_t* run_start, const int32_t* aug_idx,
                              void* rows, int B, int n_rows, int runs, int run_len, int n_aug, int row_elems, int dtype,
                              void* stream) {
    return run_copy(true, image, rows, image_stride, run_start, aug_idx, B, n_rows, runs, run_len, n_aug, row_elems, dtype,
                    stream, "ver_run_gather");
}

extern "C" int ver_run_scatter(const void* rows, void* image, long image_stride, const int32_t* run_start, int B,
                               int n_rows, int runs, int run_len, int row_elems, int dtype, void* stream) {
    return run_copy(false, rows, image, image_stride, run_start, nullptr, B, n_rows, runs, run_len, 0, row_elems, dtype,
                    stream, "ver_run_scatter");
}
