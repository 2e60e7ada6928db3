// Fused occupancy MLP (`occ_branches`, dense_heads/voxelformer_occupancy_head.py:241-248, applied at
// :580 to 504 000 voxels per viewpoint):
//
//     x[N,128] -> Linear(128,128) -> LayerNorm -> ReLU -> Linear(128,128) -> LayerNorm -> ReLU
//              -> Linear(128,16) -> logits[N,16]
//
// As separate GEMM / LayerNorm kernels every stage is one HBM round trip over [N,128] (N = 16e6 rows
// for 32 viewpoints: 37 GB forward, 70 GB backward).  Here a wave keeps a block of rows in registers
// through the whole chain: forward reads x once and writes the logits; backward re-computes the
// forward from x, runs the chain backwards and writes d(x) plus the four bf16 tensors the
// row-reduced weight gradients are formed from (include/ver_ops.h).
//
// MFMA formulation (v_mfma_f32_16x16x32_bf16; lane l: c = l & 15, g = l >> 4):
//   A[m][k]: lane holds A[c][8g..8g+7]     B[k][n]: lane holds B[8g..8g+7][c]
//   D[m][n]: lane holds D[4g..4g+3][c]
// Every layer is evaluated TRANSPOSED, T^T[o][r] = sum_k W[o][k] X[r][k], i.e. A = weights,
// B = activations with the row r on the lane's column index c.  Then
//   * a lane owns ONE row r (c) and 32 of its 128 features o = 16*ot + 4g + i: LayerNorm statistics
//     are an in-lane sum + two cross-lane adds (xor 16, 32);
//   * the D registers of tiles (2t, 2t+1) are, as they stand, the B fragment of k-step t of the next
//     layer if that layer's weights are stored with the matching k order
//     kperm(t, g, j) = 32t + (j < 4 ? 4g + j : 16 + 4g + j - 4):
//     the chain never leaves the registers (no LDS transposes, no barriers).
// Weight fragments live in LDS in fragment order (one conflict-free ds_read_b128 per fragment),
// built once per step by `ver_occ_mlp_pack` from the fp32 parameters.
#include <cstdlib>
#include "ver_common.h"

namespace {
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kW = 128;       // hidden width
constexpr int kC = 16;        // classes
constexpr int kFrag = 64 * 8; // bf16 elements of one fragment image (64 lanes x 8)

// image sections (in fragments of 1 KiB)
constexpr int kF1 = 0;            // layer-1 forward  [ot 8][kt 4], natural k
constexpr int kF2 = 32;           // layer-2 forward  [ot 8][kt 4], kperm
constexpr int kF3 = 64;           // layer-3 forward  [kt 4], kperm
constexpr int kFwdFrags = 68;
constexpr int kB3 = 68;           // layer-3 dgrad    [mt 8]        (K = 16 classes, upper half zero)
constexpr int kB2 = 76;           // layer-2 dgrad    [mt 8][ks 4]  (m = input feature, k = o via kperm)
constexpr int kB1 = 108;          // layer-1 dgrad    [mp 4][t 2][ks 4], rows ordered for 16-byte stores
constexpr int kAllFrags = 140;
// fp32 vectors: b1 g1 be1 b2 g2 be2 (128 each) b3 (16)
constexpr int kVecFloats = 6 * kW + kC;

__device__ __forceinline__ int kperm(int t, int g, int j) { return 32 * t + (j < 4 ? 4 * g + j : 16 + 4 * g + j - 4); }

__device__ __forceinline__ f32x4 mfma(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ float xor16(float v) { return __shfl_xor(v, 16, 64); }
__device__ __forceinline__ float xor32(float v) { return __shfl_xor(v, 32, 64); }

// fp32 -> bf16 (round to nearest even) as PAIRS: the compiler splits a 4-wide conversion into single-value
// v_cvt_pk_bf16_f32 (second source zero) and merges the halves with v_perm_b32 -- three instructions per two values; a 2-wide
// conversion is the one packed instruction
__device__ __forceinline__ unsigned pack2(float a, float b) {
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    const f32x2_t v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ bf16x4 pack4(f32x4 v) {
    typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
    const u32x2_t r = {pack2(v.x, v.y), pack2(v.z, v.w)};
    return __builtin_bit_cast(bf16x4, r);
}
__device__ __forceinline__ bf16x8 pack8(f32x4 lo, f32x4 hi) {
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    const u32x4_t r = {pack2(lo.x, lo.y), pack2(lo.z, lo.w), pack2(hi.x, hi.y), pack2(hi.z, hi.w)};
    return __builtin_bit_cast(bf16x8, r);
}
// ReLU AFTER the bf16 pack: a bf16 bit pattern orders like an int16 on its sign, so max(x, 0) on the packed pairs is one
// v_pk_max_i16 per TWO values (gfx950 has no packed fp32 max: the fp32 form is one v_max_f32 per value).  -0.0 and negative
// NaNs become +0.0, positive NaNs pass -- as fmaxf(x, 0) would not, but a NaN row is garbage either way.
__device__ __forceinline__ bf16x8 relu_packed(bf16x8 v) {
    typedef short s16x8_t __attribute__((ext_vector_type(8)));
    const s16x8_t z = {0, 0, 0, 0, 0, 0, 0, 0};
    return __builtin_bit_cast(bf16x8, __builtin_elementwise_max(__builtin_bit_cast(s16x8_t, v), z));
}
}  // namespace

// -------------------------------------------------------------------------------------------- pack
__global__ __launch_bounds__(256) void k_occ_mlp_pack(const float* __restrict__ W1, const float* __restrict__ W2,
                                                      const float* __restrict__ W3, __bf16* __restrict__ img) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= kAllFrags * kFrag) return;
    const int f = e / kFrag, lane = (e / 8) & 63, j = e & 7;
    const int c = lane & 15, g = lane >> 4;
    float v = 0.0f;
    if (f < kF2) {                                   // W1[o][k], natural k
        const int ot = f / 4, kt = f % 4;
        v = W1[(16 * ot + c) * kW + 32 * kt + 8 * g + j];
    } else if (f < kF3) {
        const int ot = (f - kF2) / 4, kt = (f - kF2) % 4;
        v = W2[(16 * ot + c) * kW + kperm(kt, g, j)];
    } else if (f < kB3) {
        const int kt = f - kF3;
        v = W3[c * kW + kperm(kt, g, j)];
    } else if (f < kB2) {                            // A[m = feature][k = class 8g+j], classes >= 16 are padding
        const int mt = f - kB3;
        v = g < 2 ? W3[(8 * g + j) * kW + 16 * mt + c] : 0.0f;
    } else if (f < kB1) {                            // A[m = input feature][k = o (kperm)] = W2[o][m]
        const int mt = (f - kB2) / 4, ks = (f - kB2) % 4;
        v = W2[kperm(ks, g, j) * kW + 16 * mt + c];
    } else {                                         // rows of tile (mp, t): m = 4q+i -> feature 32mp + 8q + 4t + i
        const int mp = (f - kB1) / 8, t = ((f - kB1) / 4) % 2, ks = (f - kB1) % 4;
        const int feat = 32 * mp + 8 * (c >> 2) + 4 * t + (c & 3);
        v = W1[kperm(ks, g, j) * kW + feat];
    }
    img[e] = (__bf16)v;
}

// ----------------------------------------------------------------------------------------- forward
namespace {
// LayerNorm + ReLU of one row tile held transposed: acc[ot][i] = feature 16ot + 4g + i of row c.
// Returns the B fragments of the next layer (k-step t <- tiles 2t, 2t+1); optionally the normalised
// values (for the backward pass).
// CENTERED: the caller guarantees rows with zero mean over the 128 features (the weights / bias of the producing Linear were
// centred over their OUTPUT axis, W <- W - mean_o W, b <- b - mean(b): LayerNorm is invariant to a per-row constant, so
// LN(Wx + b) = LN(PWx + Pb) with P = I - 11^T/128) -- the mean pass and its two cross-lane reductions drop out.
template <bool KEEP, bool CENTERED = false>
__device__ __forceinline__ void ln_relu_tile(f32x4 (&acc)[8], const float* gam, const float* bet, float eps,
                                             bf16x8 (&out)[4], float& rstd_out) {   // gam/bet already offset by 4g
    // (measured and dropped: starting the MFMA chains from 0 and adding the bias here -- the compiler ties vdst to src C,
    //  so the zeroes cost the same v_mov per register the bias did, and the extra live bias vectors spilled)
    if constexpr (!CENTERED) {
        float s = 0.0f;
#pragma unroll
        for (int ot = 0; ot < 8; ++ot) s += (acc[ot].x + acc[ot].y) + (acc[ot].z + acc[ot].w);
        s += xor16(s);
        s += xor32(s);
        const float mu = s * (1.0f / kW);
#pragma unroll
        for (int ot = 0; ot < 8; ++ot) acc[ot] -= mu;
    }
    float q = 0.0f;
#pragma unroll
    for (int ot = 0; ot < 8; ++ot)
        q += (acc[ot].x * acc[ot].x + acc[ot].y * acc[ot].y) + (acc[ot].z * acc[ot].z + acc[ot].w * acc[ot].w);
    q += xor16(q);
    q += xor32(q);
    const float rs = rsqrtf(q * (1.0f / kW) + eps);
    rstd_out = rs;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        f32x4 y[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int ot = 2 * t + h;
            const f32x4 gm = *reinterpret_cast<const f32x4*>(gam + 16 * ot);
            const f32x4 bt = *reinterpret_cast<const f32x4*>(bet + 16 * ot);
            if (KEEP) acc[ot] *= rs;                       // normalised value stays in acc
            y[h] = KEEP ? acc[ot] * gm + bt : acc[ot] * (gm * rs) + bt;
        }
        out[t] = relu_packed(pack8(y[0], y[1]));
    }
}
}  // namespace

namespace {
// The same on a row tile in the NATURAL fragment layout (what a 16-byte load of x gives: lane (row c, g) holds features
// 32t + 8g + j of the row in x[t][j]).  Used when the first Linear has been folded into the producer of x
// (first_linear = 0): x is the pre-LayerNorm activation itself.  gam / bet already offset by 8g.
// KEEP: also return the normalised values (bf16) and form the output from those rounded values (backward pass).
// QUAD: the four lanes that share a row are a quad (lane = 4 row + chunk) instead of the same column of the four 16-lane rows
// (lane = row + 16 chunk): the row statistics are then two DPP quad permutes instead of two ds_bpermute round trips.
// GIVEN_RS (CENTERED only): `rstd_out` holds the row's 1/std on entry (saved by the forward kernel, ver_occ_mlp_forward_stats):
// no sum of squares, no cross-lane reduction -- the step is elementwise.
template <bool KEEP, bool QUAD = false, bool CENTERED = false, bool GIVEN_RS = false>
__device__ __forceinline__ void ln_relu_nat(bf16x8 (&x)[4], const float* gam, const float* bet, float eps,
                                            bf16x8 (&xh)[4], float& rstd_out) {
    static_assert(!GIVEN_RS || CENTERED, "a saved 1/std replaces the statistics only when there is no mean pass");
    f32x4 v[4][2];
    float s = 0.0f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const bf16x4 hx = h ? __builtin_shufflevector(x[t], x[t], 4, 5, 6, 7) : __builtin_shufflevector(x[t], x[t], 0, 1, 2, 3);
            v[t][h] = __builtin_convertvector(hx, f32x4);
            if constexpr (!CENTERED) s += (v[t][h].x + v[t][h].y) + (v[t][h].z + v[t][h].w);
        }
    if constexpr (!CENTERED) {
        if constexpr (QUAD) {
            s = group_sum<4>(s);
        } else {
            s += xor16(s);
            s += xor32(s);
        }
    }
    const float mu = s * (1.0f / kW);
    float q = 0.0f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if constexpr (!CENTERED) v[t][h] -= mu;
            if constexpr (!GIVEN_RS)
                q += (v[t][h].x * v[t][h].x + v[t][h].y * v[t][h].y) + (v[t][h].z * v[t][h].z + v[t][h].w * v[t][h].w);
        }
    float rs;
    if constexpr (GIVEN_RS) {
        rs = rstd_out;
    } else {
        if constexpr (QUAD) {
            q = group_sum<4>(q);
        } else {
            q += xor16(q);
            q += xor32(q);
        }
        rs = rsqrtf(q * (1.0f / kW) + eps);
        rstd_out = rs;
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        if (KEEP) {
            xh[t] = pack8(v[t][0] * rs, v[t][1] * rs);
            // (opaque: seeing through pack + unpack the compiler rounds every element on its own -- one single-value
            //  v_cvt_pk_bf16_f32 and one shift each -- and packs the pairs AGAIN for xh: 2.5 instructions per element
            //  instead of 1.5)
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            u32x4 raw = __builtin_bit_cast(u32x4, xh[t]);
            asm("" : "+v"(raw));
            xh[t] = __builtin_bit_cast(bf16x8, raw);
        }
        f32x4 y[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const f32x4 gm = *reinterpret_cast<const f32x4*>(gam + 32 * t + 4 * h);
            const f32x4 bt = *reinterpret_cast<const f32x4*>(bet + 32 * t + 4 * h);
            if (KEEP) {
                const bf16x4 hn = h ? __builtin_shufflevector(xh[t], xh[t], 4, 5, 6, 7) : __builtin_shufflevector(xh[t], xh[t], 0, 1, 2, 3);
                y[h] = __builtin_convertvector(hn, f32x4) * gm + bt;
            } else {
                y[h] = v[t][h] * (gm * rs) + bt;
            }
        }
        x[t] = relu_packed(pack8(y[0], y[1]));
    }
}
}  // namespace

// L1 = false (first_linear = 0, ver_ops.h): x is the output of the first Linear already.  The chain starts with the first
// LayerNorm in the natural fragment layout; the image was packed with W2 in W1's place, so section kF1 holds W2 with the
// natural k order that layout needs (and kB1, in the backward kernel, W2's dgrad with natural-order output rows).
// CENTERED (flags bit 1): every LayerNorm input has zero row mean by construction (see ln_relu_tile).
// STATS: also write 1/std of both LayerNorms per row (rstd f32 [N, 2]; 32-bit buffer addressing: N < 2^28 rows).
template <int RT, bool L1, bool CENTERED = false, bool STATS = false>
__global__ __launch_bounds__(256, 2) void k_occ_mlp_fwd(const __bf16* __restrict__ x, const __bf16* __restrict__ img,
                                                        const float* __restrict__ vec, __bf16* __restrict__ logits,
                                                        long N, float eps, float* __restrict__ rstd) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16x8* frag = reinterpret_cast<bf16x8*>(smem);                       // [kFwdFrags][64]
    float* sv = reinterpret_cast<float*>(smem + kFwdFrags * 1024);        // vectors
    for (int i = threadIdx.x; i < kFwdFrags * 64; i += 256) frag[i] = reinterpret_cast<const bf16x8*>(img)[i];
    for (int i = threadIdx.x; i < kVecFloats; i += 256) sv[i] = vec[i];
    __syncthreads();
    // (wave index as a SCALAR: the per-block buffer resource below is built from it; as a vector value every load through the
    //  resource becomes a readfirstlane loop)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 15, g = lane >> 4;
    const long nblk = (N + 16 * RT - 1) / (16 * RT);
    // (stores beyond row N - 1, and those of the lanes g != 0 -- sent there on purpose -- fall outside the range and are dropped)
    const __amdgpu_buffer_rsrc_t rs_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)rstd, 0, STATS ? (int)(N * 8) : 0, 0x00020000);
    for (long blk = (long)blockIdx.x * 4 + wave; blk < nblk; blk += (long)gridDim.x * 4) {
        const long r0 = blk * (16 * RT);
        // the weight fragments are loop invariant: hide that from LICM, which would otherwise hoist
        // all 68 of them (272 VGPRs) out of the row loop
        int lane_off = lane;
        asm volatile("" : "+v"(lane_off));
        const bf16x8* fr = frag + lane_off;
        const float* sv_g = sv + 4 * (lane_off >> 4);
        bf16x8 bf[RT][4];
        {
            // the block's rows through ONE buffer resource (base = its first row, size = its rows below N): sixteen
            // unconditional loads issued back to back, rows past N read as zeros.  (As `r < N ? load : 0` every load sat in its
            // own basic block behind a branch.)
            const long left = N - r0;
            const int rows = left < 16 * RT ? (int)left : 16 * RT;
            const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(x + r0 * kW), 0, rows * kW * 2, 0x00020000);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
                    bf[rt][kt] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rx, ((rt * 16 + c) * kW + 32 * kt + 8 * g) * 2, 0, 0));
        }
        if constexpr (!L1) {
            const float* sv_n = sv + 8 * (lane_off >> 4);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                float rs;
                bf16x8 unused[4];
                ln_relu_nat<false, false, CENTERED>(bf[rt], sv_n + kW, sv_n + 2 * kW, eps, unused, rs);
                // (1/std of both LayerNorms per row for the backward kernel, which then recomputes them elementwise)
                if constexpr (STATS)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, rs), rs_rsrc,
                                                          g == 0 ? (int)(r0 + rt * 16 + c) * 8 : -8, 0, 0);
            }
        }
#pragma unroll
        for (int layer = L1 ? 0 : 1; layer < 2; ++layer) {
            const float* bias = sv_g + layer * 3 * kW;
            f32x4 acc[RT][8];
#pragma unroll
            for (int ot = 0; ot < 8; ++ot) {
                const f32x4 b = *reinterpret_cast<const f32x4*>(bias + 16 * ot);
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) acc[rt][ot] = b;
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) {
                    const bf16x8 a = fr[((layer && L1 ? kF2 : kF1) + ot * 4 + kt) * 64];
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) acc[rt][ot] = mfma(a, bf[rt][kt], acc[rt][ot]);
                }
            }
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                float rs;
                ln_relu_tile<false, CENTERED>(acc[rt], bias + kW, bias + 2 * kW, eps, bf[rt], rs);
                if constexpr (STATS)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, rs), rs_rsrc,
                                                          g == 0 ? (int)(r0 + rt * 16 + c) * 8 + 4 * layer : -8, 0, 0);
            }
        }
        // layer 3: classes 4g..4g+3 of row c
        const f32x4 b3 = *reinterpret_cast<const f32x4*>(sv_g + 6 * kW);
        f32x4 lo[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) lo[rt] = b3;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            const bf16x8 a = fr[(kF3 + kt) * 64];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) lo[rt] = mfma(a, bf[rt][kt], lo[rt]);
        }
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const long r = r0 + rt * 16 + c;
            if (r < N) *reinterpret_cast<bf16x4*>(logits + r * kC + 4 * g) = pack4(lo[rt]);
        }
    }
}

// ---------------------------------------------------------------------------------------- backward
namespace {
// acc[mt] (+)= sum_ks A[base + 4 mt + ks] * b[ks] for 8 output tiles, A fragments double-buffered
// from LDS one tile ahead.  The scheduling barriers keep the compiler from hoisting all 32 fragment
// reads (128 VGPRs) to the top of the phase.
template <bool BIAS>
__device__ __forceinline__ void gemm8(f32x4 (&acc)[8], const bf16x8* fr, int base, const bf16x8 (&b)[4],
                                      const float* bias) {
    bf16x8 a[2][4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) a[0][ks] = fr[(base + ks) * 64];
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
        if (mt + 1 < 8) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) a[(mt + 1) & 1][ks] = fr[(base + (mt + 1) * 4 + ks) * 64];
        }
        acc[mt] = BIAS ? *reinterpret_cast<const f32x4*>(bias + 16 * mt) : (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) acc[mt] = mfma(a[mt & 1][ks], b[ks], acc[mt]);
        __builtin_amdgcn_sched_barrier(0);
    }
}

__device__ __forceinline__ f32x4 unpack_half(bf16x8 p, int hi) {
    const bf16x4 h = hi ? __builtin_shufflevector(p, p, 4, 5, 6, 7) : __builtin_shufflevector(p, p, 0, 1, 2, 3);
    return __builtin_convertvector(h, f32x4);
}

// Forward LayerNorm+ReLU that also keeps what the backward needs: the normalised values (bf16, same
// pairing as the fragments) and 1/sigma.
__device__ __forceinline__ void ln_relu_keep(f32x4 (&acc)[8], const float* gam, const float* bet, float eps,
                                             bf16x8 (&out)[4], bf16x8 (&xh)[4], float& rstd_out) {
    float s = 0.0f;
#pragma unroll
    for (int ot = 0; ot < 8; ++ot) s += (acc[ot].x + acc[ot].y) + (acc[ot].z + acc[ot].w);
    s += xor16(s);
    s += xor32(s);
    const float mu = s * (1.0f / kW);
    float q = 0.0f;
#pragma unroll
    for (int ot = 0; ot < 8; ++ot) {
        acc[ot] -= mu;
        q += (acc[ot].x * acc[ot].x + acc[ot].y * acc[ot].y) + (acc[ot].z * acc[ot].z + acc[ot].w * acc[ot].w);
    }
    q += xor16(q);
    q += xor32(q);
    const float rs = rsqrtf(q * (1.0f / kW) + eps);
    rstd_out = rs;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        xh[t] = pack8(acc[2 * t] * rs, acc[2 * t + 1] * rs);
        f32x4 y[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int ot = 2 * t + h;
            const f32x4 gm = *reinterpret_cast<const f32x4*>(gam + 16 * ot);
            const f32x4 bt = *reinterpret_cast<const f32x4*>(bet + 16 * ot);
            // from the ROUNDED normalised value: the backward pass re-derives exactly this ReLU gate
            y[h] = __builtin_elementwise_max(unpack_half(xh[t], h) * gm + bt, (f32x4){0.0f, 0.0f, 0.0f, 0.0f});
        }
        out[t] = pack8(y[0], y[1]);
    }
}

// d(pre-LayerNorm) from d(post-ReLU) for one row tile (transposed layout); accumulates d(gamma), d(beta).
// The parameter gradients are sums over ROWS, which sit on the lane index here.  Rather than 128
// per-lane accumulators, dz and dz*xhat (bf16) are fed back as A operands against a constant 0/1
// selector B: D[r][n] = dz[r][feature 16ot+n], i.e. the tile comes back transposed (feature on the
// lane, 4 rows in the registers) and 4 adds fold it into ONE accumulator per (tile, quantity).
// NAT: the tile is in the natural fragment layout (d[2t+h][i], xh[t][4h+i] = feature 32t + 8g + 4h + i; gam / bet offset
// by 8g; sel = the natural-layout selectors) and the Linear in front has been folded away: no d(bias).
template <bool NAT = false>
__device__ __forceinline__ void ln_relu_bwd(f32x4 (&d)[8], const bf16x8 (&xh)[4], float rs, const float* gam,
                                            const float* bet, const bf16x8 (&sel)[2], float (&dgam)[8],
                                            float (&dbet)[8], float (&dbias)[8], bf16x8 (&out)[4]) {
    float s1 = 0.0f, s2 = 0.0f;
    const f32x4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        f32x4 dz[2], dzn[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int ot = 2 * t + h;
            const f32x4 n = unpack_half(xh[t], h);
            const f32x4 gm = *reinterpret_cast<const f32x4*>(gam + (NAT ? 32 * t + 4 * h : 16 * ot));
            const f32x4 bt = *reinterpret_cast<const f32x4*>(bet + (NAT ? 32 * t + 4 * h : 16 * ot));
            const f32x4 y = n * gm + bt;
            dz[h].x = y.x > 0.0f ? d[ot].x : 0.0f;
            dz[h].y = y.y > 0.0f ? d[ot].y : 0.0f;
            dz[h].z = y.z > 0.0f ? d[ot].z : 0.0f;
            dz[h].w = y.w > 0.0f ? d[ot].w : 0.0f;
            dzn[h] = dz[h] * n;
            const f32x4 dg = dz[h] * gm;
            d[ot] = dg;
            s1 += (dg.x + dg.y) + (dg.z + dg.w);
            const f32x4 dn = dg * n;
            s2 += (dn.x + dn.y) + (dn.z + dn.w);
        }
        const bf16x8 pz = pack8(dz[0], dz[1]), pn = pack8(dzn[0], dzn[1]);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const f32x4 tb = mfma(pz, sel[h], zero4);
            const f32x4 tg = mfma(pn, sel[h], zero4);
            dbet[2 * t + h] += (tb.x + tb.y) + (tb.z + tb.w);
            dgam[2 * t + h] += (tg.x + tg.y) + (tg.z + tg.w);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    s1 += xor16(s1);
    s1 += xor32(s1);
    s2 += xor16(s2);
    s2 += xor32(s2);
    const float m1 = s1 * (1.0f / kW), m2 = s2 * (1.0f / kW);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const f32x4 a = (d[2 * t] - m1 - unpack_half(xh[t], 0) * m2) * rs;
        const f32x4 b = (d[2 * t + 1] - m1 - unpack_half(xh[t], 1) * m2) * rs;
        out[t] = pack8(a, b);
        // d(bias of the Linear in front) = column sums of d(a): same selector trick
        if (!NAT) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f32x4 ta = mfma(out[t], sel[h], zero4);
                dbias[2 * t + h] += (ta.x + ta.y) + (ta.z + ta.w);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}
}  // namespace

// One wave = 16 rows per iteration; one wave per SIMD (the chain needs ~380 registers incl. AGPRs).
// Outputs h1, h2, da1, da2 are stored in FRAGMENT feature order: position 32t + 8g + j of a row
// holds feature kperm(t, g, j) (16-byte stores); the caller un-permutes the small weight gradients.
// L1 = false: see k_occ_mlp_fwd.  h1 is then stored in NATURAL feature order and grad_x is the gradient w.r.t. the
// kernel's input = the folded Linear's output; grad_a1 is not written.
#ifndef VER_MLP_BWD_WAVES
#define VER_MLP_BWD_WAVES 4
#endif
constexpr int kBwdWaves = VER_MLP_BWD_WAVES;       // waves per workgroup (one workgroup per CU: the image fills the LDS)
template <bool L1>
__global__ __launch_bounds__(kBwdWaves * 64) void k_occ_mlp_bwd(const __bf16* __restrict__ x, const __bf16* __restrict__ dlog,
                                                        const __bf16* __restrict__ img, const float* __restrict__ vec,
                                                        __bf16* __restrict__ dx, __bf16* __restrict__ da1,
                                                        __bf16* __restrict__ da2, __bf16* __restrict__ h1,
                                                        float* __restrict__ pgrad, long N, float eps) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16x8* frag = reinterpret_cast<bf16x8*>(smem);                       // [kAllFrags][64]
    float* sv = reinterpret_cast<float*>(smem + kAllFrags * 1024);
    for (int i = threadIdx.x; i < kAllFrags * 64; i += kBwdWaves * 64) frag[i] = reinterpret_cast<const bf16x8*>(img)[i];
    for (int i = threadIdx.x; i < kVecFloats; i += kBwdWaves * 64) sv[i] = vec[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    const f32x4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
    float dgam[2][8], dbet[2][8], dbias[2][8];   // feature 16ot + c, partial over the rows 4g..4g+3 of every block
#pragma unroll
    for (int l = 0; l < 2; ++l)
#pragma unroll
        for (int ot = 0; ot < 8; ++ot) dgam[l][ot] = dbet[l][ot] = dbias[l][ot] = 0.0f;
    // selector B fragments: k = 8g+j of a packed pair holds feature 4g+j of tile 2t (j<4) or tile 2t+1
    bf16x8 sel[2];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        sel[0][j] = (__bf16)((j < 4 && 4 * g + j == c) ? 1.0f : 0.0f);
        sel[1][j] = (__bf16)((j >= 4 && 4 * g + j - 4 == c) ? 1.0f : 0.0f);
    }
    // d(W3)[class][feature] = sum_r d(logits)[r][class] * h2[r][feature] is accumulated in-kernel (h2 never
    // goes to HBM): both tiles are turned feature/class-on-the-lane by selector MFMAs, then multiplied
    // with the rows as the (half-filled) K dimension.  dw3[ot][i] = class 4g+i, feature 16ot + c.
    f32x4 dw3[8];
#pragma unroll
    for (int ot = 0; ot < 8; ++ot) dw3[ot] = zero4;
    float db3 = 0.0f;                                // d(b3)[class c] = column sums of d(logits), rows 4g..4g+3 of every block
    bf16x8 sel_nat[2];                               // natural layout: k-slot 8g+j holds feature 8g+j of the 32-block
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        sel_nat[0][j] = (__bf16)((8 * g + j == c) ? 1.0f : 0.0f);
        sel_nat[1][j] = (__bf16)((8 * g + j == c + 16) ? 1.0f : 0.0f);
    }
    bf16x8 sel_cls;                                  // k-slot 8g'+j of the d(logits) fragment holds class 8g'+j (g' < 2)
#pragma unroll
    for (int j = 0; j < 8; ++j) sel_cls[j] = (__bf16)((g < 2 && 8 * g + j == c) ? 1.0f : 0.0f);
    const long nblk = (N + 15) / 16;
    for (long blk = (long)blockIdx.x * kBwdWaves + wave; blk < nblk; blk += (long)gridDim.x * kBwdWaves) {
        int lane_off = lane;
        asm volatile("" : "+v"(lane_off));
        const bf16x8* fr = frag + lane_off;
        const float* sv_g = sv + 4 * (lane_off >> 4);
        const long r = blk * 16 + c;
        const bool ok = r < N;
        const bf16x8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};
        bf16x8 bf[4], xh1[4], xh2[4];
        float rs1, rs2;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) bf[kt] = ok ? *reinterpret_cast<const bf16x8*>(x + r * kW + 32 * kt + 8 * g) : z8;
        f32x4 acc[8];
        const float* sv_n = sv + 8 * (lane_off >> 4);
        // ---- forward, layer 1 (folded away: x is its output, LayerNorm in the natural layout)
        if constexpr (L1) {
            gemm8<true>(acc, fr, kF1, bf, sv_g);
            __builtin_amdgcn_sched_barrier(0);
            ln_relu_keep(acc, sv_g + kW, sv_g + 2 * kW, eps, bf, xh1, rs1);
        } else {
            ln_relu_nat<true>(bf, sv_n + kW, sv_n + 2 * kW, eps, xh1, rs1);
        }
        if (ok) {
#pragma unroll
            for (int t = 0; t < 4; ++t) *reinterpret_cast<bf16x8*>(h1 + r * kW + 32 * t + 8 * g) = bf[t];
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- forward, layer 2
        gemm8<true>(acc, fr, L1 ? kF2 : kF1, bf, sv_g + 3 * kW);
        __builtin_amdgcn_sched_barrier(0);
        ln_relu_keep(acc, sv_g + 4 * kW, sv_g + 5 * kW, eps, bf, xh2, rs2);
        __builtin_amdgcn_sched_barrier(0);
        // ---- d(h2)^T = W3^T d(logits)^T   (K = 16 classes in the lower half of the k-step)
        const bf16x8 dl = (ok && g < 2) ? *reinterpret_cast<const bf16x8*>(dlog + r * kC + 8 * g) : z8;
        {   // d(W3) += d(logits)^T h2 for this row tile
            const f32x4 tdl = mfma(dl, sel_cls, zero4);                       // [row 4g+i][class c]
            db3 += (tdl.x + tdl.y) + (tdl.z + tdl.w);
            const bf16x8 a3 = pack8(tdl, zero4);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const f32x4 th = mfma(bf[t], sel[h], zero4);               // [row 4g+i][feature 16(2t+h)+c]
                    dw3[2 * t + h] = mfma(a3, pack8(th, zero4), dw3[2 * t + h]);
                }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) acc[mt] = mfma(fr[(kB3 + mt) * 64], dl, zero4);
        __builtin_amdgcn_sched_barrier(0);
        ln_relu_bwd(acc, xh2, rs2, sv_g + 4 * kW, sv_g + 5 * kW, sel, dgam[1], dbet[1], dbias[1], bf);
        if (ok) {
#pragma unroll
            for (int t = 0; t < 4; ++t) *reinterpret_cast<bf16x8*>(da2 + r * kW + 32 * t + 8 * g) = bf[t];
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!L1) {
            // ---- d(h1)^T = W2^T d(a2)^T through section kB1 (natural-order output rows), LayerNorm 1 backward in the
            // natural layout, and that is d(x)
#pragma unroll
            for (int mp = 0; mp < 4; ++mp) {
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    f32x4 o = zero4;
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) o = mfma(fr[(kB1 + (mp * 2 + t) * 4 + ks) * 64], bf[ks], o);
                    acc[2 * mp + t] = o;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            ln_relu_bwd<true>(acc, xh1, rs1, sv_n + kW, sv_n + 2 * kW, sel_nat, dgam[0], dbet[0], dbias[0], bf);
            if (ok) {
#pragma unroll
                for (int t = 0; t < 4; ++t) *reinterpret_cast<bf16x8*>(dx + r * kW + 32 * t + 8 * g) = bf[t];
            }
            continue;
        }
        // ---- d(h1)^T = W2^T d(a2)^T
        gemm8<false>(acc, fr, kB2, bf, nullptr);
        __builtin_amdgcn_sched_barrier(0);
        ln_relu_bwd(acc, xh1, rs1, sv_g + kW, sv_g + 2 * kW, sel, dgam[0], dbet[0], dbias[0], bf);
        if (ok) {
#pragma unroll
            for (int t = 0; t < 4; ++t) *reinterpret_cast<bf16x8*>(da1 + r * kW + 32 * t + 8 * g) = bf[t];
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- d(x)^T = W1^T d(a1)^T, tile rows ordered so that a lane ends up with 8 consecutive features
#pragma unroll
        for (int mp = 0; mp < 4; ++mp) {
            f32x4 o[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                o[t] = zero4;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) o[t] = mfma(fr[(kB1 + (mp * 2 + t) * 4 + ks) * 64], bf[ks], o[t]);
            }
            if (ok) *reinterpret_cast<bf16x8*>(dx + r * kW + 32 * mp + 8 * g) = pack8(o[0], o[1]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // ---- d(gamma), d(beta): add the four row groups, then one atomic per feature and wave
#pragma unroll
    for (int l = 0; l < 2; ++l)
#pragma unroll
        for (int ot = 0; ot < 8; ++ot) {
            float a = dgam[l][ot], b = dbet[l][ot], d = dbias[l][ot];
            a += xor16(a);
            a += xor32(a);
            b += xor16(b);
            b += xor32(b);
            d += xor16(d);
            d += xor32(d);
            if (g == 0) {
                atomicAdd(pgrad + (3 * l) * kW + 16 * ot + c, a);
                atomicAdd(pgrad + (3 * l + 1) * kW + 16 * ot + c, b);
                atomicAdd(pgrad + (3 * l + 2) * kW + 16 * ot + c, d);
            }
        }
    // ---- d(W3) [16][128] behind the six vectors, then d(b3) [16]
#pragma unroll
    for (int ot = 0; ot < 8; ++ot)
#pragma unroll
        for (int i = 0; i < 4; ++i) atomicAdd(pgrad + 6 * kW + (4 * g + i) * kW + 16 * ot + c, dw3[ot][i]);
    db3 += xor16(db3);
    db3 += xor32(db3);
    if (g == 0) atomicAdd(pgrad + 6 * kW + kC * kW + c, db3);            // (c < 16 = classes)
}

// ------------------------------------------------------------------------------------------ host
namespace {
int check_common(const char* who, const void* x, const void* img, const float* vec, long N, int width, int classes) {
    VER_REQUIRE(N >= 0, VER_EINVAL, "%s: negative row count", who);
    VER_REQUIRE(width == kW && classes == kC, VER_EUNSUPPORTED, "%s: built for width %d / %d classes (got %d / %d)", who,
                kW, kC, width, classes);
    VER_REQUIRE(img && vec, VER_EINVAL, "%s: null weight image / vector pointer", who);
    if (N == 0) return VER_OK;
    VER_REQUIRE(x, VER_EINVAL, "%s: null pointer argument", who);
    VER_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)img & 15) == 0, VER_EINVAL, "%s: buffers must be 16-byte aligned",
                who);
    return VER_OK;
}
}  // namespace

extern "C" long ver_occ_mlp_image_bytes(void) { return (long)kAllFrags * kFrag * 2; }
extern "C" int ver_occ_mlp_vector_floats(void) { return kVecFloats; }

extern "C" int ver_occ_mlp_pack(const float* W1, const float* W2, const float* W3, void* image, void* stream) {
    VER_REQUIRE(W1 && W2 && W3 && image, VER_EINVAL, "ver_occ_mlp_pack: null pointer argument");
    const int n = kAllFrags * kFrag;
    hipLaunchKernelGGL(k_occ_mlp_pack, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, W1, W2, W3,
                       (__bf16*)image);
    return ver_check_launch("ver_occ_mlp_pack");
}

extern "C" int ver_occ_mlp_forward(const void* x, const void* image, const float* vectors, void* logits, long N,
                                   int width, int classes, float eps, int first_linear, void* stream) {
    return ver_occ_mlp_forward_stats(x, image, vectors, logits, nullptr, N, width, classes, eps, first_linear, stream);
}

extern "C" int ver_occ_mlp_forward_stats(const void* x, const void* image, const float* vectors, void* logits, float* rstd,
                                         long N, int width, int classes, float eps, int first_linear, void* stream) {
    int rc = check_common("ver_occ_mlp_forward", x, image, vectors, N, width, classes);
    if (rc) return rc;
    if (N == 0) return VER_OK;
    VER_REQUIRE(logits, VER_EINVAL, "ver_occ_mlp_forward: null logits pointer");
    constexpr int RT = 4;
    const size_t lds = (size_t)kFwdFrags * 1024 + kVecFloats * sizeof(float);
    // (the attribute is per device: set it on every call, as ver_sca does -- it is a host-side table write)
    // first_linear: bit 0 = the chain starts with Linear 1 (0: folded into the producer of x); bit 1 (VER_OCC_MLP_CENTERED)
    // = every LayerNorm input has zero row mean by construction, the mean pass is skipped
    VER_REQUIRE((first_linear & ~3) == 0, VER_EINVAL, "ver_occ_mlp_forward: unknown flags 0x%x", first_linear);
    const bool l1 = first_linear & 1, centered = first_linear & 2;
    VER_REQUIRE(!rstd || (N < (1L << 28) && ((uintptr_t)rstd & 7) == 0), VER_EUNSUPPORTED,
                "ver_occ_mlp_forward_stats: rstd needs N < 2^28 rows and 8-byte alignment");
    auto kern = l1 ? (centered ? (rstd ? k_occ_mlp_fwd<RT, true, true, true> : k_occ_mlp_fwd<RT, true, true, false>)
                               : (rstd ? k_occ_mlp_fwd<RT, true, false, true> : k_occ_mlp_fwd<RT, true, false, false>))
                   : (centered ? (rstd ? k_occ_mlp_fwd<RT, false, true, true> : k_occ_mlp_fwd<RT, false, true, false>)
                               : (rstd ? k_occ_mlp_fwd<RT, false, false, true> : k_occ_mlp_fwd<RT, false, false, false>));
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return ver_fail(VER_ELAUNCH, "ver_occ_mlp_forward: LDS attribute: %s", hipGetErrorString(e));
    const long nblk = (N + 16 * RT - 1) / (16 * RT);
    long grid = (nblk + 3) / 4;
    if (grid > 512) grid = 512;                            // 2 workgroups per CU, persistent
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, (hipStream_t)stream, (const __bf16*)x,
                       (const __bf16*)image, vectors, (__bf16*)logits, N, eps, rstd);
    return ver_check_launch("ver_occ_mlp_forward");
}

extern "C" int ver_occ_mlp_backward(const void* x, const void* grad_logits, const void* image, const float* vectors,
                                    void* grad_x, void* grad_a1, void* grad_a2, void* h1, float* param_grads, long N,
                                    int width, int classes, float eps, int first_linear, void* stream) {
    int rc = check_common("ver_occ_mlp_backward", x, image, vectors, N, width, classes);
    if (rc) return rc;
    VER_REQUIRE(param_grads, VER_EINVAL, "ver_occ_mlp_backward: null parameter-gradient pointer");
    hipStream_t st = (hipStream_t)stream;
    if (int zrc = ver_zero_async(param_grads, (6 * kW + kC * kW + kC) * sizeof(float), st)) return zrc;   // (kernel: ver_zero_async)
    hipError_t e = hipSuccess;
    if (e != hipSuccess) return ver_fail(VER_ELAUNCH, "ver_occ_mlp_backward: memset: %s", hipGetErrorString(e));
    if (N == 0) return VER_OK;
    VER_REQUIRE(grad_logits && grad_x && (grad_a1 || !first_linear) && grad_a2 && h1, VER_EINVAL,
                "ver_occ_mlp_backward: null pointer argument");
    const size_t lds = (size_t)kAllFrags * 1024 + kVecFloats * sizeof(float);
    auto kern = first_linear ? k_occ_mlp_bwd<true> : k_occ_mlp_bwd<false>;
    e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return ver_fail(VER_ELAUNCH, "ver_occ_mlp_backward: LDS attribute: %s", hipGetErrorString(e));
    const long nblk = (N + 15) / 16;
    long grid = (nblk + kBwdWaves - 1) / kBwdWaves;
    if (grid > 256) grid = 256;                            // one workgroup per CU (LDS bound), persistent
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(kBwdWaves * 64), lds, st, (const __bf16*)x,
                       (const __bf16*)grad_logits, (const __bf16*)image, vectors, (__bf16*)grad_x, (__bf16*)grad_a1,
                       (__bf16*)grad_a2, (__bf16*)h1, param_grads, N, eps);
    return ver_check_launch("ver_occ_mlp_backward");
}

// ============================================================================================
// Backward, folded first Linear, N-SPLIT form (ver_occ_mlp_backward_fused): every parameter gradient -- d(W2)
// included -- is accumulated in the kernel; x and d(logits) are read, d(x) is written, nothing else touches HBM.
//
// The row-split kernel above gives a wave 16 rows and ALL 128 features of every layer: d(W2) would be 64 accumulator
// tiles (256 registers) per wave on top of a chain that already fills the register file, so it writes h1 and d(a2)
// for a host GEMM instead.  Here a workgroup of four waves takes a block of 64 rows and alternates between two views
// of it, with LDS tiles (bf16, row-major, 272-byte rows) as the transposer in between:
//   * ROW view (LayerNorm forward / backward, elementwise): wave w owns rows 16w..16w+15 with all 128 features in the
//     natural fragment layout (lane (row c, g) holds features 32t + 8g + j) -- statistics are in-lane sums + two
//     cross-lane adds, exactly the code of the row-split kernel;
//   * FEATURE view (everything that is a matrix product): wave w owns output features 32w..32w+31 of ALL 64 rows.
//     Its slices of W2 (as A operand of the forward and of the dgrad) and of W3 live in registers for the whole
//     kernel: no weight traffic at all.  Activation operands are 16-byte row reads of the LDS tiles; the operands of
//     the weight gradients and of the row sums need rows on the K axis and come out of the same row-major tiles
//     transposed by ds_read_b64_tr_b16.  d(W2)[o][k] = sum_r d(a2)[r][o] h1[r][k] is then 2 x 8 tiles x 2 FULL k-steps
//     per 64 rows (64 accumulator registers per wave), the LayerNorm parameter gradients are the same A fragments
//     against an all-ones B fragment.
// Per 64 rows and wave: 128 MFMAs (516 in the row-split kernel), no register shuffling through AGPRs, 7 barriers.
// Between the views the activations are rounded to bf16 (a2, d(h2), d(h1)) -- what the layer-by-layer bf16 autocast
// path does between its kernels as well.
namespace {
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
constexpr int kNsLd = kW + 8;               // LDS row stride in elements (272 B: 16-byte aligned, off the 256-B bank period)
constexpr int kNsTiles = 4;
constexpr int kNsDlLd = kC;                 // d(logits) tile [rows][16]
constexpr size_t ns_lds_bytes(int nw) {
    return (size_t)kNsTiles * (16 * nw) * kNsLd * 2 + (size_t)(16 * nw) * kNsDlLd * 2 + kVecFloats * sizeof(float);
}

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

// B (or A) fragment with ROWS on the K axis from a row-major LDS tile: rows row0 + 8g .. + 7 of column col0 + c.
__device__ __forceinline__ bf16x8 ns_tr_frag(const __bf16* tile, int ld, int row0, int col0, int c, int g) {
    const __bf16* p = tile + (row0 + 8 * g + (c >> 2)) * ld + col0 + 4 * (c & 3);
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 4 * ld));
    return __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

// LayerNorm + ReLU backward for one row tile in the natural layout (d[2t+h][i], xh[t][4h+i] = feature 32t + 8g + 4h + i;
// gam / bet offset by 8g).  Writes, for this lane's row, d(pre-LayerNorm) to `out` and the two tiles whose row sums are
// d(beta) / d(gamma) to `pz` / `pn` (pointers to the lane's 8-feature chunk of k-step 0; + 32 elements per k-step) --
// each as soon as it exists, so that no more than one k-step of them is ever live in registers.
// XH: `pn` receives the normalised values n themselves instead of d(z) n (the consumer forms sum_r d(z) n as a matrix product).
template <bool QUAD = false, bool XH = false>
__device__ __forceinline__ void ns_ln_relu_bwd(f32x4 (&d)[8], const bf16x8 (&xh)[4], float rs, const float* gam,
                                               const float* bet, __bf16* out, __bf16* pz, __bf16* pn, bool out_ok) {
    float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        f32x4 dz[2], dzn[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int ot = 2 * t + h;
            const f32x4 n = unpack_half(xh[t], h);
            const f32x4 gm = *reinterpret_cast<const f32x4*>(gam + 32 * t + 4 * h);
            const f32x4 bt = *reinterpret_cast<const f32x4*>(bet + 32 * t + 4 * h);
            const f32x4 y = n * gm + bt;
            dz[h].x = y.x > 0.0f ? d[ot].x : 0.0f;
            dz[h].y = y.y > 0.0f ? d[ot].y : 0.0f;
            dz[h].z = y.z > 0.0f ? d[ot].z : 0.0f;
            dz[h].w = y.w > 0.0f ? d[ot].w : 0.0f;
            if constexpr (!XH) dzn[h] = dz[h] * n;
            const f32x4 dg = dz[h] * gm;
            d[ot] = dg;
            s1 += (dg.x + dg.y) + (dg.z + dg.w);
            const f32x4 dn = dg * n;
            s2 += (dn.x + dn.y) + (dn.z + dn.w);
        }
        *reinterpret_cast<bf16x8*>(pz + 32 * t) = pack8(dz[0], dz[1]);
        *reinterpret_cast<bf16x8*>(pn + 32 * t) = XH ? xh[t] : pack8(dzn[0], dzn[1]);
    }
    if constexpr (QUAD) {
        s1 = group_sum<4>(s1);
        s2 = group_sum<4>(s2);
    } else {
        s1 += xor16(s1);
        s1 += xor32(s1);
        s2 += xor16(s2);
        s2 += xor32(s2);
    }
    const float m1 = s1 * (1.0f / kW), m2 = s2 * (1.0f / kW);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const f32x4 a = (d[2 * t] - m1 - unpack_half(xh[t], 0) * m2) * rs;
        const f32x4 b = (d[2 * t + 1] - m1 - unpack_half(xh[t], 1) * m2) * rs;
        if (out_ok) *reinterpret_cast<bf16x8*>(out + 32 * t) = pack8(a, b);
    }
}
}  // namespace

// NW waves per workgroup: a block is 16 NW rows, a wave owns rows 16w.. (row view) and the 128 / NW output features
// from 128 w / NW (feature view).  NW = 8: two waves per SIMD, 128-row blocks, one 16-feature tile per wave.
// LDS tiles: T0 = h1;  T1 = a2 -> d(h2) -> d(z2) n2 -> d(h1);  T2 = h2 -> d(z2) -> d(z1);  T3 = d(a2) -> d(z1) n1.
template <int NW>
__global__ __launch_bounds__(NW * 64) void k_occ_mlp_bwd_ns(const __bf16* __restrict__ x, const __bf16* __restrict__ dlog,
                                                            const float* __restrict__ W2, const float* __restrict__ W3,
                                                            const float* __restrict__ vec, __bf16* __restrict__ dx,
                                                            float* __restrict__ pgrad, long N, float eps) {
    constexpr int RB = 16 * NW;             // rows per block
    constexpr int OT = kW / NW / 16;        // 16-feature output tiles per wave
    constexpr int RT = NW;                  // 16-row tiles per block
    constexpr int KS = RB / 32;             // k-steps when ROWS are the K axis
    constexpr int TILE = RB * kNsLd;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __bf16* T0 = reinterpret_cast<__bf16*>(smem);
    __bf16* T1 = T0 + TILE;
    __bf16* T2 = T1 + TILE;
    __bf16* T3 = T2 + TILE;
    __bf16* DL = T3 + TILE;                            // d(logits) [RB][16]
    float* sv = reinterpret_cast<float*>(DL + RB * kNsDlLd);
    for (int i = threadIdx.x; i < kVecFloats; i += NW * 64) sv[i] = vec[i];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int f0 = (kW / NW) * w;                      // first output feature of this wave
    const f32x4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
    const bf16x8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};
    // ---- this wave's weight slices, as MFMA A operands, for the whole kernel
    bf16x8 wa2[OT][4];     // forward:  A[m = o][k]      = W2[f0 + 16ot + m][k]
    bf16x8 wb2[OT][4];     // dgrad:    A[m = k_in][k=o] = W2[o][f0 + 16kt + m]
    bf16x8 wa3[OT];        // d(h2):    A[m = f][k = class] = W3[class][f0 + 16ot + m]   (classes >= 16: zero)
#pragma unroll
    for (int ot = 0; ot < OT; ++ot) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const float* p = W2 + (size_t)(f0 + 16 * ot + c) * kW + 32 * ks + 8 * g;
            wa2[ot][ks] = pack8(*reinterpret_cast<const f32x4*>(p), *reinterpret_cast<const f32x4*>(p + 4));
            f32x4 lo, hi;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                lo[j] = W2[(size_t)(32 * ks + 8 * g + j) * kW + f0 + 16 * ot + c];
                hi[j] = W2[(size_t)(32 * ks + 8 * g + 4 + j) * kW + f0 + 16 * ot + c];
            }
            wb2[ot][ks] = pack8(lo, hi);
        }
        f32x4 lo = zero4, hi = zero4;
        if (g < 2) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                lo[j] = W3[(size_t)(8 * g + j) * kW + f0 + 16 * ot + c];
                hi[j] = W3[(size_t)(8 * g + 4 + j) * kW + f0 + 16 * ot + c];
            }
        }
        wa3[ot] = pack8(lo, hi);
    }
    bf16x8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (__bf16)1.0f;
    // ---- accumulators (feature view).  D[m][n]: lane (n = c, g) holds m = 4g .. 4g+3.
    f32x4 dw2[OT][8];                    // [o tile][k tile]: d(W2)[f0 + 16ot + 4g + i][16kt + c]
    f32x4 dw3[OT];                       // d(W3)[class 4g + i][f0 + 16ft + c]
    f32x4 sdb3 = zero4;                  // d(b3)[class 4g + i]            (every column n holds the same sum)
    f32x4 sbet1[OT], sgam1[OT], sbet2[OT], sgam2[OT], sb2[OT];      // [o tile]: feature f0 + 16ot + 4g + i
#pragma unroll
    for (int ot = 0; ot < OT; ++ot) {
        dw3[ot] = sbet1[ot] = sgam1[ot] = sbet2[ot] = sgam2[ot] = sb2[ot] = zero4;
#pragma unroll
        for (int kt = 0; kt < 8; ++kt) dw2[ot][kt] = zero4;
    }
    __syncthreads();
    const float* sv_n = sv + 8 * g;      // natural layout: features 32t + 8g + ...
    const int myrow = 16 * w + c;        // this lane's row of the block (row view)
    const long nblk = (N + RB - 1) / RB;
    bool pending = false;                // the previous block's LayerNorm-1 row sums are still to be taken
    auto ln1_sums = [&]() {              // step 7 (features): d(beta1), d(gamma1) += row sums of the d(z1), d(z1) n1 tiles
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int ot = 0; ot < OT; ++ot) {
                sbet1[ot] = mfma(ns_tr_frag(T2, kNsLd, 32 * ks, f0 + 16 * ot, c, g), ones, sbet1[ot]);
                sgam1[ot] = mfma(ns_tr_frag(T3, kNsLd, 32 * ks, f0 + 16 * ot, c, g), ones, sgam1[ot]);
            }
    };
    for (long blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const long r = blk * RB + myrow;
        const bool ok = r < N;
        // ---------------- step 0 (rows): x -> LayerNorm 1 + ReLU -> h1; stage d(logits)
        bf16x8 xr[4], xh1[4], xh2[4];
        float rs1, rs2;
#pragma unroll
        for (int t = 0; t < 4; ++t) xr[t] = ok ? *reinterpret_cast<const bf16x8*>(x + r * kW + 32 * t + 8 * g) : z8;
        if (g < 2) *reinterpret_cast<bf16x8*>(DL + myrow * kNsDlLd + 8 * g) = ok ? *reinterpret_cast<const bf16x8*>(dlog + r * kC + 8 * g) : z8;
        ln_relu_nat<true>(xr, sv_n + kW, sv_n + 2 * kW, eps, xh1, rs1);
#pragma unroll
        for (int t = 0; t < 4; ++t) *reinterpret_cast<bf16x8*>(T0 + myrow * kNsLd + 32 * t + 8 * g) = xr[t];
        __syncthreads();
        if (pending) ln1_sums();
        // ---------------- step 1 (features): a2 = W2 h1 + b2 for this wave's output features, all rows
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            f32x4 acc[OT];
#pragma unroll
            for (int ot = 0; ot < OT; ++ot) acc[ot] = *reinterpret_cast<const f32x4*>(sv + 3 * kW + f0 + 16 * ot + 4 * g);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 b = *reinterpret_cast<const bf16x8*>(T0 + (16 * rt + c) * kNsLd + 32 * ks + 8 * g);
#pragma unroll
                for (int ot = 0; ot < OT; ++ot) acc[ot] = mfma(wa2[ot][ks], b, acc[ot]);
            }
#pragma unroll
            for (int ot = 0; ot < OT; ++ot)
                *reinterpret_cast<bf16x4*>(T1 + (16 * rt + c) * kNsLd + f0 + 16 * ot + 4 * g) = pack4(acc[ot]);
            __builtin_amdgcn_sched_barrier(0);         // (keeps the operand reads of the next row tiles from piling up in registers)
        }
        __syncthreads();
        // ---------------- step 2 (rows): LayerNorm 2 + ReLU -> h2
#pragma unroll
        for (int t = 0; t < 4; ++t) xr[t] = *reinterpret_cast<const bf16x8*>(T1 + myrow * kNsLd + 32 * t + 8 * g);
        ln_relu_nat<true>(xr, sv_n + 4 * kW, sv_n + 5 * kW, eps, xh2, rs2);
#pragma unroll
        for (int t = 0; t < 4; ++t) *reinterpret_cast<bf16x8*>(T2 + myrow * kNsLd + 32 * t + 8 * g) = xr[t];
        __syncthreads();
        // ---------------- step 3 (features): d(h2) = W3^T d(logits); d(W3), d(b3)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const bf16x8 b = g < 2 ? *reinterpret_cast<const bf16x8*>(DL + (16 * rt + c) * kNsDlLd + 8 * g) : z8;
#pragma unroll
            for (int ot = 0; ot < OT; ++ot) {
                const f32x4 d = mfma(wa3[ot], b, zero4);
                *reinterpret_cast<bf16x4*>(T1 + (16 * rt + c) * kNsLd + f0 + 16 * ot + 4 * g) = pack4(d);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const bf16x8 a = ns_tr_frag(DL, kNsDlLd, 32 * ks, 0, c, g);           // A[m = class][k = row]
            if (w == 0) sdb3 = mfma(a, ones, sdb3);
#pragma unroll
            for (int ft = 0; ft < OT; ++ft)
                dw3[ft] = mfma(a, ns_tr_frag(T2, kNsLd, 32 * ks, f0 + 16 * ft, c, g), dw3[ft]);
        }
        __syncthreads();
        // ---------------- step 4 (rows): LayerNorm 2 backward -> d(a2); the tiles of d(beta2), d(gamma2)
        {
            f32x4 d[8];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const bf16x8 v = *reinterpret_cast<const bf16x8*>(T1 + myrow * kNsLd + 32 * t + 8 * g);
                d[2 * t] = unpack_half(v, 0);
                d[2 * t + 1] = unpack_half(v, 1);
            }
            // (T1: d(z2) n2 overwrites this lane's own d(h2) chunks, all read above)
            ns_ln_relu_bwd(d, xh2, rs2, sv_n + 4 * kW, sv_n + 5 * kW, T3 + myrow * kNsLd + 8 * g, T2 + myrow * kNsLd + 8 * g,
                           T1 + myrow * kNsLd + 8 * g, true);
        }
        __syncthreads();
        // ---------------- step 5 (features): row sums, d(W2), d(h1) = W2^T d(a2)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            bf16x8 ada[OT];
#pragma unroll
            for (int ot = 0; ot < OT; ++ot) {
                ada[ot] = ns_tr_frag(T3, kNsLd, 32 * ks, f0 + 16 * ot, c, g);             // A[m = o][k = row] of d(a2)
                sb2[ot] = mfma(ada[ot], ones, sb2[ot]);
                sbet2[ot] = mfma(ns_tr_frag(T2, kNsLd, 32 * ks, f0 + 16 * ot, c, g), ones, sbet2[ot]);
                sgam2[ot] = mfma(ns_tr_frag(T1, kNsLd, 32 * ks, f0 + 16 * ot, c, g), ones, sgam2[ot]);
            }
#pragma unroll
            for (int kt = 0; kt < 8; ++kt) {
                const bf16x8 b = ns_tr_frag(T0, kNsLd, 32 * ks, 16 * kt, c, g);           // B[k = row][n = input feature] of h1
#pragma unroll
                for (int ot = 0; ot < OT; ++ot) dw2[ot][kt] = mfma(ada[ot], b, dw2[ot][kt]);
                if (kt & 1) __builtin_amdgcn_sched_barrier(0);
            }
        }
        bf16x4 dh1[RT][OT];                                   // held over the barrier: T1 is still being read (d(z2) n2)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            f32x4 acc[OT];
#pragma unroll
            for (int kt = 0; kt < OT; ++kt) acc[kt] = zero4;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 b = *reinterpret_cast<const bf16x8*>(T3 + (16 * rt + c) * kNsLd + 32 * ks + 8 * g);
#pragma unroll
                for (int kt = 0; kt < OT; ++kt) acc[kt] = mfma(wb2[kt][ks], b, acc[kt]);
            }
#pragma unroll
            for (int kt = 0; kt < OT; ++kt) dh1[rt][kt] = pack4(acc[kt]);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int kt = 0; kt < OT; ++kt)
                *reinterpret_cast<bf16x4*>(T1 + (16 * rt + c) * kNsLd + f0 + 16 * kt + 4 * g) = dh1[rt][kt];
        __syncthreads();
        // ---------------- step 6 (rows): LayerNorm 1 backward -> d(x); the tiles of d(beta1), d(gamma1)
        {
            f32x4 d[8];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const bf16x8 v = *reinterpret_cast<const bf16x8*>(T1 + myrow * kNsLd + 32 * t + 8 * g);
                d[2 * t] = unpack_half(v, 0);
                d[2 * t + 1] = unpack_half(v, 1);
            }
            ns_ln_relu_bwd(d, xh1, rs1, sv_n + kW, sv_n + 2 * kW, dx + r * kW + 8 * g, T2 + myrow * kNsLd + 8 * g,
                           T3 + myrow * kNsLd + 8 * g, ok);
        }
        pending = true;
        // (step 7 -- the row sums of LayerNorm 1 from T2 / T3 -- runs after the NEXT barrier: the next block's step 0
        //  only writes T0 and DL, both last read before the barrier above step 6; T2 / T3 are not written again before
        //  the next block's second / fourth barrier)
    }
    if (pending) {
        __syncthreads();
        ln1_sums();
    }
    // ---- parameter gradients: one atomic per element and workgroup
#pragma unroll
    for (int ot = 0; ot < OT; ++ot) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int f = f0 + 16 * ot + 4 * g + i;
#pragma unroll
            for (int kt = 0; kt < 8; ++kt) atomicAdd(pgrad + 6 * kW + kC * kW + kC + (size_t)f * kW + 16 * kt + c, dw2[ot][kt][i]);
            atomicAdd(pgrad + 6 * kW + (4 * g + i) * kW + f0 + 16 * ot + c, dw3[ot][i]);
            if (c == 0) {
                atomicAdd(pgrad + 0 * kW + f, sgam1[ot][i]);
                atomicAdd(pgrad + 1 * kW + f, sbet1[ot][i]);
                atomicAdd(pgrad + 3 * kW + f, sgam2[ot][i]);
                atomicAdd(pgrad + 4 * kW + f, sbet2[ot][i]);
                atomicAdd(pgrad + 5 * kW + f, sb2[ot][i]);
            }
        }
    }
    if (w == 0 && c == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) atomicAdd(pgrad + 6 * kW + kC * kW + 4 * g + i, sdb3[i]);
    }
}

// --------------------------------------------------------------------------------------------
// The same arithmetic, WAVE SPECIALISED (k_occ_mlp_bwd_ws): in the kernel above all eight waves are in the row view or all in
// the feature view, so a SIMD's two waves always want the same unit -- VALU in the LayerNorm steps, LDS + matrix cores in the
// product steps -- and the two kinds of work add up (measured: 8.7 ms + 9.5 ms of a 17.5-ms launch over 32 M rows).  Here
// waves 0-3 are the ROW team (LayerNorm forward / backward of 16 rows each, nothing else: no weights, no accumulators) and
// waves 4-7 the FEATURE team (32 output features each: weights, d(W2) and the row sums in registers, nothing else), one wave
// of each team per SIMD, and TWO 64-row blocks A and B are in flight, three steps apart: in every slot the row team runs a
// row step of one block while the feature team runs a product step of the other.
//     slot     0      1      2      3      4      5      6      7          (one workgroup barrier after each)
//     rows     A.0    B.6    A.2    B.0    A.4    B.2    A.6    B.4
//     features B.5    A.1    B.7    A.3    B.1    A.5    B.3    A.7
// Steps of a block: 0 LN1 fwd | 1 a2 = W2 h1 | 2 LN2 fwd | 3 d(h2), d(W3) | 4 LN2 bwd | 5 LN2 sums, d(h1) | 6 LN1 bwd |
// 7 LN1 sums, d(W2), store d(x)  (d(W2) in step 5 with -DVER_WS_DW2_STEP7=0: the same time on the box that measured both).
// LDS: per block set T0 h1, T1 a2 -> d(h2) -> n2 -> n1, T2 h2 -> d(z2) -> d(z1), T3 d(a2), DL; ONE shared tile T4 for d(h1),
// which step 6 turns into d(x) in place (A writes it in slot 5, rewrites it in slot 6, the feature team stores it in slot 7;
// B in slots 0, 1 and 3): 9 x 17 KB + 4 KB + vectors = 159.6 KB.
// Round 6 (profiles/r06_occ_mlp_bwd_timeline.txt has the slot timeline before and after): the kernel is the VALU work of the
// ROW team -- one wave per SIMD issues an instruction every 6-8 cycles whatever it is -- so that team lost instructions and
// everything that is not arithmetic: the ReLU mask is a 16-bit integer product with the ReLU's own output, the LayerNorm
// backward two fused multiply-adds per value, global operands go through per-block buffer resources (no selects, no 64-bit
// lane arithmetic), no packed fp32 (build.py), d(x) leaves through the shared tile and the feature team, and the two
// s_waitcnt vmcnt(0) of a round are written where they cost nothing.  29.3 -> 24.8-25.7 ms over 96.8 M rows.
namespace {
constexpr int kWsRows = 64;
#ifndef VER_WS_DW2_STEP7
#define VER_WS_DW2_STEP7 1
#endif
#ifndef VER_WS_AHEAD
#define VER_WS_AHEAD 2
#endif
#ifndef VER_WS_BIAS_LATE
#define VER_WS_BIAS_LATE 0
#endif
constexpr bool kWsBiasLate = VER_WS_BIAS_LATE;         // step 1: b2 added to the finished tile (8 more registers) or the accumulators' start value
constexpr int kWsAhead = VER_WS_AHEAD;                 // operand fragments requested ahead in the 64-row products of steps 1 and 5
constexpr bool kWsDw2InStep7 = VER_WS_DW2_STEP7;       // d(W2) in step 7 (beside the row team's heavy LN2 backward) or in step 5
constexpr int kWsTile = kWsRows * kNsLd;
constexpr int kWsVec0 = kW;                                      // b1 is not staged (folded mode)
constexpr size_t kWsLds = (size_t)9 * kWsTile * 2 + (size_t)2 * kWsRows * kNsDlLd * 2 + (kVecFloats - kWsVec0) * sizeof(float);
}  // namespace

#ifdef VER_WS_TIMELINE
// Slot timeline of k_occ_mlp_bwd_ws (scratch/r06/ws_timeline.py; build with -DVER_WS_TIMELINE): s_memtime of every wave of
// kWsTlProbes workgroups at the end of each slot's work (before the barrier) and behind the barrier, for kWsTlRounds rounds
// from round kWsTlFirst on.  g_ws_tl[probe][wave][round][slot][2]; [probe][wave] start / end stamps in g_ws_tl_span.
constexpr int kWsTlProbes = 8, kWsTlRounds = 16, kWsTlFirst = 200;
__device__ long long g_ws_tl[kWsTlProbes * 8 * kWsTlRounds * 8 * 2];
__device__ long long g_ws_tl_span[kWsTlProbes * 8 * 2];
__device__ __forceinline__ int ws_tl_probe() { return (blockIdx.x % 29 == 3 && blockIdx.x / 29 < kWsTlProbes) ? (int)(blockIdx.x / 29) : -1; }
#define WS_TL(k, slot, which)                                                                                       \
    do {                                                                                                            \
        const int pr_ = ws_tl_probe();                                                                              \
        const long kk_ = (k) - kWsTlFirst;                                                                          \
        if (pr_ >= 0 && kk_ >= 0 && kk_ < kWsTlRounds && (threadIdx.x & 63) == 0)                                   \
            g_ws_tl[(((pr_ * 8 + (threadIdx.x >> 6)) * kWsTlRounds + kk_) * 8 + (slot)) * 2 + (which)] =            \
                (long long)__builtin_amdgcn_s_memtime();                                                            \
    } while (0)
#define WS_SPAN(which)                                                                                              \
    do {                                                                                                            \
        const int pr_ = ws_tl_probe();                                                                              \
        if (pr_ >= 0 && (threadIdx.x & 63) == 0)                                                                    \
            g_ws_tl_span[(pr_ * 8 + (threadIdx.x >> 6)) * 2 + (which)] = (long long)__builtin_amdgcn_s_memtime();   \
    } while (0)
// (-DVER_WS_TIMELINE_FINE: four more stamps inside the LayerNorm backward of slots 1 and 4; they cost those steps ~600 cycles)
#ifdef VER_WS_TIMELINE_FINE
__device__ long long g_ws_tl_fine[kWsTlProbes * 8 * kWsTlRounds * 8];
#define WS_TLF(k, idx)                                                                                              \
    do {                                                                                                            \
        const int pr_ = ws_tl_probe();                                                                              \
        const long kk_ = (k) - kWsTlFirst;                                                                          \
        if (pr_ >= 0 && kk_ >= 0 && kk_ < kWsTlRounds && (threadIdx.x & 63) == 0 && (idx) < 8)                      \
            g_ws_tl_fine[((pr_ * 8 + (threadIdx.x >> 6)) * kWsTlRounds + kk_) * 8 + (idx)] =                        \
                (long long)__builtin_amdgcn_s_memtime();                                                            \
    } while (0)
extern "C" int ver_ws_timeline_fine_read(long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ws_tl_fine), sizeof(long long) * kWsTlProbes * 8 * kWsTlRounds * 8);
}
#else
#define WS_TLF(k, idx) do { } while (0)
#endif
extern "C" int ver_ws_timeline_read(long long* slots, long long* span) {
    hipError_t e = hipMemcpyFromSymbol(slots, HIP_SYMBOL(g_ws_tl), sizeof(long long) * kWsTlProbes * 8 * kWsTlRounds * 8 * 2);
    if (e != hipSuccess) return (int)e;
    return (int)hipMemcpyFromSymbol(span, HIP_SYMBOL(g_ws_tl_span), sizeof(long long) * kWsTlProbes * 8 * 2);
}
#else
#define WS_TLF(k, idx) do { } while (0)
#define WS_TL(k, slot, which) do { } while (0)
#define WS_SPAN(which) do { } while (0)
#endif

namespace {
// Workgroup barrier that orders LDS traffic only: this wave's LDS operations are complete, global loads may stay in flight
// across it (__syncthreads() also waits for vmcnt(0), which would serialise the row team's prefetch of the next block's x).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
#define WS_SLOT_END(k, n) do { WS_TL(k, n, 0); lds_barrier(); WS_TL(k, n, 1); } while (0)
}  // namespace

// CENTERED: the forward ran with VER_OCC_MLP_CENTERED (zero-mean LayerNorm inputs by construction): the two recomputed
// LayerNorm-forward steps of the row team skip the mean pass as the forward kernel does.
// RSTD (with CENTERED): the forward kernel saved 1/std of both LayerNorms per row (`rstd` f32 [N, 2]): the two recomputed
// LayerNorm-forward steps are elementwise -- no sum of squares, no cross-lane reduction in the middle of the slot's chain.
// ROWS4 (with RSTD): the row team's other lane mapping -- a lane owns 8 features of FOUR rows (a row = the 16 lanes of a DPP
// row) instead of 32 features of one row (a row = a quad).  The LayerNorm parameters of a lane are then 8 + 8 floats per
// layer and live in registers for the whole kernel (the quad mapping reads 16 x 16 bytes of them from LDS in every step:
// ~20 % of a slot, all four row waves at once behind the barrier), the four rows are four independent dependency chains, the
// row sums of the backward steps are 4-step DPP reductions; with the saved statistics the forward steps have none.
template <bool CENTERED, bool RSTD = false, bool ROWS4 = false>
__global__ __launch_bounds__(512) void k_occ_mlp_bwd_ws(const __bf16* __restrict__ x, const __bf16* __restrict__ dlog,
                                                        const float* __restrict__ W2, const float* __restrict__ W3,
                                                        const float* __restrict__ vec, __bf16* __restrict__ dx,
                                                        float* __restrict__ pgrad, long N, float eps,
                                                        const float* __restrict__ grad_scale,
                                                        const float* __restrict__ rstd) {
    constexpr int OT = 2, RT = 4, KS = 2;
    const float gscale = grad_scale ? grad_scale[0] : 1.0f;        // scalar factor of d(logits) (device-side, may be null)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __bf16* const tiles = reinterpret_cast<__bf16*>(smem);              // [set][4] tiles, then the shared d(h1) tile
    __bf16* const T4 = tiles + 8 * kWsTile;
    __bf16* const DLs = T4 + kWsTile;                                    // [set][64][16]
    float* const sv = reinterpret_cast<float*>(DLs + 2 * kWsRows * kNsDlLd) - kWsVec0;   // sv[i] valid for i >= 128
    for (int i = kWsVec0 + threadIdx.x; i < kVecFloats; i += 512) sv[i] = vec[i];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int team = wv >> 2, q = wv & 3;
    // feature team: MFMA operand coordinates (c = lane & 15, g = lane >> 4); row team: row = lane >> 2, chunk = lane & 3
    // (a row's four 8-feature chunks of a k-step sit on one quad: DPP reductions)
    const int c = team ? (lane & 15) : (lane >> 2), g = team ? (lane >> 4) : (lane & 3);
    const f32x4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
    const bf16x8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};
    // this workgroup's blocks: blockIdx.x, + gridDim.x, ...; round k works on block 2k (A) and 2k + 1 (B)
    const long nblk = (N + kWsRows - 1) / kWsRows;
    const long nmine = blockIdx.x < nblk ? (nblk - 1 - blockIdx.x) / gridDim.x + 1 : 0;
    const long rounds = nmine / 2 + 1;
    __syncthreads();
    WS_SPAN(0);
    if constexpr (ROWS4) {
        static_assert(!ROWS4 || (RSTD && CENTERED), "the 4-row mapping is built for the saved-statistics form");
        if (team == 0) {
            // =============================================================== ROW team, 4 rows x 8 features per lane
            // lane l: row group l >> 4 (rows 16 q + 4 (l >> 4) + i), feature chunk fc = ((l & 15) + 12 (l >> 4)) & 15 -- the
            // rotation makes the 16-byte tile reads of a wave conflict free at the 272-byte row stride
            const int rgp = lane >> 4, fc = ((lane & 15) + 12 * rgp) & 15;
            const int rbase = 16 * q + 4 * rgp;
            f32x4 g1v[2], b1v[2], g2v[2], b2v[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                g1v[h] = *reinterpret_cast<const f32x4*>(vec + kW + 8 * fc + 4 * h);
                b1v[h] = *reinterpret_cast<const f32x4*>(vec + 2 * kW + 8 * fc + 4 * h);
                g2v[h] = *reinterpret_cast<const f32x4*>(vec + 4 * kW + 8 * fc + 4 * h);
                b2v[h] = *reinterpret_cast<const f32x4*>(vec + 5 * kW + 8 * fc + 4 * h);
            }
            struct Row4 {
                bf16x8 xh1[4], xh2[4];
                f32x4 rsv[2];                   // saved 1/std of the block in flight: {LN1, LN2} of rows 2j, 2j + 1
            };
            // ONE operand buffer for both blocks (the register file has no room for two): A's step 0 (slot 0) consumes it and
            // requests B's rows, B's step 0 (slot 3) consumes those and requests the next round's A rows -- three and five
            // slots of flight
            bf16x8 xn[4], dln;
            f32x4 rsn[2];                       // ... and the block's saved 1/std, moved into its state in step 0
            Row4 sa, sb;
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            const int dlrow = lane >> 1, dlh = lane & 1;            // d(logits) staging: lanes 0-31 of a wave, 16 rows x 2 halves
            // Global operands go through ONE BUFFER RESOURCE PER BLOCK (base = the block's first row, size = its rows below N,
            // both wave-uniform): a lane's byte offsets into a block never change, rows past N -- or a whole block this
            // workgroup does not have -- read as zeros and are not written, with no select, no branch and no 64-bit address
            // arithmetic per lane (they were a quarter of the instructions of step 0 and a tenth of step 6).
            const unsigned ones16 = 0x00010001u;
            const int xoff = (rbase * kW + 8 * fc) * 2, dloff = ((16 * q + dlrow) * kC + 8 * dlh) * 2, rsoff = rbase * 8;
            const int nfull = (int)(N / kWsRows), ntail = (int)(N % kWsRows);      // (block ids fit an int: N < 2^37)
            auto block_rsrc = [&](const void* base, long blk, int row_bytes) {
                const int b = blk < nblk ? (int)blk : nfull + 1;                // past the end: no rows
                const int rows = b < nfull ? kWsRows : (b == nfull ? ntail : 0);
                return __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)base + (long)b * (kWsRows * row_bytes)), 0, rows * row_bytes,
                                                         0x00020000);
            };
            // (the loads' results are bit_cast as a WHOLE: this hipcc compiles __builtin_bit_cast(float, v.y) of a vector element
            //  as a read of element 0, and an implicit conversion of the builtin's result to an ext_vector_type as a splat)
            // EVERY global load of the row team is issued here, once per block, UNCONDITIONALLY (a block this workgroup does not
            // have reads as zeros) and behind the step that consumed the buffer: the compiler then counts the loads in flight
            // exactly (s_waitcnt vmcnt(n) at the first use, three slots later) and lands them in the buffer's own registers.
            // Requested inside the conditional steps -- or the 1/std at the end of step 6, as round 5 had it -- every first use
            // became vmcnt(0) behind whatever was requested last (~600 cycles in the longest step of the round).
            // (1/std of rows past N: 0 -- the row contributes nothing)
            auto prefetch = [&](long blk) {
                const __amdgpu_buffer_rsrc_t rx = block_rsrc(x, blk, kW * 2), rd = block_rsrc(dlog, blk, kC * 2);
                const __amdgpu_buffer_rsrc_t rr = block_rsrc(rstd, blk, 8);
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    xn[i] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rx, xoff + i * kW * 2, 0, 0));
                dln = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rd, dloff, 0, 0));
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    rsn[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rr, rsoff + 16 * j, 0, 0));
            };
#define WS_RS1(st) {st.rsv[0].x, st.rsv[0].z, st.rsv[1].x, st.rsv[1].z}
#define WS_RS2(st) {st.rsv[0].y, st.rsv[0].w, st.rsv[1].y, st.rsv[1].w}
            // LayerNorm forward of the lane's 4 x 8 values with the saved 1/std: n = v rs (kept as bf16), y = n gamma + beta
            // from the ROUNDED n, h = relu(y) -> tile
            auto ln_fwd4 = [&](const bf16x8 (&xin)[4], const float (&rs)[4], const f32x4 (&gm)[2], const f32x4 (&bt)[2],
                               bf16x8 (&xh)[4], __bf16* dst) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    xh[i] = pack8(unpack_half(xin[i], 0) * rs[i], unpack_half(xin[i], 1) * rs[i]);
                    u32x4 raw = __builtin_bit_cast(u32x4, xh[i]);
                    asm("" : "+v"(raw));                      // (see ln_relu_nat: keeps the pack + unpack from being seen through)
                    xh[i] = __builtin_bit_cast(bf16x8, raw);
                    const f32x4 y0 = unpack_half(xh[i], 0) * gm[0] + bt[0], y1 = unpack_half(xh[i], 1) * gm[1] + bt[1];
                    *reinterpret_cast<bf16x8*>(dst + (rbase + i) * kNsLd + 8 * fc) = relu_packed(pack8(y0, y1));
                }
            };
            // LayerNorm + ReLU backward: din = gradient w.r.t. the post-ReLU values (4 rows x 8), hin = those post-ReLU values
            // themselves (still in their tile), xh = kept normalised values.  d(z) -> pz tile, n -> pn tile (the feature team
            // forms their row sums), d(pre-LayerNorm) -> put(i, 8 values of row i).
            //   * the ReLU mask comes from the ReLU's OUTPUT, as packed 16-bit integers: h is a bf16 in [+0, inf), so min(h, 1)
            //     as u16 is 0 / 1 and d(z) = d * min(h, 1) as a 16-bit INTEGER product is d or 0 -- already the packed bf16 the
            //     tile wants: one instruction per value where recomputing y = n gamma + beta, comparing, selecting and
            //     re-packing took 3.5;
            //   * a = (dg - m1 - n m2) rs is evaluated as dg rs + (n k2 + k1) with k1 = -m1 rs, k2 = -m2 rs per row: two fused
            //     multiply-adds per value instead of four operations.
            long tl_k = 0;
            int tl_base = 0;
            (void)tl_k; (void)tl_base;
            auto ln_bwd4 = [&](const bf16x8 (&din)[4], const bf16x8 (&hin)[4], const bf16x8 (&xh)[4], const float (&rs)[4],
                               const f32x4 (&gm)[2], __bf16* pz, __bf16* pn, auto&& put) {
                WS_TLF(tl_k, tl_base + 0);
                f32x4 dg[4][2], nn[4][2];
                float s1[4], s2[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const u32x4 hw = __builtin_bit_cast(u32x4, hin[i]), dw = __builtin_bit_cast(u32x4, din[i]);
                    u32x4 zw;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        // (as instructions: written as vector code the compiler turns min(h, 1) into two 16-bit compares,
                        //  two selects and a byte permute per pair)
                        unsigned t;
                        asm("v_pk_min_u16 %0, %1, %2" : "=v"(t) : "v"(hw[j]), "v"(ones16));
                        asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(t) : "v"(dw[j]), "v"(t));
                        zw[j] = t;
                    }
                    const bf16x8 dz = __builtin_bit_cast(bf16x8, zw);
                    *reinterpret_cast<bf16x8*>(pz + (rbase + i) * kNsLd + 8 * fc) = dz;
                    *reinterpret_cast<bf16x8*>(pn + (rbase + i) * kNsLd + 8 * fc) = xh[i];
                    float sa1 = 0.0f, sb1 = 0.0f, sa2 = 0.0f, sb2 = 0.0f;
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        nn[i][h] = unpack_half(xh[i], h);
                        dg[i][h] = unpack_half(dz, h) * gm[h];
                        sa1 += dg[i][h].x + dg[i][h].y;
                        sb1 += dg[i][h].z + dg[i][h].w;
                        sa2 = __builtin_fmaf(dg[i][h].x, nn[i][h].x, sa2);
                        sb2 = __builtin_fmaf(dg[i][h].y, nn[i][h].y, sb2);
                        sa2 = __builtin_fmaf(dg[i][h].z, nn[i][h].z, sa2);
                        sb2 = __builtin_fmaf(dg[i][h].w, nn[i][h].w, sb2);
                    }
                    s1[i] = sa1 + sb1;
                    s2[i] = sa2 + sb2;
                }
                WS_TLF(tl_k, tl_base + 1);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    s1[i] = group_sum<16>(s1[i]);
                    s2[i] = group_sum<16>(s2[i]);
                }
                WS_TLF(tl_k, tl_base + 2);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float k1 = -(s1[i] * (1.0f / kW)) * rs[i], k2 = -(s2[i] * (1.0f / kW)) * rs[i];
                    f32x4 a[2];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        a[h].x = __builtin_fmaf(dg[i][h].x, rs[i], __builtin_fmaf(nn[i][h].x, k2, k1));
                        a[h].y = __builtin_fmaf(dg[i][h].y, rs[i], __builtin_fmaf(nn[i][h].y, k2, k1));
                        a[h].z = __builtin_fmaf(dg[i][h].z, rs[i], __builtin_fmaf(nn[i][h].z, k2, k1));
                        a[h].w = __builtin_fmaf(dg[i][h].w, rs[i], __builtin_fmaf(nn[i][h].w, k2, k1));
                    }
                    put(i, pack8(a[0], a[1]));
                }
                WS_TLF(tl_k, tl_base + 3);
            };
            auto r0 = [&](Row4& st, __bf16* T, __bf16* DL) {
                st.rsv[0] = rsn[0];
                st.rsv[1] = rsn[1];
                if (lane < 32) {
                    bf16x8 dl = dln;
                    if (gscale != 1.0f) {        // (wave-uniform) d(logits) arrives unscaled: ver_focal_loss_forward_grad
                        const f32x4 lo = __builtin_convertvector(__builtin_shufflevector(dl, dl, 0, 1, 2, 3), f32x4) * gscale;
                        const f32x4 hi = __builtin_convertvector(__builtin_shufflevector(dl, dl, 4, 5, 6, 7), f32x4) * gscale;
                        dl = pack8(lo, hi);
                    }
                    *reinterpret_cast<bf16x8*>(DL + (16 * q + dlrow) * kNsDlLd + 8 * dlh) = dl;
                }
                const float rs[4] = WS_RS1(st);
                ln_fwd4(xn, rs, g1v, b1v, st.xh1, T);
            };
            auto r2 = [&](Row4& st, __bf16* T) {                                      // a2 (T1) -> LN2 + ReLU -> h2 (T2)
                bf16x8 xr[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) xr[i] = *reinterpret_cast<const bf16x8*>(T + kWsTile + (rbase + i) * kNsLd + 8 * fc);
                const float rs[4] = WS_RS2(st);
                ln_fwd4(xr, rs, g2v, b2v, st.xh2, T + 2 * kWsTile);
            };
            auto r4 = [&](Row4& st, __bf16* T) {                 // d(h2) (T1), h2 (T2) -> LN2 bwd -> d(a2) T3, d(z2) T2, n2 T1
                bf16x8 d[4], h[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    d[i] = *reinterpret_cast<const bf16x8*>(T + kWsTile + (rbase + i) * kNsLd + 8 * fc);
                    h[i] = *reinterpret_cast<const bf16x8*>(T + 2 * kWsTile + (rbase + i) * kNsLd + 8 * fc);
                }
                __bf16* const out = T + 3 * kWsTile + rbase * kNsLd + 8 * fc;
                const float rs[4] = WS_RS2(st);
                ln_bwd4(d, h, st.xh2, rs, g2v, T + 2 * kWsTile, T + kWsTile,
                        [&](int i, bf16x8 v) { *reinterpret_cast<bf16x8*>(out + i * kNsLd) = v; });
            };
            auto r6 = [&](Row4& st, __bf16* T) {                 // d(h1) (T4), h1 (T0) -> LN1 bwd -> d(x) T4 (in place); d(z1) T2, n1 T1
                bf16x8 d[4], h[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    d[i] = *reinterpret_cast<const bf16x8*>(T4 + (rbase + i) * kNsLd + 8 * fc);
                    h[i] = *reinterpret_cast<const bf16x8*>(T + (rbase + i) * kNsLd + 8 * fc);
                }
                // d(x) replaces d(h1) in the shared tile, lane for lane; the FEATURE team stores it in its next light slot
                // (four 1-KB stores cost this wave ~500 cycles of a slot it is the critical path of)
                __bf16* const out = T4 + rbase * kNsLd + 8 * fc;
                const float rs[4] = WS_RS1(st);
                ln_bwd4(d, h, st.xh1, rs, g1v, T + 2 * kWsTile, T + kWsTile,
                        [&](int i, bf16x8 v) { *reinterpret_cast<bf16x8*>(out + i * kNsLd) = v; });
            };
            __bf16* const TA = tiles;
            __bf16* const TB = tiles + 4 * kWsTile;
            prefetch(blockIdx.x);
            for (long k = 0; k < rounds; ++k) {
                const bool va = 2 * k < nmine, vb = 2 * k + 1 < nmine, vp = k > 0 && 2 * k - 1 < nmine;
                const long blk_a = blockIdx.x + (2 * k) * (long)gridDim.x, blk_b = blk_a + gridDim.x;
#ifdef VER_WS_ABL_NOROW
                for (int sl = 0; sl < 8; ++sl) lds_barrier();
                continue;
#endif
                if (va) r0(sa, TA, DLs);
                prefetch(blk_b);                                    // B's block of this round: step 0 in slot 3
                WS_SLOT_END(k, 0);
                tl_k = k; tl_base = 4;
                if (vp) r6(sb, TB);
                WS_SLOT_END(k, 1);
                if (va) r2(sa, TA);
                WS_SLOT_END(k, 2);
                if (vb) r0(sb, TB, DLs + kWsRows * kNsDlLd);
                prefetch(blk_a + 2 * (long)gridDim.x);              // A's block of the next round: step 0 in slot 0
                WS_SLOT_END(k, 3);
                tl_base = 0;
                if (va) r4(sa, TA);
                tl_base = 8;
                WS_SLOT_END(k, 4);
                if (vb) r2(sb, TB);
                WS_SLOT_END(k, 5);
                if (va) r6(sa, TA);
                WS_SLOT_END(k, 6);
                if (vb) r4(sb, TB);
                WS_SLOT_END(k, 7);
            }
            WS_SPAN(1);
            return;
        }
    }
    if (!ROWS4 && team == 0) {
        // =================================================================== ROW team
        const float* sv_n = sv + 8 * g;
        const int myrow = 16 * q + c;
        struct RowState {
            bf16x8 xh1[4], xh2[4];
            float rs1, rs2;
            long r;
            bool ok;
            bf16x8 xn[4], dln;            // the NEXT block's x / d(logits) chunks of this lane, requested a round ahead
            bool okn;
        };
        RowState sa, sb;
        // a lone wave per SIMD has nothing to hide an HBM round trip behind: the operands of a block's step 0 are
        // requested one round (eight slots) earlier
        // (unconditional loads from a clamped row: loads under a branch make the compiler wait for vmcnt(0) everywhere)
        auto prefetch = [&](RowState& st, long blk, bool valid) {
            const long rn = blk * kWsRows + myrow;
            st.okn = valid && rn < N;
            const long rc = st.okn ? rn : N - 1;
#pragma unroll
            for (int t = 0; t < 4; ++t) st.xn[t] = *reinterpret_cast<const bf16x8*>(x + rc * kW + 32 * t + 8 * g);
            st.dln = *reinterpret_cast<const bf16x8*>(dlog + rc * kC + 8 * (g & 1));
        };
        // RSTD: the saved 1/std pair of a block goes straight into rs1 / rs2, requested when the previous block of this
        // state has used them for the last time (end of its step 6, two slots before the new block's step 0)
        auto prefetch_rs = [&](RowState& st, long blk) {
            if constexpr (RSTD) {
                const long rn = blk * kWsRows + myrow;
                const float2 v = *reinterpret_cast<const float2*>(rstd + 2 * (rn < N ? rn : N - 1));
                st.rs1 = v.x;
                st.rs2 = v.y;
            }
        };
        auto r0 = [&](RowState& st, __bf16* T, __bf16* DL, long blk) {          // x -> LN1 + ReLU -> h1 (T0); stage d(logits)
            st.r = blk * kWsRows + myrow;
            st.ok = st.r < N;
            bf16x8 xr[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) xr[t] = st.okn ? st.xn[t] : z8;
            if (g < 2) {
                bf16x8 dl = st.okn ? st.dln : z8;
                if (gscale != 1.0f) {            // (wave-uniform) d(logits) arrives unscaled: ver_focal_loss_forward_grad
                    const f32x4 lo = __builtin_convertvector(__builtin_shufflevector(dl, dl, 0, 1, 2, 3), f32x4) * gscale;
                    const f32x4 hi = __builtin_convertvector(__builtin_shufflevector(dl, dl, 4, 5, 6, 7), f32x4) * gscale;
                    dl = pack8(lo, hi);
                }
                *reinterpret_cast<bf16x8*>(DL + myrow * kNsDlLd + 8 * g) = dl;
            }
            prefetch(st, blk + 2 * (long)gridDim.x, blk + 2 * (long)gridDim.x < nblk);
            ln_relu_nat<true, true, CENTERED, RSTD>(xr, sv_n + kW, sv_n + 2 * kW, eps, st.xh1, st.rs1);
#pragma unroll
            for (int t = 0; t < 4; ++t) *reinterpret_cast<bf16x8*>(T + myrow * kNsLd + 32 * t + 8 * g) = xr[t];
        };
        auto r2 = [&](RowState& st, __bf16* T) {                                  // a2 (T1) -> LN2 + ReLU -> h2 (T2)
            bf16x8 xr[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) xr[t] = *reinterpret_cast<const bf16x8*>(T + kWsTile + myrow * kNsLd + 32 * t + 8 * g);
            ln_relu_nat<true, true, CENTERED, RSTD>(xr, sv_n + 4 * kW, sv_n + 5 * kW, eps, st.xh2, st.rs2);
#pragma unroll
            for (int t = 0; t < 4; ++t) *reinterpret_cast<bf16x8*>(T + 2 * kWsTile + myrow * kNsLd + 32 * t + 8 * g) = xr[t];
        };
        auto r4 = [&](RowState& st, __bf16* T) {                                  // d(h2) (T1) -> LN2 bwd -> d(a2) T3, d(z2) T2, d(z2)n2 T1
            f32x4 d[8];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const bf16x8 v = *reinterpret_cast<const bf16x8*>(T + kWsTile + myrow * kNsLd + 32 * t + 8 * g);
                d[2 * t] = unpack_half(v, 0);
                d[2 * t + 1] = unpack_half(v, 1);
            }
            ns_ln_relu_bwd<true, true>(d, st.xh2, st.rs2, sv_n + 4 * kW, sv_n + 5 * kW, T + 3 * kWsTile + myrow * kNsLd + 8 * g,
                           T + 2 * kWsTile + myrow * kNsLd + 8 * g, T + kWsTile + myrow * kNsLd + 8 * g, true);
        };
        auto r6 = [&](RowState& st, __bf16* T, long blk_next) {                   // d(h1) (T4) -> LN1 bwd -> d(x); d(z1) T2, n1 T1
            f32x4 d[8];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const bf16x8 v = *reinterpret_cast<const bf16x8*>(T4 + myrow * kNsLd + 32 * t + 8 * g);
                d[2 * t] = unpack_half(v, 0);
                d[2 * t + 1] = unpack_half(v, 1);
            }
            ns_ln_relu_bwd<true, true>(d, st.xh1, st.rs1, sv_n + kW, sv_n + 2 * kW, dx + st.r * kW + 8 * g, T + 2 * kWsTile + myrow * kNsLd + 8 * g,
                           T + kWsTile + myrow * kNsLd + 8 * g, st.ok);
            prefetch_rs(st, blk_next < nblk ? blk_next : nblk - 1);
        };
        __bf16* const TA = tiles;
        __bf16* const TB = tiles + 4 * kWsTile;
        prefetch(sa, blockIdx.x, blockIdx.x < nblk);
        prefetch(sb, blockIdx.x + (long)gridDim.x, blockIdx.x + (long)gridDim.x < nblk);
        prefetch_rs(sa, blockIdx.x < nblk ? (long)blockIdx.x : nblk - 1);
        prefetch_rs(sb, blockIdx.x + (long)gridDim.x < nblk ? blockIdx.x + (long)gridDim.x : nblk - 1);
        for (long k = 0; k < rounds; ++k) {
            const bool va = 2 * k < nmine, vb = 2 * k + 1 < nmine, vp = k > 0 && 2 * k - 1 < nmine;
            const long blk_a = blockIdx.x + (2 * k) * (long)gridDim.x, blk_b = blk_a + gridDim.x;
#ifdef VER_WS_ABL_NOROW
            (void)blk_a; (void)blk_b; (void)va; (void)vb; (void)vp;
            for (int sl = 0; sl < 8; ++sl) lds_barrier();
#else
            if (va) r0(sa, TA, DLs, blk_a);
            lds_barrier();
            if (vp) r6(sb, TB, blk_b);
            lds_barrier();
            if (va) r2(sa, TA);
            lds_barrier();
            if (vb) r0(sb, TB, DLs + kWsRows * kNsDlLd, blk_b);
            lds_barrier();
            if (va) r4(sa, TA);
            lds_barrier();
            if (vb) r2(sb, TB);
            lds_barrier();
            if (va) r6(sa, TA, blk_a + 2 * (long)gridDim.x);
            lds_barrier();
            if (vb) r4(sb, TB);
            lds_barrier();
#endif
        }
        return;
    }
    // ======================================================================= FEATURE team
    const int f0 = 32 * q;
    bf16x8 wa2[OT][4], wb2[OT][4], wa3[OT];
#pragma unroll
    for (int ot = 0; ot < OT; ++ot) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const float* p = W2 + (size_t)(f0 + 16 * ot + c) * kW + 32 * ks + 8 * g;
            wa2[ot][ks] = pack8(*reinterpret_cast<const f32x4*>(p), *reinterpret_cast<const f32x4*>(p + 4));
            f32x4 lo, hi;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                lo[j] = W2[(size_t)(32 * ks + 8 * g + j) * kW + f0 + 16 * ot + c];
                hi[j] = W2[(size_t)(32 * ks + 8 * g + 4 + j) * kW + f0 + 16 * ot + c];
            }
            wb2[ot][ks] = pack8(lo, hi);
        }
        f32x4 lo = zero4, hi = zero4;
        if (g < 2) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                lo[j] = W3[(size_t)(8 * g + j) * kW + f0 + 16 * ot + c];
                hi[j] = W3[(size_t)(8 * g + 4 + j) * kW + f0 + 16 * ot + c];
            }
        }
        wa3[ot] = pack8(lo, hi);
    }
    bf16x8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (__bf16)1.0f;
    f32x4 dw2[OT][8], dw3[OT], sdb3 = zero4, sbet1[OT], sgam1[OT], sbet2[OT], sgam2[OT], sb2[OT];
#pragma unroll
    for (int ot = 0; ot < OT; ++ot) {
        dw3[ot] = sbet1[ot] = sgam1[ot] = sbet2[ot] = sgam2[ot] = sb2[ot] = zero4;
#pragma unroll
        for (int kt = 0; kt < 8; ++kt) dw2[ot][kt] = zero4;
    }
    // The product steps of the feature team are LATENCY bound, and the register file is why: 188 of a wave's 256 registers hold
    // weights and accumulators for the whole kernel, and with what was left hipcc kept ONE operand fragment in flight in steps 1
    // and 5 ("ds_read_b128, s_waitcnt lgkmcnt(0), two MFMAs" sixteen times: 2.6 k cycles for 0.5 k cycles of matrix work).
    // Round 6: (a) the fully pipelined form (all four k-step fragments of a 16-row tile requested together, the next tile's before
    // the current one is multiplied; b2 added to the finished tile) compiles as written, runs step 1 in 2.15 k -- LDS contention
    // with the row team's step keeps it there -- and its 40 extra registers spill accumulators to scratch INSIDE the round:
    // 24.1 -> 31.4 ms; (b) a ring of THREE fragments (two requested ahead) fits: steps 1 / 5 2.6 / 2.5 -> 2.2 / 2.05 k,
    // 24.1 -> 23.1 ms; (c) the same ring for the transposed fragments of d(W2) in step 7 spills again and is not used.
    // (what fits: a ring of kWsAhead + 1 fragments -- the fragment of k-step i + kWsAhead is requested before k-step i is multiplied)
    auto rows_product = [&](const __bf16* src, __bf16* dst, const bf16x8 (&w)[OT][4], const float* init) {
        constexpr int NF = 4 * RT, RING = kWsAhead + 1;
        bf16x8 b[RING];
        auto frag = [&](int i) { return *reinterpret_cast<const bf16x8*>(src + (16 * (i >> 2) + c) * kNsLd + 32 * (i & 3) + 8 * g); };
#pragma unroll
        for (int i = 0; i < kWsAhead; ++i) b[i] = frag(i);
        f32x4 acc[OT], bias[OT];
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            const int rt = i >> 2, ks = i & 3;
            if (i + kWsAhead < NF) b[(i + kWsAhead) % RING] = frag(i + kWsAhead);
            // (the bias is ADDED to the finished tile, not the accumulators' start value: requested here it has four k-steps to
            //  arrive; as the start value every tile began with an exposed LDS round trip)
            if (ks == 0 && init) {
#pragma unroll
                for (int ot = 0; ot < OT; ++ot) bias[ot] = *reinterpret_cast<const f32x4*>(init + 16 * ot);
            }
            __builtin_amdgcn_sched_barrier(0);
            const bool late = kWsBiasLate || !init;
#pragma unroll
            for (int ot = 0; ot < OT; ++ot) acc[ot] = mfma(w[ot][ks], b[i % RING], ks == 0 ? (late ? zero4 : bias[ot]) : acc[ot]);
            if (ks == 3) {
#pragma unroll
                for (int ot = 0; ot < OT; ++ot)
                    *reinterpret_cast<bf16x4*>(dst + (16 * rt + c) * kNsLd + f0 + 16 * ot + 4 * g) =
                        pack4(init && kWsBiasLate ? acc[ot] + bias[ot] : acc[ot]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto f1 = [&](__bf16* T) {                                            // a2 = W2 h1 + b2: T0 -> T1
        rows_product(T, T + kWsTile, wa2, sv + 3 * kW + f0 + 4 * g);
    };
    auto f3 = [&](__bf16* T, __bf16* DL) {                                 // d(h2) = W3^T d(logits) -> T1; d(W3), d(b3) from DL, T2
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const bf16x8 b = g < 2 ? *reinterpret_cast<const bf16x8*>(DL + (16 * rt + c) * kNsDlLd + 8 * g) : z8;
#pragma unroll
            for (int ot = 0; ot < OT; ++ot) {
                const f32x4 d = mfma(wa3[ot], b, zero4);
                *reinterpret_cast<bf16x4*>(T + kWsTile + (16 * rt + c) * kNsLd + f0 + 16 * ot + 4 * g) = pack4(d);
            }
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const bf16x8 a = ns_tr_frag(DL, kNsDlLd, 32 * ks, 0, c, g);
            if (q == 0) sdb3 = mfma(a, ones, sdb3);
#pragma unroll
            for (int ft = 0; ft < OT; ++ft) dw3[ft] = mfma(a, ns_tr_frag(T + 2 * kWsTile, kNsLd, 32 * ks, f0 + 16 * ft, c, g), dw3[ft]);
        }
    };
    auto f5 = [&](__bf16* T) {                                            // LN2 row sums, d(h1) = W2^T d(a2) -> T4
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            bf16x8 ada[OT];
#pragma unroll
            for (int ot = 0; ot < OT; ++ot) {
                ada[ot] = ns_tr_frag(T + 3 * kWsTile, kNsLd, 32 * ks, f0 + 16 * ot, c, g);
#ifndef VER_WS_ABL_NOSUMS
                sb2[ot] = mfma(ada[ot], ones, sb2[ot]);
                const bf16x8 adz = ns_tr_frag(T + 2 * kWsTile, kNsLd, 32 * ks, f0 + 16 * ot, c, g);
                sbet2[ot] = mfma(adz, ones, sbet2[ot]);
                // d(gamma2)[f] = sum_r d(z2)[r][f] n2[r][f] = the DIAGONAL of d(z2)^T n2 (T1 holds n2 itself)
                sgam2[ot] = mfma(adz, ns_tr_frag(T + kWsTile, kNsLd, 32 * ks, f0 + 16 * ot, c, g), sgam2[ot]);
#endif
            }
            if constexpr (!kWsDw2InStep7) {
#pragma unroll
                for (int kt = 0; kt < 8; ++kt) {
                    const bf16x8 b = ns_tr_frag(T, kNsLd, 32 * ks, 16 * kt, c, g);
#pragma unroll
                    for (int ot = 0; ot < OT; ++ot) dw2[ot][kt] = mfma(ada[ot], b, dw2[ot][kt]);
                    if (kt & 1) __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        rows_product(T + 3 * kWsTile, T4, wb2, nullptr);
    };
    // d(W2) += d(a2)^T h1 runs HERE, not in step 5 beside the product that step 6 waits for: nothing downstream reads it, T3
    // (d(a2)) and T0 (h1) stay untouched until the block's set starts over (step 6 puts n1 into T1), and step 7 was 0.5 k
    // cycles of a slot the row team needs 2 k for, while step 5 (3.1 k) had the row team idle for 1.5 k.
    auto f7 = [&](__bf16* T) {                                            // LN1 row sums from T2, T1; d(W2) from T3, T0
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            bf16x8 ada[OT];
#pragma unroll
            for (int ot = 0; ot < OT; ++ot) {
                if constexpr (kWsDw2InStep7) ada[ot] = ns_tr_frag(T + 3 * kWsTile, kNsLd, 32 * ks, f0 + 16 * ot, c, g);
#ifndef VER_WS_ABL_NOSUMS
                const bf16x8 adz = ns_tr_frag(T + 2 * kWsTile, kNsLd, 32 * ks, f0 + 16 * ot, c, g);
                sbet1[ot] = mfma(adz, ones, sbet1[ot]);
                sgam1[ot] = mfma(adz, ns_tr_frag(T + kWsTile, kNsLd, 32 * ks, f0 + 16 * ot, c, g), sgam1[ot]);      // diagonal, as in step 5
#endif
            }
            if constexpr (kWsDw2InStep7) {
#pragma unroll
                for (int kt = 0; kt < 8; ++kt) {
                    const bf16x8 b = ns_tr_frag(T, kNsLd, 32 * ks, 16 * kt, c, g);
#pragma unroll
                    for (int ot = 0; ot < OT; ++ot) dw2[ot][kt] = mfma(ada[ot], b, dw2[ot][kt]);
                    if (kt & 1) __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    };
    // d(x) of a block, left in T4 by the row team's step 6 (4-row mapping): wave q stores rows 16 q .. + 15, four whole
    // 256-byte rows per instruction, through a per-block buffer resource (rows past N are dropped by its size)
    auto store_dx = [&](long blk) {
        if constexpr (ROWS4) {
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            const long left = N - blk * kWsRows;
            const int rows = left < kWsRows ? (int)left : kWsRows;
            const __amdgpu_buffer_rsrc_t ro =
                __builtin_amdgcn_make_buffer_rsrc((void*)(dx + blk * (long)(kWsRows * kW)), 0, rows * kW * 2, 0x00020000);
            const int row = 16 * q + (lane >> 4), ch = lane & 15;
            bf16x8 v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const bf16x8*>(T4 + (row + 4 * i) * kNsLd + 8 * ch);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v[i]), ro, ((row + 4 * i) * kW + 8 * ch) * 2, 0, 0);
        }
    };
    {
        __bf16* const TA = tiles;
        __bf16* const TB = tiles + 4 * kWsTile;
        __bf16* const DLA = DLs;
        __bf16* const DLB = DLs + kWsRows * kNsDlLd;
        for (long k = 0; k < rounds; ++k) {
            const bool va = 2 * k < nmine, vb = 2 * k + 1 < nmine, vp = k > 0 && 2 * k - 1 < nmine;
#ifdef VER_WS_ABL_NOFEAT
            (void)va; (void)vb; (void)vp; (void)DLA; (void)DLB;
            for (int sl = 0; sl < 8; ++sl) lds_barrier();
#else
            const long blk_a = blockIdx.x + (2 * k) * (long)gridDim.x;
            if (vp) f5(TB);
            WS_SLOT_END(k, 0);
            if (va) f1(TA);
            WS_SLOT_END(k, 1);
            if (vp) f7(TB);
            WS_SLOT_END(k, 2);
            if (va) f3(TA, DLA);
            if (vp) store_dx(blk_a - (long)gridDim.x);        // B's d(x): in T4 since slot 1, T4 is rewritten in slot 5
            WS_SLOT_END(k, 3);
            if (vb) f1(TB);
            WS_SLOT_END(k, 4);
            if (va) f5(TA);
            WS_SLOT_END(k, 5);
            if (vb) f3(TB, DLB);
            WS_SLOT_END(k, 6);
            if (va) f7(TA);
            if (va) store_dx(blk_a);                          // A's d(x): in T4 since slot 6, T4 is rewritten in slot 0
            WS_SLOT_END(k, 7);
#endif
        }
    }
    WS_SPAN(1);
    // ---- parameter gradients: one atomic per element and workgroup
#pragma unroll
    for (int ot = 0; ot < OT; ++ot) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int f = f0 + 16 * ot + 4 * g + i;
#pragma unroll
            for (int kt = 0; kt < 8; ++kt) atomicAdd(pgrad + 6 * kW + kC * kW + kC + (size_t)f * kW + 16 * kt + c, dw2[ot][kt][i]);
            atomicAdd(pgrad + 6 * kW + (4 * g + i) * kW + f0 + 16 * ot + c, dw3[ot][i]);
            if (c == 0) {
                atomicAdd(pgrad + 1 * kW + f, sbet1[ot][i]);
                atomicAdd(pgrad + 4 * kW + f, sbet2[ot][i]);
                atomicAdd(pgrad + 5 * kW + f, sb2[ot][i]);
            }
            if (c == 4 * g + i) {               // D[m][n] with m == n: feature f0 + 16ot + c
                atomicAdd(pgrad + 0 * kW + f, sgam1[ot][i]);
                atomicAdd(pgrad + 3 * kW + f, sgam2[ot][i]);
            }
        }
    }
    if (q == 0 && c == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) atomicAdd(pgrad + 6 * kW + kC * kW + 4 * g + i, sdb3[i]);
    }
}

extern "C" int ver_occ_mlp_backward_fused(const void* x, const void* grad_logits, const float* W2, const float* W3,
                                          const float* vectors, void* grad_x, float* param_grads, long N, int width,
                                          int classes, float eps, const float* grad_scale, int flags, void* stream) {
    return ver_occ_mlp_backward_fused_stats(x, grad_logits, W2, W3, vectors, nullptr, grad_x, param_grads, N, width, classes,
                                            eps, grad_scale, flags, stream);
}

extern "C" int ver_occ_mlp_backward_fused_stats(const void* x, const void* grad_logits, const float* W2, const float* W3,
                                                const float* vectors, const float* rstd, void* grad_x, float* param_grads,
                                                long N, int width, int classes, float eps, const float* grad_scale, int flags,
                                                void* stream) {
    VER_REQUIRE(N >= 0, VER_EINVAL, "ver_occ_mlp_backward_fused: negative row count");
    VER_REQUIRE((flags & ~VER_OCC_MLP_CENTERED) == 0, VER_EINVAL, "ver_occ_mlp_backward_fused: unknown flags 0x%x", flags);
    VER_REQUIRE(width == kW && classes == kC, VER_EUNSUPPORTED,
                "ver_occ_mlp_backward_fused: built for width %d / %d classes (got %d / %d)", kW, kC, width, classes);
    VER_REQUIRE(W2 && W3 && vectors && param_grads, VER_EINVAL, "ver_occ_mlp_backward_fused: null pointer argument");
    hipStream_t st = (hipStream_t)stream;
    if (int zrc = ver_zero_async(param_grads, (6 * kW + kC * kW + kC + kW * kW) * sizeof(float), st)) return zrc;   // (kernel: ver_zero_async)
    hipError_t e = hipSuccess;
    if (e != hipSuccess) return ver_fail(VER_ELAUNCH, "ver_occ_mlp_backward_fused: memset: %s", hipGetErrorString(e));
    if (N == 0) return VER_OK;
    VER_REQUIRE(x && grad_logits && grad_x, VER_EINVAL, "ver_occ_mlp_backward_fused: null pointer argument");
    VER_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)grad_x & 15) == 0 && ((uintptr_t)grad_logits & 15) == 0 &&
                    ((uintptr_t)W2 & 15) == 0,
                VER_EINVAL, "ver_occ_mlp_backward_fused: buffers must be 16-byte aligned");
    static const int ws = [] {
        const char* ev = getenv("VER_OCC_MLP_WS");              // 1 (default): wave-specialised kernel; 0: phase-locked N-split
        return ev ? atoi(ev) : 1;
    }();
    if (ws) {
        // (the saved statistics replace the recomputation only on centred rows; otherwise they are ignored)
        const bool use_rstd = rstd && (flags & VER_OCC_MLP_CENTERED);
        static const int rows4 = [] {
            const char* ev = getenv("VER_OCC_MLP_ROWS4");       // 1 (default): 4 rows x 8 features per row-team lane; 0: quad mapping
            return ev ? atoi(ev) : 1;
        }();
        auto kern = (flags & VER_OCC_MLP_CENTERED)
                        ? (use_rstd ? (rows4 ? k_occ_mlp_bwd_ws<true, true, true> : k_occ_mlp_bwd_ws<true, true, false>)
                                    : k_occ_mlp_bwd_ws<true, false, false>)
                        : k_occ_mlp_bwd_ws<false, false, false>;
        e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kWsLds);
        if (e != hipSuccess) return ver_fail(VER_ELAUNCH, "ver_occ_mlp_backward_fused: LDS attribute: %s", hipGetErrorString(e));
        const long nb = (N + kWsRows - 1) / kWsRows;
        const long gridw = nb < 256 ? nb : 256;               // one workgroup per CU (LDS bound), persistent
        hipLaunchKernelGGL(kern, dim3((unsigned)gridw), dim3(512), kWsLds, st, (const __bf16*)x,
                           (const __bf16*)grad_logits, W2, W3, vectors, (__bf16*)grad_x, param_grads, N, eps, grad_scale,
                           use_rstd ? rstd : nullptr);
        return ver_check_launch("ver_occ_mlp_backward_fused");
    }
    VER_REQUIRE(!grad_scale, VER_EUNSUPPORTED, "ver_occ_mlp_backward_fused: grad_scale needs the wave-specialised kernel");
    // (VER_OCC_MLP_CENTERED is a hint, not a contract: this kernel's LayerNorm always computes the row mean, which is
    //  ~0 on centred rows -- same results, no skipped pass)
    static const int nw = [] {
        const char* ev = getenv("VER_OCC_MLP_NS_WAVES");       // 8: 128-row blocks, two waves per SIMD; 4: 64-row blocks
        const int v = ev ? atoi(ev) : 8;
        return v == 4 ? 4 : 8;
    }();
    const size_t lds = ns_lds_bytes(nw);
    const void* kern = nw == 8 ? (const void*)k_occ_mlp_bwd_ns<8> : (const void*)k_occ_mlp_bwd_ns<4>;
    e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return ver_fail(VER_ELAUNCH, "ver_occ_mlp_backward_fused: LDS attribute: %s", hipGetErrorString(e));
    const long nblk = (N + 16 * nw - 1) / (16 * nw);
    const long per_cu = nw == 4 ? 2 : 1;                   // 4 waves: 75 KB of LDS, two workgroups per CU
    const long grid = nblk < 256 * per_cu ? nblk : 256 * per_cu;          // persistent
    if (nw == 8)
        hipLaunchKernelGGL(k_occ_mlp_bwd_ns<8>, dim3((unsigned)grid), dim3(512), lds, st, (const __bf16*)x,
                           (const __bf16*)grad_logits, W2, W3, vectors, (__bf16*)grad_x, param_grads, N, eps);
    else
        hipLaunchKernelGGL(k_occ_mlp_bwd_ns<4>, dim3((unsigned)grid), dim3(256), lds, st, (const __bf16*)x,
                           (const __bf16*)grad_logits, W2, W3, vectors, (__bf16*)grad_x, param_grads, N, eps);
    return ver_check_launch("ver_occ_mlp_backward_fused");
}
