// Forward product of the occupancy head's GEMM layers on gfx950 matrix cores:
//     C[M, N] = A[M, K] * W[K, N] (+ bias[N])        (bf16 operands, fp32 accumulation, bf16 result)
// A = rows of the tap matrix of a lattice layer / of the gathered occ_proj operand (row-major, K contiguous; may be a
// column range of a wider matrix), W = the class weight matrix [K, N] as the layers keep it (row-major, N contiguous):
// dense_heads/upsample.py (_Layer0Z4, _LatticeLayerZ4, _LatticeLayer) and occ_proj_lattice.py, i.e. the reference's three
// ConvTranspose3d and occ_proj (voxelformer_occupancy_head.py:251-258, :560, :571) evaluated on the even lattice.
//
// The kernel is the sibling of k_wgrad_tn (ver_wgrad.hip: same 256 x 256 tile, 8 waves as 2 x 4, two wave groups half a
// phase apart, LDS-DMA ring, counted vmcnt, raw barriers); what differs is the operand with the contraction index on its
// FAST axis:
//   * A tile rows are streamed as [256 rows][32 k] images (64 B per row; a wave instruction moves 16 rows x 64 B), their
//     16-byte chunks XOR-ed by bits 2-3 of the row on the SOURCE address, and the MFMA fragment of a lane (8 consecutive k
//     of one row) is ONE ds_read_b128, conflict free by that swizzle;
//   * W is streamed in 16-row slabs and read through ds_read_b64_tr_b16 exactly as both operands of k_wgrad_tn;
//   * a phase is 32 of K (two MFMA k-steps: 8 + 8 fragment reads, 4 LDS-DMA pieces, 16 MFMAs per wave), the ring holds
//     four phases (128 KB), pieces are requested two phases ahead;
//   * no split over K for the tall products (K = 6 304 .. 38 400 here: 197+ phases per tile; skinny ones -- the 450- and
//     1 800-row operands of a one-viewpoint step, 12-42 tiles -- are cut into K slices with fp32 partial tiles, see
//     ver_gemm_nn_splitk); tiles are dealt to the XCDs in contiguous
//     ranges, the N tiles of one row block next to each other, so that the 32 tiles in flight on an XCD share their A
//     rows and W columns in its L2.
#include "ver_common.h"

namespace {
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;

constexpr int kTile = 256;
constexpr int kStageBytes = 32768;          // A image [256][32] (16 KB) + two W slabs [16][256] (2 x 8 KB)
constexpr int kStages = 4;
constexpr int kLdsBytes = kStages * kStageBytes;

// LDS reads as inline asm: behind the builtins the compiler waits for vmcnt(0) in front of every LDS read that follows an
// LDS-DMA (ver_wgrad.hip).  Results are valid behind the s_waitcnt lgkmcnt(0) of phase().
template <int OFF>
__device__ __forceinline__ i32x2 tr_read(int addr) {
    i32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
template <int OFF>
__device__ __forceinline__ i32x4 row_read(int addr) {
    i32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
__device__ __forceinline__ bf16x8 frag(i32x2 lo, i32x2 hi) {
    return __builtin_bit_cast(bf16x8, (i32x4)__builtin_shufflevector(lo, hi, 0, 1, 2, 3));
}

// Implicit operand (ver_gemm_nn_taps): A is never materialised -- row r of the product is cell (b, zl, y, x) of a Z = 4
// lattice (r = ((b 2 + zl) H + y) W + x), its K axis `ntaps` blocks of C channels, block t = the C-vector of the neighbouring
// cell (zl + dz[t], y + dy[t], x + dx[t]) of the SOURCE lattice (zeros outside): exactly the tap matrix ver_lattice_gather
// writes, read straight from the lattice by the LDS-DMA (a tap block of a row is one contiguous 2 C-byte vector).
struct TapArgs {
    const __bf16* lattice;
    long lattice_bytes;
    int B, H, W, C, ntaps, P;           // H, W: the combined lattice = the rows' grid; P = 2 H W rows per viewpoint
    const float* rowpos;                // [P][N] fp32 or null: added to row r's result by its position r % P
    // constant-pattern segments (kind[t] = 1): CW columns that depend on the row's POSITION r % P only -- block dz[t] of
    // cst [P][ncst][CW] bf16 (dense_heads/upsample.py: the 0/1 patterns of the bias-valued odd positions + the 1 column)
    const __bf16* cst;
    long cst_bytes;
    int CW, ncst;
    signed char kind[64], dz[64], dy[64], dx[64];
    // several source lattices of one shape `plane_elems` elements apart (the four class planes of a layer's output gradient,
    // ver_gemm_nn_segments' d(input) form): tap t reads plane pl[t]; lattice_bytes is the size of ONE plane
    signed char pl[64];
    long plane_elems;
};

struct GemmArgs {
    const __bf16* A;
    const __bf16* W;
    const float* bias;      // [N] or null
    __bf16* C;
    long lda, ldw, ldc, M;
    int K, N, tiles_n, T, per_xcd;
    int S, Kc;              // split over K (skinny products: a one-viewpoint step has 450 / 1 800 rows, 12-42 tiles):
    float* ws;              //   S slices of Kc columns, fp32 partial tiles [S][M][N], added up by k_gemm_reduce
    TapArgs t;
};

struct GemmLane {
    int offA0, offA1;       // A fragment reads of k-step 0 / 1 of a phase (stage 0, row tile 0)
    int offB0, offB1;       // W fragment reads (stage 0, slab 0, column tile 0 / 1)
    int voA, voW;           // per-lane source offsets of this wave's LDS-DMA pieces
    int stepW;              // bytes per 16-row slab of W
    int stepA16;            // bytes per 16 rows of A (second LDS-DMA piece of a wave)
    int dmaA, dmaW;         // offsets of this wave's pieces inside a stage
};

// byte offset of the C-vector of cell (b, z = zl + dz, y, x) in a source lattice of layout L (ver_lattice_gather's layouts:
// 0 plain [B,4,H,W,C], 2 z-split [B,2,H,W,2,C], 3 planar z-split [4,B,2,H/2,W/2,2,C]); outside the lattice: an offset
// beyond every buffer range (the DMA then writes zeros).  All lattices here are < 2 GiB (checked by the launcher).
constexpr int kOutside = (int)0x80000000;
template <int L>
__device__ __forceinline__ int cell_off(const TapArgs& t, int b, int zl, int y, int x, int dz) {
    if ((unsigned)y >= (unsigned)t.H || (unsigned)x >= (unsigned)t.W || b < 0) return kOutside;
    const int j = dz >> 1;                                  // z = zl + dz, dz in {0, 2}: same zl, upper half j
    int v;
    if (L == 0)
        v = ((b * 4 + zl + dz) * t.H + y) * t.W + x;
    else if (L == 2)
        v = ((((b * 2 + zl) * t.H + y) * t.W + x) << 1) + j;
    else
        v = (((((((y & 1) << 1 | (x & 1)) * t.B + b) * 2 + zl) * (t.H >> 1) + (y >> 1)) * (t.W >> 1) + (x >> 1)) << 1) + j;
    return v * t.C * 2;
}

struct TapLane {                // implicit operand: this lane's two rows and the running segment / channel of its DMA stream
    int b1, zl1, y1, x1, b2, zl2, y2, x2;
    int pos1, pos2;             // r % P of the two rows (constant-pattern segments)
    int vo1, vo2;               // byte offsets of the current segment's vectors (+ this lane's 16-byte chunk)
    int chunk;                  // pch * 16
    int tap, ch;                // wave-uniform: segment and byte offset inside it of the NEXT piece to request
    int seglen, is_cst;         // wave-uniform: bytes of the current segment; it is read from the pattern table
};

template <int L>
__device__ __forceinline__ void tap_offsets(const TapArgs& t, TapLane& tl, __amdgpu_buffer_rsrc_t& ra) {
    if (t.plane_elems && tl.tap < t.ntaps && !t.kind[tl.tap])
        ra = __builtin_amdgcn_make_buffer_rsrc((void*)(t.lattice + (long)t.pl[tl.tap] * t.plane_elems), 0,
                                               (int)max(0L, min(t.lattice_bytes, 0x7FFFFFFFL)), 0x00020000);
    tl.is_cst = 0;
    tl.seglen = 2 * t.C;
    if (tl.tap < t.ntaps) {
        const int dz = t.dz[tl.tap], dy = t.dy[tl.tap], dx = t.dx[tl.tap];
        if (t.kind[tl.tap]) {
            tl.is_cst = 1;
            tl.seglen = 2 * t.CW;
            tl.vo1 = tl.b1 < 0 ? kOutside : (tl.pos1 * t.ncst + dz) * t.CW * 2 + tl.chunk;
            tl.vo2 = tl.b2 < 0 ? kOutside : (tl.pos2 * t.ncst + dz) * t.CW * 2 + tl.chunk;
        } else {
            const int o1 = cell_off<L>(t, tl.b1, tl.zl1, tl.y1 + dy, tl.x1 + dx, dz);
            const int o2 = cell_off<L>(t, tl.b2, tl.zl2, tl.y2 + dy, tl.x2 + dx, dz);
            tl.vo1 = o1 == kOutside ? kOutside : o1 + tl.chunk;
            tl.vo2 = o2 == kOutside ? kOutside : o2 + tl.chunk;
        }
    } else {
        tl.vo1 = tl.vo2 = kOutside;                          // (pieces requested past the last phase: nobody reads them)
    }
}

// this wave's two A pieces of one phase: explicit operand (IMPL < 0) or straight from the lattice
template <int IMPL>
__device__ __forceinline__ void request_a(char* dst, const GemmLane& c, __amdgpu_buffer_rsrc_t& ra, __amdgpu_buffer_rsrc_t rcst,
                                          int& soA, const TapArgs& t, TapLane& tl) {
    if constexpr (IMPL < 0) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void*)dst, 16, c.voA, soA, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void*)(dst + 1024), 16, c.voA, soA + c.stepA16, 0, 0);
        soA += 64;
    } else {
        const int ch = __builtin_amdgcn_readfirstlane(tl.ch);
        if (__builtin_amdgcn_readfirstlane(tl.is_cst)) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rcst, (lds_void*)dst, 16, tl.vo1, ch, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rcst, (lds_void*)(dst + 1024), 16, tl.vo2, ch, 0, 0);
        } else {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void*)dst, 16, tl.vo1, ch, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void*)(dst + 1024), 16, tl.vo2, ch, 0, 0);
        }
        tl.ch = ch + 64;
        if (tl.ch == __builtin_amdgcn_readfirstlane(tl.seglen)) {    // next segment: the lane's two source vectors move
            tl.ch = 0;
            tl.tap = __builtin_amdgcn_readfirstlane(tl.tap) + 1;
            tap_offsets<IMPL>(t, tl, ra);
        }
    }
}


template <int ST, int IMPL>
__device__ __forceinline__ void phase(char* lds, f32x16 (&acc)[4][2], const GemmLane& c, __amdgpu_buffer_rsrc_t& ra,
                                      __amdgpu_buffer_rsrc_t rw, int& soA, int& soW, const TapArgs& t, TapLane& tl,
                                      __amdgpu_buffer_rsrc_t rcst) {
    // (the offset field of a DS instruction has 16 bits: stages 2-3 go through base registers 64 KiB up)
    constexpr int SB = (ST & 1) * kStageBytes;
    constexpr int UP = ST >= 2 ? 65536 : 0;
    const int a0 = c.offA0 + UP, a1 = c.offA1 + UP, b0 = c.offB0 + UP, b1 = c.offB1 + UP;
    i32x4 av[2][4];
    i32x2 bl[2][2], bh[2][2];
    // k-step 0: A chunks (l >> 5), W slab 0; k-step 1: A chunks 2 + (l >> 5), W slab 1
    av[0][0] = row_read<SB>(a0);
    av[0][1] = row_read<SB + 2048>(a0);
    bl[0][0] = tr_read<SB + 16384>(b0);
    bh[0][0] = tr_read<SB + 16384 + 512>(b0);
    bl[0][1] = tr_read<SB + 16384>(b1);
    bh[0][1] = tr_read<SB + 16384 + 512>(b1);
    av[0][2] = row_read<SB + 4096>(a0);
    av[0][3] = row_read<SB + 6144>(a0);
    av[1][0] = row_read<SB>(a1);
    av[1][1] = row_read<SB + 2048>(a1);
    bl[1][0] = tr_read<SB + 24576>(b0);
    bh[1][0] = tr_read<SB + 24576 + 512>(b0);
    bl[1][1] = tr_read<SB + 24576>(b1);
    bh[1][1] = tr_read<SB + 24576 + 512>(b1);
    av[1][2] = row_read<SB + 4096>(a1);
    av[1][3] = row_read<SB + 6144>(a1);
    // LDS-DMA of the phase two ahead: this wave's 2 x 16 rows of the A image and its piece of each W slab
    constexpr int DS = ((ST + 2) % kStages) * kStageBytes;
    request_a<IMPL>(lds + DS + c.dmaA, c, ra, rcst, soA, t, tl);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void*)(lds + DS + 16384 + c.dmaW), 16, c.voW, soW, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void*)(lds + DS + 24576 + c.dmaW), 16, c.voW, soW + c.stepW, 0, 0);
    soW += 2 * c.stepW;
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");       // this wave's pieces of the NEXT phase have landed
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(av[0][0]), "+v"(av[0][1]), "+v"(av[0][2]), "+v"(av[0][3]), "+v"(av[1][0]), "+v"(av[1][1]), "+v"(av[1][2]),
                   "+v"(av[1][3]), "+v"(bl[0][0]), "+v"(bh[0][0]), "+v"(bl[0][1]), "+v"(bh[0][1]), "+v"(bl[1][0]), "+v"(bh[1][0]),
                   "+v"(bl[1][1]), "+v"(bh[1][1])
                 :
                 : "memory");
    bf16x8 a[2][4], b[2][2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
        for (int i = 0; i < 4; ++i) a[u][i] = __builtin_bit_cast(bf16x8, av[u][i]);
        b[u][0] = frag(bl[u][0], bh[u][0]);
        b[u][1] = frag(bl[u][1], bh[u][1]);
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int it = 0; it < 4; ++it)
#pragma unroll
            for (int jt = 0; jt < 2; ++jt)
                acc[it][jt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[u][it], b[u][jt], acc[it][jt], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
}

template <int IMPL>
__global__ __launch_bounds__(512) void k_gemm_nn(GemmArgs p) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    // block b runs on XCD b % 8; an XCD works through a contiguous range of tiles (N tiles of a row block adjacent)
    const int item = (int)(blockIdx.x & 7) * p.per_xcd + (int)(blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= p.per_xcd || item >= p.T * p.S) return;
    const int split = item / p.T, tile = item - split * p.T;
    const int mt = tile / p.tiles_n, nt = tile - mt * p.tiles_n;
    const long row0 = (long)mt * kTile;
    const int k0 = split * p.Kc, klen = min(p.Kc, p.K - k0);
    const int nphase = klen / 32;

    // A: rows row0 .. row0 + 255 (past M: zeros by the buffer range), this slice's columns of the operand -- or the whole
    // source lattice (implicit operand: the per-lane offsets address the tap vectors)
    const __bf16* ab = IMPL < 0 ? p.A + row0 * p.lda + k0 : p.t.lattice;
    const long abytes = IMPL < 0 ? ((p.M - row0 - 1) * p.lda + klen) * 2 : p.t.lattice_bytes;
    __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)ab, 0, (int)max(0L, min(abytes, 0xFFFFFFFFL)), 0x00020000);
    const __amdgpu_buffer_rsrc_t rcst = __builtin_amdgcn_make_buffer_rsrc((void*)(IMPL < 0 ? ab : p.t.cst), 0,
                                                                          (int)(IMPL < 0 ? 0L : max(0L, min(p.t.cst_bytes, 0x7FFFFFFFL))), 0x00020000);
    // W: columns nt * 256 .., the slice's rows (behind the last element: zeros)
    const __bf16* wb = p.W + (long)k0 * p.ldw + (long)nt * kTile;
    const long wbytes = ((long)(klen - 1) * p.ldw + p.N - (long)nt * kTile) * 2;
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)wb, 0, (int)max(0L, min(wbytes, 0xFFFFFFFFL)), 0x00020000);
    GemmLane c;
    {
        // A image piece: 16 rows x 64 B per instruction; lane -> (row l >> 2, position l & 3), position p holds source
        // chunk p ^ ((row >> 2) & 3).  This wave moves rows 32 w .. 32 w + 31 (two instructions, + 16 rows = + 1 KiB).
        const int prow = 32 * wave + (lane >> 2);
        const int pch = (lane & 3) ^ ((prow >> 2) & 3);
        c.voA = (int)(prow * p.lda * 2) + pch * 16;
        c.dmaA = wave * 2048;
        c.stepA16 = (int)(16 * p.lda * 2);          // (second instruction: rows + 16; same swizzle term)
        // W slab piece: 8 rows x 128 B, chunks XOR-ed by bit 1 of the row (ver_wgrad.hip)
        const int wrow = lane >> 3, wch = (lane & 7) ^ (((wrow >> 1) & 1) << 2);
        c.voW = (int)(((8 * (wave >> 2) + wrow) * p.ldw + 64 * (wave & 3)) * 2) + wch * 16;
        c.stepW = (int)(16 * p.ldw * 2);
        c.dmaW = wave * 1024;
    }
    // fragment reads.  A (row mode): lane (i = l & 31, k half = l >> 5) reads chunk (2 step + (l >> 5)) ^ ((i >> 2) & 3)
    // of row 128 wr + 32 it + i;  W (transposing): as in ver_wgrad.hip
    {
        const int lbase = (int)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
        const int i = lane & 31, sw = (i >> 2) & 3;
        c.offA0 = lbase + (128 * wr + i) * 64 + (((lane >> 5) ^ sw) << 4);
        c.offA1 = c.offA0 ^ 32;
        const int q = lane >> 4, cl = lane & 15;
        const int lowch = (2 * (q & 1) + ((cl & 3) >> 1)) ^ (((cl >> 3) & 1) << 2);
        c.offB0 = lbase + ((q >> 1) * 32 + (cl >> 2)) * 128 + lowch * 16 + (cl & 1) * 8 + wc * 1024;
        c.offB1 = c.offB0 ^ 64;
    }
    TapLane tl = {};
    if constexpr (IMPL >= 0) {
        // this lane's two rows of the tile -> cells (b, zl, y, x); rows past M: no cell (every tap outside)
        const int prow = 32 * wave + (lane >> 2);
        const int hw = p.t.H * p.t.W;
        auto decode = [&](long r, int& b, int& zl, int& y, int& x) {
            if (r >= p.M) {
                b = -1, zl = y = x = 0;
                return;
            }
            b = (int)(r / p.t.P);
            const int pos = (int)(r - (long)b * p.t.P);
            zl = pos / hw;
            const int rem = pos - zl * hw;
            y = rem / p.t.W;
            x = rem - y * p.t.W;
        };
        decode(row0 + prow, tl.b1, tl.zl1, tl.y1, tl.x1);
        decode(row0 + prow + 16, tl.b2, tl.zl2, tl.y2, tl.x2);
        tl.pos1 = (tl.zl1 * p.t.H + tl.y1) * p.t.W + tl.x1;
        tl.pos2 = (tl.zl2 * p.t.H + tl.y2) * p.t.W + tl.x2;
        tl.chunk = ((lane & 3) ^ ((prow >> 2) & 3)) * 16;
        tl.tap = 0, tl.ch = 0;
        tap_offsets<IMPL>(p.t, tl, ra);
    }
    f32x16 acc[4][2];
#pragma unroll
    for (int it = 0; it < 4; ++it)
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[it][jt][r] = 0.0f;

    int soA = 0, soW = 0;
    // prologue: phases 0 and 1 in flight, phase 0 landed
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        request_a<IMPL>(lds + s * kStageBytes + c.dmaA, c, ra, rcst, soA, p.t, tl);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void*)(lds + s * kStageBytes + 16384 + c.dmaW), 16, c.voW, soW, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void*)(lds + s * kStageBytes + 24576 + c.dmaW), 16, c.voW, soW + c.stepW, 0, 0);
        soW += 2 * c.stepW;
    }
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();          // the second wave group runs half a phase behind

    int s = 0;
    for (; s + kStages <= nphase; s += kStages) {
        phase<0, IMPL>(lds, acc, c, ra, rw, soA, soW, p.t, tl, rcst);
        phase<1, IMPL>(lds, acc, c, ra, rw, soA, soW, p.t, tl, rcst);
        phase<2, IMPL>(lds, acc, c, ra, rw, soA, soW, p.t, tl, rcst);
        phase<3, IMPL>(lds, acc, c, ra, rw, soA, soW, p.t, tl, rcst);
    }
    const int rem = nphase - s;
    if (rem > 0) phase<0, IMPL>(lds, acc, c, ra, rw, soA, soW, p.t, tl, rcst);
    if (rem > 1) phase<1, IMPL>(lds, acc, c, ra, rw, soA, soW, p.t, tl, rcst);
    if (rem > 2) phase<2, IMPL>(lds, acc, c, ra, rw, soA, soW, p.t, tl, rcst);
    if (wr == 0) __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    if (p.S > 1) {
        // partial tile of this K slice -> workspace (fp32, [S][M][N]); bias and rounding happen in k_gemm_reduce
        float* out = p.ws + (long)split * p.M * p.N;
        const long i0 = row0 + 128 * wr + 4 * (lane >> 5);
        const int j0 = nt * kTile + 64 * wc + (lane & 31);
#pragma unroll
        for (int it = 0; it < 4; ++it)
#pragma unroll
            for (int jt = 0; jt < 2; ++jt) {
                const int j = j0 + 32 * jt;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const long i = i0 + 32 * it + (r & 3) + 8 * (r >> 2);
                    if (i < p.M && j < p.N) out[i * p.N + j] = acc[it][jt][r];
                }
            }
        return;
    }
    // C tile: register r of tile (it, jt) is row (r & 3) + 8 (r >> 2) + 4 (lane >> 5), column lane & 31.  Buffer stores:
    // one 32-bit lane offset + a scalar row offset per store; rows past M fall outside the range and are dropped, columns
    // past N are sent there on purpose.
    __bf16* cb = p.C + row0 * p.ldc + (long)nt * kTile;
    const long cbytes = ((p.M - row0 - 1) * p.ldc + p.N - (long)nt * kTile) * 2;
    const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc((void*)cb, 0, (int)max(0L, min(cbytes, 0xFFFFFFFFL)), 0x00020000);
    const int ldc2 = (int)(p.ldc * 2);
    const int vbase = (128 * wr + 4 * (lane >> 5)) * ldc2 + (64 * wc + (lane & 31)) * 2;
    // implicit operand: + rowpos[row % P][j], the position-dependent constant part of the row's result (rows of a tile are
    // consecutive: one conditional subtraction per row instead of a division)
    const bool haspos = IMPL >= 0 && p.t.rowpos != nullptr;
    const int pos0 = haspos ? (int)((row0 + 128 * wr + 4 * (lane >> 5)) % p.t.P) : 0;
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) {
        const int j = nt * kTile + 64 * wc + 32 * jt + (lane & 31);
        const float bj = (p.bias && j < p.N) ? p.bias[j] : 0.0f;
        const int vo = j < p.N ? vbase + 64 * jt : -2;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            float add[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                add[r] = bj;
                if (haspos) {
                    int pp = pos0 + 32 * it + (r & 3) + 8 * (r >> 2);
                    while (pp >= p.t.P) pp -= p.t.P;          // (at most once for P >= 256: the step's lattices have 450+)
                    const long row = row0 + 128 * wr + 4 * (lane >> 5) + 32 * it + (r & 3) + 8 * (r >> 2);
                    if (j < p.N && row < p.M) add[r] += p.t.rowpos[(long)pp * p.N + j];
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const __bf16 v = (__bf16)(acc[it][jt][r] + add[r]);
                __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(short, v), rc, vo, (32 * it + (r & 3) + 8 * (r >> 2)) * ldc2, 0);
            }
        }
    }
}
// C[i][j] (bf16, row pitch ldc) = sum over the S partial products + bias[j]; 4 columns per thread (N % 4 == 0)
__global__ __launch_bounds__(256) void k_gemm_reduce(const float* __restrict__ ws, const float* __restrict__ bias,
                                                     __bf16* __restrict__ c, long ldc, int S, long M, int N) {
    const long n4 = M * N / 4;
    const long stride = M * N;
    for (long e = blockIdx.x * 256L + threadIdx.x; e < n4; e += (long)gridDim.x * 256) {
        float4 acc = *reinterpret_cast<const float4*>(ws + 4 * e);
        for (int s = 1; s < S; ++s) {
            const float4 v = *reinterpret_cast<const float4*>(ws + s * stride + 4 * e);
            acc.x += v.x, acc.y += v.y, acc.z += v.z, acc.w += v.w;
        }
        const long i = (4 * e) / N;
        const int j = (int)(4 * e - i * N);
        if (bias) acc.x += bias[j], acc.y += bias[j + 1], acc.z += bias[j + 2], acc.w += bias[j + 3];
        typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
        bf16x4 v;
        v.x = (__bf16)acc.x, v.y = (__bf16)acc.y, v.z = (__bf16)acc.z, v.w = (__bf16)acc.w;
        *reinterpret_cast<bf16x4*>(c + i * ldc + j) = v;
    }
}

// K slices of a skinny product: enough workgroups to fill the chip (>= ~192), slices of at least 512 columns (16 phases: the
// fill / drain of the ring is ~6), whole phases of 32
int pick_k_splits(long M, int K, int N) {
    const long tiles = ((M + kTile - 1) / kTile) * ((N + kTile - 1) / kTile);
    if (tiles >= 96 || K < 2048 || N % 4) return 1;
    int s = (int)((224 + tiles - 1) / tiles);
    if (s > K / 512) s = K / 512;
    if (s > 64) s = 64;
    return s < 2 ? 1 : s;
}
}  // namespace

extern "C" int ver_gemm_nn_splits(long M, int K, int N) {
    if (M <= 0 || K <= 0 || N <= 0) return 1;
    return pick_k_splits(M, K, N);
}

extern "C" int ver_gemm_nn_splitk(const void* a, long lda, const void* w, long ldw, const float* bias, void* c, long ldc, long M,
                                  int K, int N, int splits, void* workspace, long workspace_bytes, void* stream);

extern "C" int ver_gemm_nn(const void* a, long lda, const void* w, long ldw, const float* bias, void* c, long ldc, long M,
                           int K, int N, int flags, void* stream) {
    VER_REQUIRE(flags == 0, VER_EINVAL, "ver_gemm_nn: unknown flags 0x%x", flags);
    return ver_gemm_nn_splitk(a, lda, w, ldw, bias, c, ldc, M, K, N, 1, nullptr, 0, stream);
}

extern "C" int ver_gemm_nn_splitk(const void* a, long lda, const void* w, long ldw, const float* bias, void* c, long ldc, long M,
                                  int K, int N, int splits, void* workspace, long workspace_bytes, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    VER_REQUIRE(M >= 0 && K > 0 && N > 0, VER_EINVAL, "ver_gemm_nn: bad sizes M=%ld K=%d N=%d", M, K, N);
    if (M == 0) return VER_OK;
    VER_REQUIRE(a && w && c, VER_EINVAL, "ver_gemm_nn: null pointer argument");
    VER_REQUIRE(splits >= 1 && splits <= 1024, VER_EINVAL, "ver_gemm_nn: %d K slices", splits);
    VER_REQUIRE(splits == 1 || (workspace && workspace_bytes >= (long)splits * M * N * (long)sizeof(float) && N % 4 == 0 && ldc % 4 == 0 &&
                                ((uintptr_t)c & 7) == 0 && ((uintptr_t)workspace & 15) == 0),
                VER_EINVAL, "ver_gemm_nn: %d K slices need an fp32 workspace of %ld bytes, N and ldc multiples of 4", splits,
                (long)splits * M * N * (long)sizeof(float));
    VER_REQUIRE(lda >= K && ldw >= N && ldc >= N, VER_EINVAL, "ver_gemm_nn: row pitch smaller than the row");
    VER_REQUIRE(K % 32 == 0 && K >= 64, VER_EUNSUPPORTED, "ver_gemm_nn: K = %d must be a multiple of 32 (>= 64)", K);
    VER_REQUIRE(lda % 8 == 0 && ldw % 8 == 0 && ((uintptr_t)a & 15) == 0 && ((uintptr_t)w & 15) == 0, VER_EUNSUPPORTED,
                "ver_gemm_nn: operands must be 16-byte aligned with row pitches that are multiples of 8 elements");
    VER_REQUIRE(256L * lda * 2 + K * 2L < 0x7FFFFFFFL && (long)K * ldw * 2 < 0xFFFFFFFFL && 256L * ldc * 2 < 0x7FFFFFFFL,
                VER_EUNSUPPORTED, "ver_gemm_nn: a tile's operand range exceeds the 32-bit offsets");
    GemmArgs p;
    p.A = (const __bf16*)a;
    p.W = (const __bf16*)w;
    p.bias = bias;
    p.C = (__bf16*)c;
    p.lda = lda;
    p.ldw = ldw;
    p.ldc = ldc;
    p.M = M;
    p.K = K;
    p.N = N;
    p.tiles_n = (N + kTile - 1) / kTile;
    p.T = (int)((M + kTile - 1) / kTile) * p.tiles_n;
    // whole phases per slice; the last slice takes what is left (K % 32 == 0)
    p.Kc = splits == 1 ? K : ((K + splits - 1) / splits + 31) / 32 * 32;
    p.S = splits == 1 ? 1 : (K + p.Kc - 1) / p.Kc;
    p.ws = (float*)workspace;
    p.per_xcd = (p.T * p.S + 7) / 8;
    p.t = TapArgs{};
    hipError_t e = hipFuncSetAttribute((const void*)k_gemm_nn<-1>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
    if (e != hipSuccess) return ver_fail(VER_ELAUNCH, "ver_gemm_nn: LDS attribute: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(k_gemm_nn<-1>, dim3((unsigned)(8 * p.per_xcd)), dim3(512), kLdsBytes, st, p);
    if (p.S > 1) {
        long grid = (M * N / 4 + 255) / 256;
        if (grid > 4096) grid = 4096;
        hipLaunchKernelGGL(k_gemm_reduce, dim3((unsigned)grid), dim3(256), 0, st, (const float*)workspace, bias, (__bf16*)c, ldc, p.S, M, N);
    }
    return ver_check_launch("ver_gemm_nn");
}

extern "C" int ver_gemm_nn_segments(const void* lattice, int layout, int B, int H, int W, int C, const int* taps, int ntaps,
                                    const void* cst, int ncst, int cw, const void* w, long ldw, const float* rowpos,
                                    const float* bias, void* c, long ldc, int N, void* stream);
extern "C" int ver_gemm_nn_planes(const void* lattice, int layout, int B, int H, int W, int C, long plane_elems, int nplanes,
                                  const int* tap_plane, const int* taps, int ntaps, const void* cst, int ncst, int cw,
                                  const void* w, long ldw, const float* rowpos, const float* bias, void* c, long ldc, int N,
                                  void* stream);

extern "C" int ver_gemm_nn_taps(const void* lattice, int layout, int B, int H, int W, int C, const int* taps, int ntaps,
                                const void* w, long ldw, const float* rowpos, const float* bias, void* c, long ldc, int N,
                                void* stream) {
    return ver_gemm_nn_segments(lattice, layout, B, H, W, C, taps, ntaps, nullptr, 0, 0, w, ldw, rowpos, bias, c, ldc, N, stream);
}

extern "C" int ver_gemm_nn_segments(const void* lattice, int layout, int B, int H, int W, int C, const int* taps, int ntaps,
                                    const void* cst, int ncst, int cw, const void* w, long ldw, const float* rowpos,
                                    const float* bias, void* c, long ldc, int N, void* stream) {
    return ver_gemm_nn_planes(lattice, layout, B, H, W, C, 0, 0, nullptr, taps, ntaps, cst, ncst, cw, w, ldw, rowpos, bias, c, ldc, N, stream);
}

extern "C" int ver_gemm_nn_planes(const void* lattice, int layout, int B, int H, int W, int C, long plane_elems, int nplanes,
                                  const int* tap_plane, const int* taps, int ntaps, const void* cst, int ncst, int cw,
                                  const void* w, long ldw, const float* rowpos, const float* bias, void* c, long ldc, int N,
                                  void* stream) {
    hipStream_t st = (hipStream_t)stream;
    VER_REQUIRE(B >= 0 && H > 0 && W > 0 && C > 0 && N > 0 && ntaps > 0, VER_EINVAL, "ver_gemm_nn_taps: bad sizes");
    VER_REQUIRE(layout == 0 || layout == 2 || layout == 3, VER_EINVAL, "ver_gemm_nn_taps: layout %d (0 plain, 2 z-split, 3 planar z-split)", layout);
    VER_REQUIRE(layout != 3 || (H % 2 == 0 && W % 2 == 0), VER_EINVAL, "ver_gemm_nn_taps: planar needs even H, W");
    if (B == 0) return VER_OK;
    VER_REQUIRE(lattice && taps && w && c, VER_EINVAL, "ver_gemm_nn_taps: null pointer argument");
    VER_REQUIRE(ntaps <= 64, VER_EUNSUPPORTED, "ver_gemm_nn_taps: %d taps (at most 64)", ntaps);
    VER_REQUIRE(C % 32 == 0 && C >= 64, VER_EUNSUPPORTED, "ver_gemm_nn_taps: C = %d must be a multiple of 32 (>= 64)", C);
    const long M = (long)B * 2 * H * W, lbytes = (long)B * 4 * H * W * C * 2;
    VER_REQUIRE(cst == nullptr || (ncst > 0 && ncst <= 64 && cw >= 32 && cw % 32 == 0 && ((uintptr_t)cst & 15) == 0), VER_EINVAL,
                "ver_gemm_nn_segments: the constant-pattern table needs 1..64 blocks of a width that is a multiple of 32");
    long K = 0;
    for (int i = 0; i < ntaps; ++i) {
        const bool is_cst = taps[3 * i] < 0;                 // (-1 - block, 0, 0): a constant-pattern segment
        VER_REQUIRE(!is_cst || (cst && -1 - taps[3 * i] < ncst), VER_EINVAL, "ver_gemm_nn_segments: segment %d names pattern block %d of %d",
                    i, -1 - taps[3 * i], cst ? ncst : 0);
        K += is_cst ? cw : C;
    }
    VER_REQUIRE(lbytes < 0x7FFFFFFFL, VER_EUNSUPPORTED, "ver_gemm_nn_taps: the source lattice (%ld bytes) exceeds the 2-GiB range of the tap offsets", lbytes);
    VER_REQUIRE(ldw >= N && ldc >= N && ldw % 8 == 0 && ((uintptr_t)w & 15) == 0 && ((uintptr_t)lattice & 15) == 0, VER_EUNSUPPORTED,
                "ver_gemm_nn_taps: w must be 16-byte aligned with a row pitch that is a multiple of 8 elements");
    VER_REQUIRE(K * ldw * 2 < 0xFFFFFFFFL && 256L * ldc * 2 < 0x7FFFFFFFL && K < 0x7FFFFFFFL, VER_EUNSUPPORTED,
                "ver_gemm_nn_taps: a tile's operand range exceeds the 32-bit offsets");
    GemmArgs p;
    p.A = nullptr;
    p.W = (const __bf16*)w;
    p.bias = bias;
    p.C = (__bf16*)c;
    p.lda = 0;
    p.ldw = ldw;
    p.ldc = ldc;
    p.M = M;
    p.K = (int)K;
    p.N = N;
    p.tiles_n = (N + kTile - 1) / kTile;
    p.T = (int)((M + kTile - 1) / kTile) * p.tiles_n;
    p.Kc = (int)K;
    p.S = 1;
    p.ws = nullptr;
    p.per_xcd = (p.T + 7) / 8;
    p.t = TapArgs{};
    p.t.lattice = (const __bf16*)lattice;
    p.t.lattice_bytes = lbytes;
    p.t.B = B, p.t.H = H, p.t.W = W, p.t.C = C, p.t.ntaps = ntaps, p.t.P = 2 * H * W;
    p.t.rowpos = rowpos;
    VER_REQUIRE((plane_elems == 0) == (tap_plane == nullptr) && plane_elems >= 0 && (plane_elems == 0 || (nplanes > 0 && nplanes <= 64 &&
                plane_elems % 8 == 0)), VER_EINVAL, "ver_gemm_nn_planes: plane_elems, nplanes and tap_plane come together");
    p.t.plane_elems = plane_elems;
    for (int i = 0; i < ntaps; ++i) {
        VER_REQUIRE(!tap_plane || (tap_plane[i] >= 0 && tap_plane[i] < nplanes), VER_EINVAL, "ver_gemm_nn_planes: tap %d reads plane %d of %d", i,
                    tap_plane ? tap_plane[i] : 0, nplanes);
        p.t.pl[i] = (signed char)(tap_plane ? tap_plane[i] : 0);
    }
    p.t.cst = (const __bf16*)cst;
    p.t.cst_bytes = cst ? (long)2 * H * W * ncst * cw * 2 : 0;
    p.t.CW = cw, p.t.ncst = ncst;
    for (int i = 0; i < ntaps; ++i) {
        const int dz = taps[3 * i], dy = taps[3 * i + 1], dx = taps[3 * i + 2];
        if (dz < 0) {
            p.t.kind[i] = 1, p.t.dz[i] = (signed char)(-1 - dz), p.t.dy[i] = p.t.dx[i] = 0;
            continue;
        }
        VER_REQUIRE((dz == 0 || dz == 2) && dy >= -64 && dy <= 64 && dx >= -64 && dx <= 64, VER_EINVAL,
                    "ver_gemm_nn_taps: tap %d = (%d, %d, %d): dz must be 0 or 2", i, dz, dy, dx);
        p.t.kind[i] = 0, p.t.dz[i] = (signed char)dz, p.t.dy[i] = (signed char)dy, p.t.dx[i] = (signed char)dx;
    }
    hipError_t e = hipSuccess;
#define VER_GEMM_TAPS(L)                                                                                                  \
    do {                                                                                                                  \
        e = hipFuncSetAttribute((const void*)k_gemm_nn<L>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);       \
        if (e == hipSuccess) hipLaunchKernelGGL(k_gemm_nn<L>, dim3((unsigned)(8 * p.per_xcd)), dim3(512), kLdsBytes, st, p); \
    } while (0)
    if (layout == 0) VER_GEMM_TAPS(0);
    else if (layout == 2) VER_GEMM_TAPS(2);
    else VER_GEMM_TAPS(3);
#undef VER_GEMM_TAPS
    if (e != hipSuccess) return ver_fail(VER_ELAUNCH, "ver_gemm_nn_taps: LDS attribute: %s", hipGetErrorString(e));
    return ver_check_launch("ver_gemm_nn_taps");
}
