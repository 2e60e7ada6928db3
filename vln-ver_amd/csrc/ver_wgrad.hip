// Weight-gradient product of the occupancy head's GEMM layers on gfx950 matrix cores:
//     dW[Ka, N] = A[M, Ka]^T * G[M, N]        (bf16 operands, fp32 accumulation, rows on the contraction axis)
// A is the tap matrix / gathered occ_proj operand of the forward GEMM (possibly a column range of a wider row-major
// matrix), G the gradient of the GEMM's output: dense_heads/upsample.py::rows_tn, i.e. the d(weight) of the reference's
// three ConvTranspose3d and of occ_proj (voxelformer_occupancy_head.py:251-258, :560, :571).
//
// Both operands are row-major with the CONTRACTION index on the slow axis, which is the one layout class the library
// answers with depth-32 macro-tiles (0.35-0.43 of the bf16 peak).  Here:
//   * rows are streamed in slabs of 16 (one k-step of v_mfma_f32_32x32x16_bf16) straight into LDS by LDS-DMA
//     (buffer_load_dwordx4 ... lds): a wave instruction moves 8 rows x 128 B, i.e. whole 128-byte lines of the source;
//   * the MFMA fragments (8 consecutive k of ONE column per lane) come out of that row-major image through
//     ds_read_b64_tr_b16, two per fragment; the 16-byte chunks of a 128-byte line are XOR-ed by bit 1 of the row on the
//     SOURCE address (the LDS-DMA destination is lane-linear), which makes every transposing read conflict free;
//   * a workgroup of 8 waves (2 x 4, wave tile 128 x 64 = 4 x 2 MFMA tiles, 128 accumulator registers) owns a
//     256 x 256 output tile of one row chunk; the two wave groups (wr = 0 / 1, one wave of each per SIMD) run
//     half a phase apart: while one issues its 12 fragment reads + 2 LDS-DMA pieces of the slab PF phases ahead, the
//     other issues its 8 MFMAs (ring of 8 slabs = 128 KB, counted vmcnt, raw s_barrier);
//   * split over row chunks: block b runs on XCD b % 8 and XCD x works on chunks x, x + 8, ...: the rows of a chunk are
//     fetched into ONE L2, where the ~32 tiles of the XCD that are in flight share them; fp32 partial tiles go to a
//     workspace and k_wgrad_reduce adds them up in fp32 (no bf16 rounding of partial sums).
#include <cstdlib>
#include "ver_common.h"

namespace {
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
typedef __attribute__((address_space(3))) void lds_void;

constexpr int kTile = 256;                  // output tile edge
constexpr int kSlabRows = 16;               // rows per slab = k of one MFMA
constexpr int kSlabBytes = 2 * kSlabRows * kTile * 2;   // A part + G part
constexpr int kRing = 8;
constexpr int kLdsBytes = kRing * kSlabBytes;           // 128 KiB

typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

// ds_read_b64_tr_b16 as inline asm: behind the builtin the compiler waits for vmcnt(0) in front of every LDS read
// that follows an LDS-DMA (it cannot tell the ring slots apart), which would serialise the whole pipeline.  The
// results are only valid behind the s_waitcnt lgkmcnt(0) of phase() (the compiler does not count these reads).
template <int OFF>
__device__ __forceinline__ i32x2 tr_read(int addr) {
    i32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
__device__ __forceinline__ bf16x8 frag(i32x2 lo, i32x2 hi) {
    return __builtin_bit_cast(bf16x8, (i32x4)__builtin_shufflevector(lo, hi, 0, 1, 2, 3));
}

struct WgradArgs {
    const __bf16* A;
    const __bf16* G;
    float* ws;          // [S][Ka][N] fp32 partial products
    long lda, ldg, M, Mc;
    int S, Ka, N, tiles_n, T;
    void* out;          // S == 1: the tile goes straight to the result (row pitch ldo, bf16 or fp32), no workspace pass
    long ldo;
    int out_bf16;
    // implicit A (ver_wgrad_tn_segments): the tap matrix of a Z = 4 lattice layer, never materialised (ver_gemm.hip: TapArgs)
    const __bf16* lattice;
    long lattice_bytes;
    const __bf16* cst;  // constant-pattern table [P][ncst][CW] bf16
    long cst_bytes;
    int B, H, W, C, P, CW, ncst, nseg;
    int seg_start[65];  // first column of segment i (seg_start[nseg] = Ka); boundaries are multiples of 64
    signed char kind[64], dz[64], dy[64], dx[64];
};

constexpr int kOutsideW = (int)0x80000000;
template <int L>
__device__ __forceinline__ int wcell_off(const WgradArgs& t, int b, int zl, int y, int x, int dz) {
    if ((unsigned)y >= (unsigned)t.H || (unsigned)x >= (unsigned)t.W) return kOutsideW;
    const int j = dz >> 1;
    int v;
    if (L == 0)
        v = ((b * 4 + zl + dz) * t.H + y) * t.W + x;
    else if (L == 2)
        v = ((((b * 2 + zl) * t.H + y) * t.W + x) << 1) + j;
    else
        v = (((((((y & 1) << 1 | (x & 1)) * t.B + b) * 2 + zl) * (t.H >> 1) + (y >> 1)) * (t.W >> 1) + (x >> 1)) << 1) + j;
    return v * t.C * 2;
}

struct WTapLane {       // implicit A: the lane's running row of the DMA stream, addressed through its wave's offset table
    int taddr;                  // LDS byte address of table[pos] for the row AFTER the ones already looked up
    int boff;                   // b * (bytes per viewpoint) + the lane's byte offset inside the segment's vector
    int bstep;                  // bytes per viewpoint (0 for a pattern segment)
    int pend;                   // bstep when the entry at taddr belongs to the next viewpoint (applied when it is fetched)
    int tend;                   // LDS byte address one past the wave's table (wrap -> next viewpoint)
    int rows_left;              // rows of the operand from the next row to request on (<= 0: past the chunk / M)
    int vo;                     // voffset of the next piece to request (made in the MFMA shadow of the phase before)
};

// The wave's offset table (one per 64-column block of the tile, 4 per workgroup, behind the ring in LDS): entry pos = byte
// offset of the segment's vector for the row at position pos of viewpoint 0 (tap: the neighbouring cell's channel vector, or
// kOutsideW outside the lattice; pattern block: its row of the pattern table).  A row's source is then ONE LDS read + one add
// per slab: the per-slab decode (two quotients, the layout's index arithmetic, the bounds checks -- ~40 VALU instructions in
// front of every 8 MFMAs) cost 20 ms of the 192-viewpoint step.
template <int L>
__device__ __forceinline__ int wtap_table_entry(const WgradArgs& t, int seg, int pos) {
    if (seg < 0) return kOutsideW;                           // (columns past Ka: nothing to read)
    if (t.kind[seg]) return (pos * t.ncst + t.dz[seg]) * t.CW * 2;
    const int hw = t.H * t.W;
    const int zl = pos / hw, rem = pos - zl * hw;
    const int y = rem / t.W, x = rem - y * t.W;
    return wcell_off<L>(t, 0, zl, y + t.dy[seg], x + t.dx[seg], t.dz[seg]);
}

__device__ __forceinline__ int lds_read_b32(int addr) {
    int v;
    asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(addr) : "memory");
    return v;
}

// bookkeeping of one slab of the DMA stream, all VALU, placed in the MFMA shadow of a phase: the table entry fetched in this
// phase (`tfetched`, valid behind the phase's lgkmcnt(0)) becomes the next piece's voffset; the address of the entry after
// it is made ready for the next phase's fetch.
__device__ __forceinline__ void wtap_step(const WgradArgs& t, WTapLane& tl, int tfetched) {
    tl.boff += tl.pend;                                     // (the fetched entry's row may belong to the next viewpoint)
    tl.rows_left -= kSlabRows;
    tl.vo = tl.rows_left > 0 ? (int)((unsigned)tfetched + (unsigned)tl.boff) : kOutsideW;
    tl.taddr += 4 * kSlabRows;
    tl.pend = 0;
    if (tl.taddr >= tl.tend) tl.taddr -= 4 * t.P, tl.pend = tl.bstep;
}

struct WgradLane {          // per-lane constants of the main loop
    int offA0, offA1, offB0, offB1;     // LDS byte addresses of the fragment reads in ring slot 0
    int voA, voG;                       // per-lane source offsets of the two LDS-DMA pieces this wave moves
    int stepA, stepG;                   // bytes per slab of the sources
    int dmaoff;                         // offset of this wave's piece inside the A / G part of a slab
};

// Fragments of ring slot SLOT -> registers (12 transposing reads: A tiles 0..3, G tiles 0..1, lo and hi k halves).
template <int SLOT>
__device__ __forceinline__ void read_slab(const WgradLane& c, i32x2 (&al)[4], i32x2 (&ah)[4], i32x2 (&bl)[2], i32x2 (&bh)[2]) {
    // (the offset field of a DS instruction has 16 bits: slots 4-7 go through base registers 64 KiB up)
    constexpr int SB = (SLOT & 3) * kSlabBytes;
    constexpr int UP = SLOT >= 4 ? 65536 : 0;
    const int a0 = c.offA0 + UP, a1 = c.offA1 + UP, b0 = c.offB0 + UP, b1 = c.offB1 + UP;
    al[0] = tr_read<SB>(a0);
    ah[0] = tr_read<SB + 512>(a0);
    bl[0] = tr_read<SB>(b0);
    bh[0] = tr_read<SB + 512>(b0);
    al[1] = tr_read<SB>(a1);
    ah[1] = tr_read<SB + 512>(a1);
    bl[1] = tr_read<SB>(b1);
    bh[1] = tr_read<SB + 512>(b1);
    al[2] = tr_read<SB + 1024>(a0);
    ah[2] = tr_read<SB + 1536>(a0);
    al[3] = tr_read<SB + 1024>(a1);
    ah[3] = tr_read<SB + 1536>(a1);
}

// One phase = KP slabs (ring slots PS*KP ..): fragments -> registers, LDS-DMA of the slabs PF ahead, 8 KP MFMAs.
template <int PS, int KP, int PF, int IMPL>
__device__ __forceinline__ void phase(char* lds, f32x16 (&acc)[4][2], const WgradLane& c, __amdgpu_buffer_rsrc_t ra,
                                      __amdgpu_buffer_rsrc_t rg, unsigned& soA, unsigned& soG, const WgradArgs& t, WTapLane& tl) {
    i32x2 al[KP][4], ah[KP][4], bl[KP][2], bh[KP][2];
    int tnext = 0;
    read_slab<PS * KP>(c, al[0], ah[0], bl[0], bh[0]);
    if constexpr (KP == 2) read_slab<PS * KP + 1>(c, al[KP - 1], ah[KP - 1], bl[KP - 1], bh[KP - 1]);
#pragma unroll
    for (int u = 0; u < KP; ++u) {
        const int ds = (PS * KP + u + PF) % kRing;
        if constexpr (IMPL < 0) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void*)(lds + ds * kSlabBytes + c.dmaoff), 16, c.voA, (int)soA, 0, 0);
        } else {
            static_assert(IMPL < 0 || KP == 1, "implicit A: one slab per phase");
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void*)(lds + ds * kSlabBytes + c.dmaoff), 16, tl.vo, 0, 0, 0);
            tnext = lds_read_b32(tl.taddr);                // the entry of the row after (consumed in this phase's MFMA shadow)
        }
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rg, (lds_void*)(lds + ds * kSlabBytes + kSlabBytes / 2 + c.dmaoff), 16, c.voG, (int)soG, 0, 0);
        soA += (unsigned)c.stepA;
        soG += (unsigned)c.stepG;
    }
    // this wave's pieces of the NEXT phase's slabs have landed (everything younger stays in flight)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (PF - KP)) : "memory");
    __builtin_amdgcn_s_barrier();
    // the fragment halves pass through the wait as operands: whatever the compiler does to them (copies into the
    // MFMA's register tuples) is ordered behind it
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(al[0][0]), "+v"(ah[0][0]), "+v"(al[0][1]), "+v"(ah[0][1]), "+v"(al[0][2]), "+v"(ah[0][2]), "+v"(al[0][3]),
                   "+v"(ah[0][3]), "+v"(bl[0][0]), "+v"(bh[0][0]), "+v"(bl[0][1]), "+v"(bh[0][1]), "+v"(tnext)
                 :
                 : "memory");

    if constexpr (KP == 2)
        asm volatile(""
                     : "+v"(al[KP - 1][0]), "+v"(ah[KP - 1][0]), "+v"(al[KP - 1][1]), "+v"(ah[KP - 1][1]), "+v"(al[KP - 1][2]),
                       "+v"(ah[KP - 1][2]), "+v"(al[KP - 1][3]), "+v"(ah[KP - 1][3]), "+v"(bl[KP - 1][0]), "+v"(bh[KP - 1][0]),
                       "+v"(bl[KP - 1][1]), "+v"(bh[KP - 1][1])
                     :
                     : "memory");
    bf16x8 a[KP][4], b[KP][2];
#pragma unroll
    for (int u = 0; u < KP; ++u) {
#pragma unroll
        for (int i = 0; i < 4; ++i) a[u][i] = frag(al[u][i], ah[u][i]);
        b[u][0] = frag(bl[u][0], bh[u][0]);
        b[u][1] = frag(bl[u][1], bh[u][1]);
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int u = 0; u < KP; ++u)
#pragma unroll
        for (int it = 0; it < 4; ++it)
#pragma unroll
            for (int jt = 0; jt < 2; ++jt)
                acc[it][jt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[u][it], b[u][jt], acc[it][jt], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (IMPL >= 0) {
        wtap_step(t, tl, tnext);                            // (a dozen VALU instructions under the MFMAs just issued)
        __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_s_barrier();
}

template <int PS, int KP, int PF, int IMPL>
__device__ __forceinline__ void phases_from(int left, char* lds, f32x16 (&acc)[4][2], const WgradLane& c,
                                            __amdgpu_buffer_rsrc_t ra, __amdgpu_buffer_rsrc_t rg, unsigned& soA, unsigned& soG,
                                            const WgradArgs& t, WTapLane& tl) {
    if constexpr (PS < kRing / KP) {
        if (left > PS * KP) {
            phase<PS, KP, PF, IMPL>(lds, acc, c, ra, rg, soA, soG, t, tl);
            phases_from<PS + 1, KP, PF, IMPL>(left, lds, acc, c, ra, rg, soA, soG, t, tl);
        }
    }
}

template <int KP, int PF, int IMPL>
__device__ __forceinline__ void wgrad_body(const WgradArgs& p) {
    static_assert(PF >= KP && PF + (KP == 1 ? 2 : 4) <= kRing, "prefetch distance against the ring (WAR on the slot)");
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    int chunk, tile;
    {
        const int b = blockIdx.x;
        if ((p.S & 7) == 0) {               // block b runs on XCD b % 8: an XCD keeps to its own row chunks
            const int xcd = b & 7, idx = b >> 3;
            chunk = xcd + 8 * (idx / p.T);
            tile = idx % p.T;
        } else {
            chunk = b / p.T;
            tile = b % p.T;
        }
    }
    const int mt = tile / p.tiles_n, nt = tile - mt * p.tiles_n;
    const long row0 = (long)chunk * p.Mc;
    const long rows = max(0L, min(p.Mc, p.M - row0));
    // whole phases; rows past M read as zeros (buffer range), and Mc is a multiple of 16 KP
    const int nslab = (int)((rows + kSlabRows * KP - 1) / (kSlabRows * KP)) * KP;

    // LDS-DMA: wave w moves piece (row group w >> 2, 64-column block w & 3) of the A part and of the G part
    const __bf16* ab = IMPL < 0 ? p.A + row0 * p.lda + (long)mt * kTile : nullptr;
    const __bf16* gb = p.G + row0 * p.ldg + (long)nt * kTile;
    // buffer ranges end with the last valid element of the operand (column Ka / N of row M - 1): what lies behind reads
    // as zero, what lies beside a row (other columns of a wider matrix) only reaches outputs that are never stored
    const long abytes = ((p.M - row0 - 1) * p.lda + p.Ka - (long)mt * kTile) * 2;
    const long gbytes = ((p.M - row0 - 1) * p.ldg + p.N - (long)nt * kTile) * 2;
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void*)gb, 0, (int)max(0L, min(gbytes, 0xFFFFFFFFL)), 0x00020000);
    WgradLane c;
    const int prow = lane >> 3, pch = (lane & 7) ^ (((prow >> 1) & 1) << 2);
    // implicit A: the segment of this wave's 64 columns (boundaries are multiples of 64) -> its source (lattice or pattern
    // table) and tap; the lane's first row
    WTapLane tl = {};
    bool a_cst = false;
    if constexpr (IMPL >= 0) {
        // the four offset tables of the tile's 64-column blocks, behind the ring (4 x P ints; the launcher has the LDS for them)
        int* tables = reinterpret_cast<int*>(lds + kLdsBytes);
        auto seg_of = [&](int col0) {
            int seg = -1;
            for (int i = 0; i < p.nseg; ++i)
                if (col0 >= p.seg_start[i] && col0 < p.seg_start[i + 1]) seg = i;
            return seg;
        };
        for (int q = 0; q < 4; ++q) {
            const int seg = seg_of(mt * kTile + 64 * q);
            for (int i = tid; i < p.P; i += 512) tables[q * p.P + i] = wtap_table_entry<IMPL>(p, seg, i);
        }
        __syncthreads();
        const int col0 = mt * kTile + 64 * (wave & 3);
        const int seg = seg_of(col0);
        a_cst = seg >= 0 && p.kind[seg];
        const long r = row0 + 8 * (wave >> 2) + prow;
        const int b0 = (int)(r / p.P), pos0 = (int)(r - (long)b0 * p.P);
        // bytes per viewpoint of the source (pattern table: none): layouts 0 / 2 hold a viewpoint's 4 H W cells together,
        // the planar one H W cells per plane
        tl.bstep = a_cst ? 0 : (IMPL == 3 ? p.H * p.W : 4 * p.H * p.W) * p.C * 2;
        tl.boff = b0 * tl.bstep + (seg < 0 ? 0 : (col0 - p.seg_start[seg]) * 2) + pch * 16;
        const int lbase = (int)(uintptr_t)(__attribute__((address_space(3))) char*)(lds + kLdsBytes) + (wave & 3) * p.P * 4;
        tl.tend = lbase + 4 * p.P;
        tl.taddr = lbase + 4 * pos0;
        tl.pend = 0;
        tl.rows_left = (int)min(min(p.M, row0 + p.Mc) - r, 0x7FFFFFF0L) + kSlabRows;   // (a chunk ends where the next one starts)
        int t0 = lds_read_b32(tl.taddr);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(t0)::"memory");
        wtap_step(p, tl, t0);                                   // -> tl.vo of the lane's first row, taddr at its second
    }
    const __amdgpu_buffer_rsrc_t ra =
        IMPL < 0 ? __builtin_amdgcn_make_buffer_rsrc((void*)ab, 0, (int)max(0L, min(abytes, 0xFFFFFFFFL)), 0x00020000)
                 : __builtin_amdgcn_make_buffer_rsrc((void*)(a_cst ? p.cst : p.lattice), 0,
                                                     (int)max(0L, min(a_cst ? p.cst_bytes : p.lattice_bytes, 0x7FFFFFFFL)), 0x00020000);
    c.voA = (int)(((8 * (wave >> 2) + prow) * p.lda + 64 * (wave & 3)) * 2) + pch * 16;
    c.voG = (int)(((8 * (wave >> 2) + prow) * p.ldg + 64 * (wave & 3)) * 2) + pch * 16;
    c.stepA = (int)(kSlabRows * p.lda * 2);
    c.stepG = (int)(kSlabRows * p.ldg * 2);
    c.dmaoff = wave * 1024;
    unsigned soA = 0, soG = 0;          // (byte offsets of the running slab: up to 4 GiB, unsigned arithmetic)

    // transposing fragment reads: 16-lane group q = (k half, column half), lane c of it addresses row c >> 2,
    // columns 4 (c & 3) .. + 3 and receives column c of the group's 4 x 16 block
    {
        const int q = lane >> 4, cl = lane & 15;
        const int lowch = (2 * (q & 1) + ((cl & 3) >> 1)) ^ (((cl >> 3) & 1) << 2);
        const int lbase = (int)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
        const int rowoff = lbase + ((q >> 1) * 32 + (cl >> 2)) * 128 + lowch * 16 + (cl & 1) * 8;
        c.offA0 = rowoff + 2 * wr * 1024;
        c.offA1 = c.offA0 ^ 64;
        c.offB0 = rowoff + wc * 1024 + kSlabBytes / 2;
        c.offB1 = c.offB0 ^ 64;
    }

    f32x16 acc[4][2];
#pragma unroll
    for (int it = 0; it < 4; ++it)
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[it][jt][r] = 0.0f;

    // prologue: slabs 0 .. PF-1 in flight, the first phase's slabs landed
#pragma unroll
    for (int s = 0; s < PF; ++s) {
        if constexpr (IMPL < 0) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void*)(lds + s * kSlabBytes + c.dmaoff), 16, c.voA, (int)soA, 0, 0);
        } else {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void*)(lds + s * kSlabBytes + c.dmaoff), 16, tl.vo, 0, 0, 0);
            int tn = lds_read_b32(tl.taddr);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(tn)::"memory");
            wtap_step(p, tl, tn);
        }
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rg, (lds_void*)(lds + s * kSlabBytes + kSlabBytes / 2 + c.dmaoff), 16, c.voG, (int)soG, 0, 0);
        soA += (unsigned)c.stepA;
        soG += (unsigned)c.stepG;
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (PF - KP)) : "memory");
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();          // the second wave group runs half a phase behind

    int s = 0;
    for (; s + kRing <= nslab; s += kRing) phases_from<0, KP, PF, IMPL>(kRing, lds, acc, c, ra, rg, soA, soG, p, tl);
    phases_from<0, KP, PF, IMPL>(nslab - s, lds, acc, c, ra, rg, soA, soG, p, tl);     // < 8 slabs left: wave-uniform exits
    if (wr == 0) __builtin_amdgcn_s_barrier();          // barrier counts of the two groups match again
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the run-ahead pieces nobody reads

    // partial tile -> workspace: register r of tile (it, jt) is row (r & 3) + 8 (r >> 2) + 4 (lane >> 5), column lane & 31
    const int i0 = mt * kTile + 128 * wr + 4 * (lane >> 5), j0 = nt * kTile + 64 * wc + (lane & 31);
    if (p.S == 1 && p.out) {
        // one row chunk: there is nothing to add up -- the same fp32 sums, rounded once, straight into the result (a
        // one-viewpoint step has 450 / 1 800 rows: the workspace round trip was 10 of its 12 bytes per element)
#pragma unroll
        for (int it = 0; it < 4; ++it)
#pragma unroll
            for (int jt = 0; jt < 2; ++jt) {
                const int j = j0 + 32 * jt;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int i = i0 + 32 * it + (r & 3) + 8 * (r >> 2);
                    if (i < p.Ka && j < p.N) {
                        if (p.out_bf16)
                            reinterpret_cast<__bf16*>(p.out)[(long)i * p.ldo + j] = (__bf16)acc[it][jt][r];
                        else
                            reinterpret_cast<float*>(p.out)[(long)i * p.ldo + j] = acc[it][jt][r];
                    }
                }
            }
        return;
    }
    float* out = p.ws + (long)chunk * p.Ka * p.N;
#pragma unroll
    for (int it = 0; it < 4; ++it)
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) {
            const int j = j0 + 32 * jt;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = i0 + 32 * it + (r & 3) + 8 * (r >> 2);
                if (i < p.Ka && j < p.N) out[(long)i * p.N + j] = acc[it][jt][r];
            }
        }
}

template <int KP, int PF>
__global__ __launch_bounds__(512) void k_wgrad_tn(WgradArgs p) {
    wgrad_body<KP, PF, -1>(p);
}

// the same kernel with the implicit tap matrix of a lattice in layout L as A (ver_wgrad_tn_segments)
template <int L, int PF>
__global__ __launch_bounds__(512) void k_wgrad_tn_seg(WgradArgs p) {
    wgrad_body<1, PF, L>(p);
}

// out[i][j] (bf16 or fp32, row pitch ldo) = sum over the S partial products, fp32; 4 elements per thread
template <typename OT>
__global__ __launch_bounds__(256) void k_wgrad_reduce(const float* __restrict__ ws, OT* __restrict__ out, long ldo, int S,
                                                      int Ka, int N) {
    const long n4 = (long)Ka * N / 4;
    const long stride = (long)Ka * N;
    for (long e = blockIdx.x * 256L + threadIdx.x; e < n4; e += (long)gridDim.x * 256) {
        float4 acc = *reinterpret_cast<const float4*>(ws + 4 * e);
        for (int s = 1; s < S; ++s) {
            const float4 v = *reinterpret_cast<const float4*>(ws + s * stride + 4 * e);
            acc.x += v.x;
            acc.y += v.y;
            acc.z += v.z;
            acc.w += v.w;
        }
        const long i = (4 * e) / N, j = (4 * e) - i * N;
        OT* o = out + i * ldo + j;
        if constexpr (sizeof(OT) == 4) {
            *reinterpret_cast<float4*>(o) = acc;
        } else {
            typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
            bf16x4 v;
            v.x = (__bf16)acc.x;
            v.y = (__bf16)acc.y;
            v.z = (__bf16)acc.z;
            v.w = (__bf16)acc.w;
            *reinterpret_cast<bf16x4*>(o) = v;
        }
    }
}

// rows of one chunk that keep every slab offset inside the 4-GiB range of a buffer offset, for row pitch `ld` elements
// (the launcher's check: (Mc + 16 * kRing) * ld * 2 < 2^32, Mc rounded up to 2 * kSlabRows rows)
long max_chunk_rows(long ld) {
    if (ld <= 0) return 45000;
    const long rows = (0xFFFFFFFFL / (ld * 2)) - 16 * kRing - 2 * kSlabRows - 1;
    return rows < 45000 ? (rows > 2 * kSlabRows ? rows : 2 * kSlabRows) : 45000;
}

int pick_splits(long M, int Ka, int N, long ld = 0) {
    const long cap = max_chunk_rows(ld);
    const long tiles = (long)((Ka + kTile - 1) / kTile) * ((N + kTile - 1) / kTile);
    // Cost model (microseconds, from the measurements in profiles/r05_wgrad_microbench.txt): rounds of 256 workgroups, each
    // one chunk of rows at 0.37 us per 16-row slab + 8 us of fill / drain / epilogue, plus 8 B of workspace traffic per
    // output element and chunk at ~3 TB/s.  Multiples of 8 chunks keep every XCD on its own rows (ties go to them; above
    // 32 768 rows nothing else is considered);
    // a chunk has to stay inside the 4-GiB range of a buffer offset: at most `cap` rows (45 000 at the step's widest operands,
    // fewer for a wider row pitch `ld`).
    static const int cand[] = {1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64};
    int best = 0;
    double best_us = 0.0;
    for (int s : cand) {
        if ((M + s - 1) / s > cap) continue;
        if (s > 1 && M / s < 256) break;
        if (M > 32768 && s % 8) continue;       // (long row ranges: only the XCD-local form has been measured)
        const long rounds = (tiles * s + 255) / 256;
        const double slabs = (double)((M + s - 1) / s + 15) / 16.0;
        // (one chunk: the tile goes straight to the result, ~2-4 B per element instead of 8 per chunk + the reduce)
        double us = (double)rounds * (slabs * 0.37 + 8.0) + (s == 1 ? 0.25 : (double)s) * (double)Ka * (double)N * 8.0 / 3.0e6;
        if (s % 8 == 0) us *= 0.97;
        if (!best || us < best_us) {
            best = s;
            best_us = us;
        }
    }
    if (!best) {
        best = 64;
        while (best < 65536 && (M + best - 1) / best > cap) best *= 2;
    }
    return best;
}

template <int KP, int PF>
void launch_tn(const WgradArgs& p, int blocks, hipStream_t st, hipError_t& e) {
    auto kern = k_wgrad_tn<KP, PF>;
    e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
    if (e == hipSuccess) hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(512), kLdsBytes, st, p);
}
template <int L, int PF>
void launch_tn_segments_pf(const WgradArgs& p, int blocks, hipStream_t st, hipError_t& e) {
    const int ldsb = kLdsBytes + 4 * p.P * 4;            // the ring + four offset tables of P ints
    e = hipFuncSetAttribute((const void*)k_wgrad_tn_seg<L, PF>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);
    if (e == hipSuccess) hipLaunchKernelGGL((k_wgrad_tn_seg<L, PF>), dim3((unsigned)blocks), dim3(512), ldsb, st, p);
}

template <int L>
void launch_tn_segments(const WgradArgs& p, int blocks, hipStream_t st, hipError_t& e) {
    static const int pf = [] {                             // (prefetch distance in slabs; VER_WGRAD_SEG_PF: experiments)
        const char* v = getenv("VER_WGRAD_SEG_PF");
        return v ? atoi(v) : 3;
    }();
    if (pf == 4) launch_tn_segments_pf<L, 4>(p, blocks, st, e);
    else if (pf == 5) launch_tn_segments_pf<L, 5>(p, blocks, st, e);
    else if (pf == 6) launch_tn_segments_pf<L, 6>(p, blocks, st, e);
    else launch_tn_segments_pf<L, 3>(p, blocks, st, e);
}
}  // namespace

extern "C" int ver_wgrad_tn_splits(long M, int Ka, int N) {
    if (M <= 0 || Ka <= 0 || N <= 0) return 1;
    return pick_splits(M, Ka, N);
}

extern "C" int ver_wgrad_tn_splits_ld(long M, int Ka, int N, long ld) {
    if (M <= 0 || Ka <= 0 || N <= 0) return 1;
    return pick_splits(M, Ka, N, ld);
}

extern "C" long ver_wgrad_tn_workspace(long M, int Ka, int N, int splits) {
    if (Ka <= 0 || N <= 0) return 0;
    if (splits <= 0) splits = ver_wgrad_tn_splits(M, Ka, N);
    return (long)splits * Ka * N * (long)sizeof(float);
}

extern "C" int ver_wgrad_tn(const void* a, long lda, const void* g, long ldg, long M, int Ka, int N, void* out, long ldo,
                            int out_dtype, int splits, int flags, void* workspace, long workspace_bytes, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    VER_REQUIRE(M >= 0 && Ka > 0 && N > 0, VER_EINVAL, "ver_wgrad_tn: bad sizes M=%ld Ka=%d N=%d", M, Ka, N);
    VER_REQUIRE(out && workspace && ((a && g) || M == 0), VER_EINVAL, "ver_wgrad_tn: null pointer argument");
    VER_REQUIRE(out_dtype == VER_F32 || out_dtype == VER_BF16, VER_EINVAL, "ver_wgrad_tn: out_dtype %d", out_dtype);
    VER_REQUIRE(lda >= Ka && ldg >= N && ldo >= N, VER_EINVAL, "ver_wgrad_tn: row pitch smaller than the row");
    VER_REQUIRE(lda % 8 == 0 && ldg % 8 == 0 && ((uintptr_t)a & 15) == 0 && ((uintptr_t)g & 15) == 0, VER_EUNSUPPORTED,
                "ver_wgrad_tn: operands must be 16-byte aligned with row pitches that are multiples of 8 elements");
    VER_REQUIRE(N % 4 == 0 && ldo % 4 == 0 && ((uintptr_t)out & 15) == 0, VER_EUNSUPPORTED,
                "ver_wgrad_tn: N and the output pitch must be multiples of 4");
    const int S = splits > 0 ? splits : pick_splits(M, Ka, N, lda > ldg ? lda : ldg);
    VER_REQUIRE(S <= 65536, VER_EINVAL, "ver_wgrad_tn: %d row chunks", S);
    const long Mc = ((M + S - 1) / S + 2 * kSlabRows - 1) / (2 * kSlabRows) * (2 * kSlabRows);  // rows per chunk, whole phases
    VER_REQUIRE((Mc + 16 * kRing) * (lda > ldg ? lda : ldg) * 2 < 0xFFFFFFFFL, VER_EUNSUPPORTED,
                "ver_wgrad_tn: a row chunk exceeds the 4-GiB range of a buffer offset (more splits)");
    VER_REQUIRE(workspace_bytes >= (long)S * Ka * N * (long)sizeof(float), VER_EINVAL, "ver_wgrad_tn: workspace of %ld bytes, %ld needed",
                workspace_bytes, (long)S * Ka * N * (long)sizeof(float));
    WgradArgs p = {};
    p.A = (const __bf16*)a;
    p.G = (const __bf16*)g;
    p.ws = (float*)workspace;
    p.lda = lda;
    p.ldg = ldg;
    p.M = M;
    p.Mc = Mc;
    p.S = S;
    p.Ka = Ka;
    p.N = N;
    p.tiles_n = (N + kTile - 1) / kTile;
    p.T = ((Ka + kTile - 1) / kTile) * p.tiles_n;
    const bool direct = S == 1 && M > 0;
    p.out = direct ? out : nullptr;
    p.ldo = ldo;
    p.out_bf16 = out_dtype == VER_BF16;
    hipError_t e = hipSuccess;
    if (M > 0) {
        switch (flags & 15) {
            case 4: launch_tn<1, 4>(p, p.T * S, st, e); break;
            case 5: launch_tn<1, 5>(p, p.T * S, st, e); break;
            case 6: launch_tn<1, 6>(p, p.T * S, st, e); break;
            case 8 + 2: launch_tn<2, 2>(p, p.T * S, st, e); break;
            case 8 + 4: launch_tn<2, 4>(p, p.T * S, st, e); break;
            default: launch_tn<1, 3>(p, p.T * S, st, e); break;
        }
        if (e != hipSuccess) return ver_fail(VER_ELAUNCH, "ver_wgrad_tn: LDS attribute: %s", hipGetErrorString(e));
    } else {
        e = hipMemsetAsync(workspace, 0, (size_t)S * Ka * N * sizeof(float), st);
        if (e != hipSuccess) return ver_fail(VER_ELAUNCH, "ver_wgrad_tn: memset: %s", hipGetErrorString(e));
    }
    if (direct) return ver_check_launch("ver_wgrad_tn");
    const long n4 = (long)Ka * N / 4;
    long grid = (n4 + 255) / 256;
    if (grid > 8192) grid = 8192;
    if (out_dtype == VER_F32)
        hipLaunchKernelGGL(k_wgrad_reduce<float>, dim3((unsigned)grid), dim3(256), 0, st, (const float*)workspace, (float*)out, ldo, S, Ka, N);
    else
        hipLaunchKernelGGL(k_wgrad_reduce<__bf16>, dim3((unsigned)grid), dim3(256), 0, st, (const float*)workspace, (__bf16*)out, ldo, S, Ka, N);
    return ver_check_launch("ver_wgrad_tn");
}

extern "C" int ver_wgrad_tn_segments(const void* lattice, int layout, int B, int H, int W, int C, const int* taps, int nseg,
                                     const void* cst, int ncst, int cw, const void* g, long ldg, int N, void* out, long ldo,
                                     int out_dtype, int splits, void* workspace, long workspace_bytes, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    VER_REQUIRE(B >= 0 && H > 0 && W > 0 && C > 0 && N > 0 && nseg > 0, VER_EINVAL, "ver_wgrad_tn_segments: bad sizes");
    VER_REQUIRE(layout == 0 || layout == 2 || layout == 3, VER_EINVAL, "ver_wgrad_tn_segments: layout %d (0 plain, 2 z-split, 3 planar z-split)", layout);
    VER_REQUIRE(layout != 3 || (H % 2 == 0 && W % 2 == 0), VER_EINVAL, "ver_wgrad_tn_segments: planar needs even H, W");
    VER_REQUIRE(out && workspace && ((lattice && g && taps) || B == 0), VER_EINVAL, "ver_wgrad_tn_segments: null pointer argument");
    VER_REQUIRE(out_dtype == VER_F32 || out_dtype == VER_BF16, VER_EINVAL, "ver_wgrad_tn_segments: out_dtype %d", out_dtype);
    VER_REQUIRE(nseg <= 64, VER_EUNSUPPORTED, "ver_wgrad_tn_segments: %d segments (at most 64)", nseg);
    VER_REQUIRE(C % 64 == 0 && (cst == nullptr || (ncst > 0 && ncst <= 64 && cw > 0 && cw % 64 == 0 && ((uintptr_t)cst & 15) == 0)),
                VER_EUNSUPPORTED, "ver_wgrad_tn_segments: segment widths (C = %d, pattern blocks of %d) must be multiples of 64", C, cw);
    const long M = (long)B * 2 * H * W, lbytes = (long)B * 4 * H * W * C * 2;
    const int P = 2 * H * W;
    VER_REQUIRE(P >= kSlabRows && P <= 2048 && lbytes < 0x7FFFFFFFL, VER_EUNSUPPORTED,
                "ver_wgrad_tn_segments: %d rows per viewpoint / a source lattice of %ld bytes (16 <= rows <= 2048: four offset tables "
                "next to the 128-KB ring in LDS; below 2 GiB)", P, lbytes);
    VER_REQUIRE(ldg >= N && ldo >= N && ldg % 8 == 0 && ((uintptr_t)g & 15) == 0 && ((uintptr_t)lattice & 15) == 0, VER_EUNSUPPORTED,
                "ver_wgrad_tn_segments: g must be 16-byte aligned with a row pitch that is a multiple of 8 elements");
    VER_REQUIRE(N % 4 == 0 && ldo % 4 == 0 && ((uintptr_t)out & 15) == 0, VER_EUNSUPPORTED,
                "ver_wgrad_tn_segments: N and the output pitch must be multiples of 4");
    WgradArgs p = {};
    long Ka = 0;
    for (int i = 0; i < nseg; ++i) {
        const int dz = taps[3 * i], dy = taps[3 * i + 1], dx = taps[3 * i + 2];
        p.seg_start[i] = (int)Ka;
        if (dz < 0) {
            VER_REQUIRE(cst && -1 - dz < ncst, VER_EINVAL, "ver_wgrad_tn_segments: segment %d names pattern block %d of %d", i, -1 - dz, cst ? ncst : 0);
            p.kind[i] = 1, p.dz[i] = (signed char)(-1 - dz);
            Ka += cw;
        } else {
            VER_REQUIRE((dz == 0 || dz == 2) && dy >= -64 && dy <= 64 && dx >= -64 && dx <= 64, VER_EINVAL,
                        "ver_wgrad_tn_segments: tap %d = (%d, %d, %d): dz must be 0 or 2", i, dz, dy, dx);
            p.kind[i] = 0, p.dz[i] = (signed char)dz, p.dy[i] = (signed char)dy, p.dx[i] = (signed char)dx;
            Ka += C;
        }
    }
    p.seg_start[nseg] = (int)Ka;
    VER_REQUIRE(Ka < 0x7FFFFFFFL, VER_EUNSUPPORTED, "ver_wgrad_tn_segments: %ld columns", Ka);
    const int S = splits > 0 ? splits : pick_splits(M, (int)Ka, N, ldg);
    VER_REQUIRE(S <= 65536, VER_EINVAL, "ver_wgrad_tn_segments: %d row chunks", S);
    const long Mc = ((M + S - 1) / S + 2 * kSlabRows - 1) / (2 * kSlabRows) * (2 * kSlabRows);
    VER_REQUIRE((Mc + 16 * kRing) * ldg * 2 < 0xFFFFFFFFL, VER_EUNSUPPORTED,
                "ver_wgrad_tn_segments: a row chunk exceeds the 4-GiB range of a buffer offset (more splits)");
    VER_REQUIRE(workspace_bytes >= (long)S * Ka * N * (long)sizeof(float), VER_EINVAL, "ver_wgrad_tn_segments: workspace of %ld bytes, %ld needed",
                workspace_bytes, (long)S * Ka * N * (long)sizeof(float));
    p.A = nullptr;
    p.G = (const __bf16*)g;
    p.ws = (float*)workspace;
    p.lda = 0;
    p.ldg = ldg;
    p.M = M;
    p.Mc = Mc;
    p.S = S;
    p.Ka = (int)Ka;
    p.N = N;
    p.tiles_n = (N + kTile - 1) / kTile;
    p.T = (int)((Ka + kTile - 1) / kTile) * p.tiles_n;
    const bool direct = S == 1 && M > 0;
    p.out = direct ? out : nullptr;
    p.ldo = ldo;
    p.out_bf16 = out_dtype == VER_BF16;
    p.lattice = (const __bf16*)lattice;
    p.lattice_bytes = lbytes;
    p.cst = (const __bf16*)cst;
    p.cst_bytes = cst ? (long)P * ncst * cw * 2 : 0;
    p.B = B, p.H = H, p.W = W, p.C = C, p.P = P, p.CW = cw, p.ncst = ncst, p.nseg = nseg;
    hipError_t e = hipSuccess;
    if (M > 0) {
        if (layout == 0) launch_tn_segments<0>(p, p.T * S, st, e);
        else if (layout == 2) launch_tn_segments<2>(p, p.T * S, st, e);
        else launch_tn_segments<3>(p, p.T * S, st, e);
        if (e != hipSuccess) return ver_fail(VER_ELAUNCH, "ver_wgrad_tn_segments: LDS attribute: %s", hipGetErrorString(e));
    } else {
        e = hipMemsetAsync(workspace, 0, (size_t)S * Ka * N * sizeof(float), st);
        if (e != hipSuccess) return ver_fail(VER_ELAUNCH, "ver_wgrad_tn_segments: memset: %s", hipGetErrorString(e));
    }
    if (direct) return ver_check_launch("ver_wgrad_tn_segments");
    const long n4 = Ka * N / 4;
    long grid = (n4 + 255) / 256;
    if (grid > 8192) grid = 8192;
    if (out_dtype == VER_F32)
        hipLaunchKernelGGL(k_wgrad_reduce<float>, dim3((unsigned)grid), dim3(256), 0, st, (const float*)workspace, (float*)out, ldo, S, (int)Ka, N);
    else
        hipLaunchKernelGGL(k_wgrad_reduce<__bf16>, dim3((unsigned)grid), dim3(256), 0, st, (const float*)workspace, (__bf16*)out, ldo, S, (int)Ka, N);
    return ver_check_launch("ver_wgrad_tn_segments");
}

extern "C" int ver_wgrad_tn_segments_splits(int B, int H, int W, long Ka, int N, long ldg) {
    const long M = (long)B * 2 * H * W;
    if (M <= 0 || Ka <= 0 || N <= 0) return 1;
    return pick_splits(M, (int)Ka, N, ldg);
}
