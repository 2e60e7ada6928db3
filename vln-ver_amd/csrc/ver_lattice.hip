// Lattice im2col / col2im for the even-lattice upsample (include/ver_ops.h, ver_lattice_*).
//
// The coarse-to-fine head (reference: three ConvTranspose3d, dense_heads/voxelformer_occupancy_head.py
// :251-258,560) is evaluated as im2col + GEMM over a channels-last data lattice [B,Z,H,W,C]
// (vln-ver_amd/dense_heads/upsample.py).  These two kernels are the data movement around the
// GEMMs: pure HBM streaming, 16 bytes per lane, every (row, tap) a contiguous C-vector, bounds
// handled in-kernel (no padded copy of the lattice).
//   im2col : col[(b,z,y,x), t, :] = src[b, z+dz_t, y+dy_t, x+dx_t, :]   (0 outside the lattice)
//   col2im : gsrc[b,z,y,x,:]      = sum_t gcol[(b, z-dz_t, y-dy_t, x-dx_t), t, :]   (gather form:
//            the adjoint without atomics, deterministic)
#include "ver_common.h"

constexpr int kMaxTaps = 80;
struct TapList {
    int n;
    signed char dz[kMaxTaps], dy[kMaxTaps], dx[kMaxTaps];
};

// Generalised form (ver_lattice_gather / ver_lattice_scatter): every tap has its own column offset
// inside a row of `col_stride` elements (the upsample keeps constant-pattern columns between tap
// blocks so that each output parity class reads ONE column range of a shared tap matrix).  The rows
// enumerate (b, zr < Zr, y, x); tap t reads source position (zr + dz_t, y + dy_t, x + dx_t) of a
// lattice with Zs z-layers (zero outside).  Source layouts (`layout`):
//   0 plain      [B, Zs, H, W, C]
//   1 planar     4 x [B, Zs, H/2, W/2, C]: plane 2*pm+pn holds the positions (2y'+pm, 2x'+pn) of the
//                combined (H, W) lattice -- what the previous layer's four class GEMMs leave behind
//   2 z-split    [B, 2, H, W, 2, C] (Zs = 4): element (b,z,y,x) at row (b, z&1, y, x), channel block z>>1
//                -- the output of the Z = 4 layers, whose GEMM yields both z halves side by side (N = 2C)
//   3 planar z-split  4 x [B, 2, H/2, W/2, 2, C]
struct TapListEx {
    int n;
    signed char dz[64], dy[64], dx[64];
    int off[64];        // column offset of the tap block, in 16-byte vectors
};

// index of the C-vector of lattice position (b, z, y, x), in units of C elements
template <int LAYOUT>
__device__ __forceinline__ long lattice_index(int b, int z, int y, int x, int B, int Zs, int H, int W) {
    if (LAYOUT == 1) {
        const int plane = ((y & 1) << 1) | (x & 1);
        return ((((long)plane * B + b) * Zs + z) * (H >> 1) + (y >> 1)) * (W >> 1) + (x >> 1);
    }
    if (LAYOUT == 2) return (((((long)b * 2 + (z & 1)) * H + y) * W + x) << 1) + (z >> 1);
    if (LAYOUT == 3) {
        const int plane = ((y & 1) << 1) | (x & 1);
        return ((((((long)plane * B + b) * 2 + (z & 1)) * (H >> 1) + (y >> 1)) * (W >> 1) + (x >> 1)) << 1) + (z >> 1);
    }
    return (((long)b * Zs + z) * H + y) * W + x;
}

// storage index (in C-vectors) -> (b, z, y, x) of the combined lattice
template <int LAYOUT>
__device__ __forceinline__ void lattice_decode(long r, int B, int Zs, int H, int W, int& b, int& z, int& y, int& x) {
    if (LAYOUT == 0) {
        x = (int)(r % W); r /= W;
        y = (int)(r % H); r /= H;
        z = (int)(r % Zs);
        b = (int)(r / Zs);
    } else if (LAYOUT == 1) {
        const int Wh = W >> 1, Hh = H >> 1;
        const int xh = (int)(r % Wh); r /= Wh;
        const int yh = (int)(r % Hh); r /= Hh;
        z = (int)(r % Zs); r /= Zs;
        b = (int)(r % B);
        const int plane = (int)(r / B);
        y = 2 * yh + (plane >> 1);
        x = 2 * xh + (plane & 1);
    } else if (LAYOUT == 2) {
        const int zh = (int)(r & 1); r >>= 1;
        x = (int)(r % W); r /= W;
        y = (int)(r % H); r /= H;
        z = 2 * zh + (int)(r & 1);
        b = (int)(r >> 1);
    } else {
        const int Wh = W >> 1, Hh = H >> 1;
        const int zh = (int)(r & 1); r >>= 1;
        const int xh = (int)(r % Wh); r /= Wh;
        const int yh = (int)(r % Hh); r /= Hh;
        z = 2 * zh + (int)(r & 1); r >>= 1;
        b = (int)(r % B);
        const int plane = (int)(r / B);
        y = 2 * yh + (plane >> 1);
        x = 2 * xh + (plane & 1);
    }
}

// One workgroup walks rows; inside a row the (tap, vector) index advances without divisions
// (the first version decoded every 16-byte vector from a flat index: five runtime divisions each).
// Constant-pattern blocks of a tap matrix (the columns BETWEEN the tap blocks: 0/1 boundary patterns + a 1 per parity class,
// the same for every viewpoint): `rows` [rows_per_sample][nblk][wv] 16-byte vectors, copied to the column offsets off[].
// Filled by the gather itself (round 4) -- as 16 strided torch copies per step they cost 0.8 ms.
struct ConstBlocks {
    const uint4* rows;
    long rows_per_sample;
    int nblk, wv;
    int off[8];
};

template <int LAYOUT>
__global__ __launch_bounds__(256) void k_lattice_gather(const uint4* __restrict__ src, uint4* __restrict__ col,
                                                        TapListEx taps, long stride_v, int B, int Zr, int Zs, int H,
                                                        int W, int CV, ConstBlocks cb) {
    // per row: the source vector index of every tap (clamped into the lattice) and whether the tap is inside, computed
    // once by the first taps.n threads into LDS (two tables, by row parity: one barrier per row); the copy loop is then
    // a table look-up, a 16-byte load and a nontemporal 16-byte store per vector -- the tap matrix is written once and
    // read by GEMMs much later, it should not displace the lattice (which every row re-reads 18-27 times) from the L2.
    __shared__ long s_src[2][64];
    __shared__ int s_off[2][64];
    const long rows = (long)B * Zr * H * W;
    const int per_row = taps.n * CV;
    const int t0 = (int)threadIdx.x / CV, v0 = (int)threadIdx.x % CV;
    const int dt = 256 / CV, dv = 256 % CV;
    int par = 0;
    for (long row = blockIdx.x; row < rows; row += gridDim.x, par ^= 1) {
        if ((int)threadIdx.x < taps.n) {
            long r = row;
            const int x = (int)(r % W);
            r /= W;
            const int y = (int)(r % H);
            r /= H;
            const int z = (int)(r % Zr);
            const int b = (int)(r / Zr);
            const int t = threadIdx.x;
            const int sz = z + taps.dz[t], sy = y + taps.dy[t], sx = x + taps.dx[t];
            const bool ok = sz >= 0 && sz < Zs && sy >= 0 && sy < H && sx >= 0 && sx < W;
            const int cz = min(max(sz, 0), Zs - 1), cy = min(max(sy, 0), H - 1), cx = min(max(sx, 0), W - 1);
            const long idx = lattice_index<LAYOUT>(b, cz, cy, cx, B, Zs, H, W) * CV;
            s_src[par][t] = ok ? idx : ~idx;          // negative: outside (the clamped index is still loaded, then masked)
            s_off[par][t] = taps.off[t];
        }
        __syncthreads();
        uint4* dst = col + row * stride_v;
        int t = t0, v = v0;
        // unconditional (clamped, then masked) loads, four vectors per thread in flight
#pragma unroll 4
        for (int i = threadIdx.x; i < per_row; i += 256) {
            const long sidx = s_src[par][t];
            const bool ok = sidx >= 0;
            const uint4 val = src[(ok ? sidx : ~sidx) + v];
            const uint4 outv = ok ? val : make_uint4(0u, 0u, 0u, 0u);
            typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
            __builtin_nontemporal_store(u32x4_t{outv.x, outv.y, outv.z, outv.w}, reinterpret_cast<u32x4_t*>(dst + s_off[par][t] + v));
            t += dt;
            v += dv;
            if (v >= CV) {
                v -= CV;
                ++t;
            }
        }
        if (cb.rows) {
            const uint4* cs = cb.rows + (row % cb.rows_per_sample) * (long)(cb.nblk * cb.wv);
            for (int i = threadIdx.x; i < cb.nblk * cb.wv; i += 256) {
                const int bk = i / cb.wv, vv = i - bk * cb.wv;
                const uint4 val = cs[i];
                typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
                __builtin_nontemporal_store(u32x4_t{val.x, val.y, val.z, val.w}, reinterpret_cast<u32x4_t*>(dst + cb.off[bk] + vv));
            }
        }
    }
}

template <bool BF16, int LAYOUT>
__global__ __launch_bounds__(256) void k_lattice_scatter(const uint4* __restrict__ gcol, uint4* __restrict__ gsrc,
                                                         TapListEx taps, long stride_v, int B, int Zr, int Zs, int H,
                                                         int W, int CV) {
    // A workgroup folds G = 256 / CV source-lattice positions at a time (enumerated in STORAGE order): for each of them the
    // tap-matrix vector index of every tap block (or "outside") is computed once into LDS by G * taps.n threads -- two
    // tables by parity, one barrier per group -- and a thread then owns one 16-byte vector of one position: taps.n
    // look-ups, masked loads (six in flight) and fp32 adds.
    __shared__ long s_idx[2][4][64];
    const int G = min(256 / CV, 4);
    const long npos = (long)B * Zs * H * W;
    const int g = (int)threadIdx.x / CV, v = (int)threadIdx.x % CV;
    const bool worker = g < G;
    int par = 0;
    for (long p0 = (long)blockIdx.x * G; p0 < npos; p0 += (long)gridDim.x * G, par ^= 1) {
        if ((int)threadIdx.x < G * taps.n) {
            const int gg = (int)threadIdx.x / taps.n, t = (int)threadIdx.x % taps.n;
            const long pos = min(p0 + gg, npos - 1);
            int b, z, y, x;
            lattice_decode<LAYOUT>(pos, B, Zs, H, W, b, z, y, x);
            const int oz = z - taps.dz[t], oy = y - taps.dy[t], ox = x - taps.dx[t];
            const bool ok = oz >= 0 && oz < Zr && oy >= 0 && oy < H && ox >= 0 && ox < W;
            const int cz = min(max(oz, 0), Zr - 1), cy = min(max(oy, 0), H - 1), cx = min(max(ox, 0), W - 1);
            const long idx = ((((long)b * Zr + cz) * H + cy) * W + cx) * stride_v + taps.off[t];
            s_idx[par][gg][t] = ok ? idx : ~idx;
        }
        __syncthreads();
        const long pos = p0 + g;
        if (worker && pos < npos) {
            float acc[BF16 ? 8 : 4];
#pragma unroll
            for (int j = 0; j < (BF16 ? 8 : 4); ++j) acc[j] = 0.0f;
            for (int tb = 0; tb < taps.n; tb += 6) {
                uint4 gv[6];
#pragma unroll
                for (int u = 0; u < 6; ++u) {
                    const int t = min(tb + u, taps.n - 1);
                    const long sidx = s_idx[par][g][t];
                    const bool ok = tb + u < taps.n && sidx >= 0;
                    const uint4 q = gcol[(sidx >= 0 ? sidx : ~sidx) + v];
                    gv[u] = ok ? q : make_uint4(0u, 0u, 0u, 0u);
                }
#pragma unroll
                for (int u = 0; u < 6; ++u) {
                    const uint4 q = gv[u];
                    if (BF16) {
                        const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            acc[2 * j] += __uint_as_float(w[j] << 16);
                            acc[2 * j + 1] += __uint_as_float(w[j] & 0xffff0000u);
                        }
                    } else {
                        acc[0] += __uint_as_float(q.x);
                        acc[1] += __uint_as_float(q.y);
                        acc[2] += __uint_as_float(q.z);
                        acc[3] += __uint_as_float(q.w);
                    }
                }
            }
            uint4 o;
            if (BF16) {
                auto rne = [](float f) -> uint32_t {
                    uint32_t u = __float_as_uint(f);
                    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
                    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
                };
                o.x = rne(acc[0]) | (rne(acc[1]) << 16);
                o.y = rne(acc[2]) | (rne(acc[3]) << 16);
                o.z = rne(acc[4]) | (rne(acc[5]) << 16);
                o.w = rne(acc[6]) | (rne(acc[7]) << 16);
            } else {
                o = make_uint4(__float_as_uint(acc[0]), __float_as_uint(acc[1]), __float_as_uint(acc[2]),
                               __float_as_uint(acc[3]));
            }
            gsrc[pos * CV + v] = o;
        }
    }
}

__global__ __launch_bounds__(256) void k_lattice_im2col(const uint4* __restrict__ src, uint4* __restrict__ col,
                                                        TapList taps, int B, int Z, int H, int W, int CV) {
    // one thread = one 16-byte vector of one (row, tap); vectors of a (row, tap) are consecutive
    const long total = (long)B * Z * H * W * taps.n * CV;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int v = (int)(i % CV);
        long r = i / CV;
        const int t = (int)(r % taps.n);
        r /= taps.n;
        const int x = (int)(r % W);
        long q = r / W;
        const int y = (int)(q % H);
        q /= H;
        const int z = (int)(q % Z);
        const int b = (int)(q / Z);
        const int sz = z + taps.dz[t], sy = y + taps.dy[t], sx = x + taps.dx[t];
        uint4 val = make_uint4(0u, 0u, 0u, 0u);
        if (sz >= 0 && sz < Z && sy >= 0 && sy < H && sx >= 0 && sx < W)
            val = src[((((long)b * Z + sz) * H + sy) * W + sx) * CV + v];
        col[i] = val;
    }
}

template <bool BF16>
__global__ __launch_bounds__(256) void k_lattice_col2im(const uint4* __restrict__ gcol, uint4* __restrict__ gsrc,
                                                        TapList taps, int B, int Z, int H, int W, int CV) {
    const long total = (long)B * Z * H * W * CV;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int v = (int)(i % CV);
        long r = i / CV;
        const int x = (int)(r % W);
        long q = r / W;
        const int y = (int)(q % H);
        q /= H;
        const int z = (int)(q % Z);
        const int b = (int)(q / Z);
        float acc[BF16 ? 8 : 4];
#pragma unroll
        for (int j = 0; j < (BF16 ? 8 : 4); ++j) acc[j] = 0.0f;
        for (int t = 0; t < taps.n; ++t) {
            const int oz = z - taps.dz[t], oy = y - taps.dy[t], ox = x - taps.dx[t];
            if (oz < 0 || oz >= Z || oy < 0 || oy >= H || ox < 0 || ox >= W) continue;
            const uint4 g = gcol[(((((long)b * Z + oz) * H + oy) * W + ox) * taps.n + t) * CV + v];
            if (BF16) {
                const uint32_t w[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[2 * j] += __uint_as_float(w[j] << 16);
                    acc[2 * j + 1] += __uint_as_float(w[j] & 0xffff0000u);
                }
            } else {
                acc[0] += __uint_as_float(g.x);
                acc[1] += __uint_as_float(g.y);
                acc[2] += __uint_as_float(g.z);
                acc[3] += __uint_as_float(g.w);
            }
        }
        uint4 o;
        if (BF16) {
            auto rne = [](float f) -> uint32_t {   // fp32 -> bf16, round to nearest even
                uint32_t u = __float_as_uint(f);
                if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
                return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
            };
            o.x = rne(acc[0]) | (rne(acc[1]) << 16);
            o.y = rne(acc[2]) | (rne(acc[3]) << 16);
            o.z = rne(acc[4]) | (rne(acc[5]) << 16);
            o.w = rne(acc[6]) | (rne(acc[7]) << 16);
        } else {
            o = make_uint4(__float_as_uint(acc[0]), __float_as_uint(acc[1]), __float_as_uint(acc[2]),
                           __float_as_uint(acc[3]));
        }
        gsrc[i] = o;
    }
}

namespace {
int fill_taps(TapList& tl, const int* taps, int ntaps) {
    VER_REQUIRE(taps && ntaps > 0 && ntaps <= kMaxTaps, VER_EINVAL, "ver_lattice: 1..%d taps required", kMaxTaps);
    tl.n = ntaps;
    for (int t = 0; t < ntaps; ++t) {
        for (int a = 0; a < 3; ++a)
            VER_REQUIRE(taps[3 * t + a] >= -127 && taps[3 * t + a] <= 127, VER_EINVAL, "ver_lattice: tap offset range");
        tl.dz[t] = (signed char)taps[3 * t];
        tl.dy[t] = (signed char)taps[3 * t + 1];
        tl.dx[t] = (signed char)taps[3 * t + 2];
    }
    return VER_OK;
}
int check_lattice(const void* a, const void* b, int B, int Z, int H, int W, int C, int dtype) {
    VER_REQUIRE(B >= 0 && Z > 0 && H > 0 && W > 0 && C > 0, VER_EINVAL, "ver_lattice: bad sizes");
    VER_REQUIRE(dtype == VER_F32 || dtype == VER_BF16, VER_EINVAL, "ver_lattice: dtype %d", dtype);
    VER_REQUIRE((C * (dtype == VER_BF16 ? 2 : 4)) % 16 == 0, VER_EUNSUPPORTED,
                "ver_lattice: channel vector must be a multiple of 16 bytes");
    if (B == 0) return VER_OK;
    VER_REQUIRE(a && b, VER_EINVAL, "ver_lattice: null pointer argument");
    VER_REQUIRE((((uintptr_t)a | (uintptr_t)b) & 15) == 0, VER_EINVAL, "ver_lattice: buffers must be 16-byte aligned");
    return VER_OK;
}
}  // namespace

extern "C" int ver_lattice_im2col(const void* src, void* col, const int* taps, int ntaps, int B, int Z, int H, int W,
                                  int C, int dtype, void* stream) {
    int rc = check_lattice(src, col, B, Z, H, W, C, dtype);
    if (rc) return rc;
    TapList tl;
    rc = fill_taps(tl, taps, ntaps);
    if (rc) return rc;
    if (B == 0) return VER_OK;
    const int CV = C * (dtype == VER_BF16 ? 2 : 4) / 16;
    const long total = (long)B * Z * H * W * ntaps * CV;
    const unsigned blocks = (unsigned)((total + 255) / 256 < 256L * 32 ? (total + 255) / 256 : 256L * 32);
    hipLaunchKernelGGL(k_lattice_im2col, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const uint4*)src,
                       (uint4*)col, tl, B, Z, H, W, CV);
    return ver_check_launch("ver_lattice_im2col");
}

extern "C" int ver_lattice_col2im(const void* gcol, void* gsrc, const int* taps, int ntaps, int B, int Z, int H, int W,
                                  int C, int dtype, void* stream) {
    int rc = check_lattice(gcol, gsrc, B, Z, H, W, C, dtype);
    if (rc) return rc;
    TapList tl;
    rc = fill_taps(tl, taps, ntaps);
    if (rc) return rc;
    if (B == 0) return VER_OK;
    const int CV = C * (dtype == VER_BF16 ? 2 : 4) / 16;
    const long total = (long)B * Z * H * W * CV;
    const unsigned blocks = (unsigned)((total + 255) / 256 < 256L * 32 ? (total + 255) / 256 : 256L * 32);
    if (dtype == VER_BF16)
        hipLaunchKernelGGL(k_lattice_col2im<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const uint4*)gcol,
                           (uint4*)gsrc, tl, B, Z, H, W, CV);
    else
        hipLaunchKernelGGL(k_lattice_col2im<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                           (const uint4*)gcol, (uint4*)gsrc, tl, B, Z, H, W, CV);
    return ver_check_launch("ver_lattice_col2im");
}

namespace {
int fill_taps_ex(TapListEx& tl, const int* taps, const long* col_offset, int ntaps, long col_stride, int C, int esize) {
    VER_REQUIRE(taps && col_offset && ntaps > 0 && ntaps <= 64, VER_EINVAL, "ver_lattice: 1..64 taps with offsets required");
    VER_REQUIRE(col_stride > 0 && (col_stride * esize) % 16 == 0, VER_EINVAL, "ver_lattice: row stride must be a multiple of 16 bytes");
    tl.n = ntaps;
    for (int t = 0; t < ntaps; ++t) {
        for (int a = 0; a < 3; ++a)
            VER_REQUIRE(taps[3 * t + a] >= -127 && taps[3 * t + a] <= 127, VER_EINVAL, "ver_lattice: tap offset range");
        VER_REQUIRE(col_offset[t] >= 0 && col_offset[t] + C <= col_stride && (col_offset[t] * esize) % 16 == 0, VER_EINVAL,
                    "ver_lattice: column offset %ld of tap %d outside the row / unaligned", col_offset[t], t);
        tl.dz[t] = (signed char)taps[3 * t];
        tl.dy[t] = (signed char)taps[3 * t + 1];
        tl.dx[t] = (signed char)taps[3 * t + 2];
        tl.off[t] = (int)(col_offset[t] * esize / 16);
    }
    return VER_OK;
}
int check_layout(const char* who, int layout, int Zr, int Zs, int H, int W) {
    VER_REQUIRE(layout >= 0 && layout <= 3, VER_EINVAL, "%s: layout %d", who, layout);
    VER_REQUIRE(Zr > 0 && Zs > 0, VER_EINVAL, "%s: bad z sizes", who);
    VER_REQUIRE(!(layout & 1) || (H % 2 == 0 && W % 2 == 0), VER_EINVAL, "%s: planar source needs even H, W", who);
    VER_REQUIRE(layout < 2 || Zs == 4, VER_EUNSUPPORTED, "%s: the z-split layouts are built for 4 z-layers", who);
    return VER_OK;
}
unsigned lattice_blocks(long total) {
    return (unsigned)((total + 255) / 256 < 256L * 32 ? (total + 255) / 256 : 256L * 32);
}
}  // namespace

extern "C" int ver_lattice_gather(const void* src, void* col, const int* taps, const long* col_offset, long col_stride,
                                  int ntaps, int B, int Zr, int Zs, int H, int W, int C, int layout, int dtype,
                                  const void* const_rows, const long* const_offset, int const_blocks, int const_width,
                                  void* stream) {
    int rc = check_lattice(src, col, B, Zs, H, W, C, dtype);
    if (rc) return rc;
    rc = check_layout("ver_lattice_gather", layout, Zr, Zs, H, W);
    if (rc) return rc;
    const int esize = dtype == VER_BF16 ? 2 : 4;
    TapListEx tl;
    rc = fill_taps_ex(tl, taps, col_offset, ntaps, col_stride, C, esize);
    if (rc) return rc;
    if (B == 0) return VER_OK;
    const int CV = C * esize / 16;
    const long rows = (long)B * Zr * H * W;
    const long stride_v = col_stride * esize / 16;
    const unsigned blocks = (unsigned)(rows < 256L * 32 ? rows : 256L * 32);
    hipStream_t st = (hipStream_t)stream;
    ConstBlocks cb = {};
    if (const_rows) {
        VER_REQUIRE(const_offset && const_blocks > 0 && const_blocks <= 8 && const_width > 0 && (const_width * esize) % 16 == 0 &&
                        ((uintptr_t)const_rows & 15) == 0,
                    VER_EINVAL, "ver_lattice_gather: constant blocks: 1..8 blocks of whole 16-byte vectors from an aligned table");
        cb.rows = (const uint4*)const_rows;
        cb.rows_per_sample = (long)Zr * H * W;
        cb.nblk = const_blocks;
        cb.wv = const_width * esize / 16;
        for (int i = 0; i < const_blocks; ++i) {
            VER_REQUIRE(const_offset[i] >= 0 && (const_offset[i] * esize) % 16 == 0 && const_offset[i] + const_width <= col_stride,
                        VER_EINVAL, "ver_lattice_gather: constant block %d at column %ld does not fit the 16-byte grid", i,
                        const_offset[i]);
            cb.off[i] = (int)(const_offset[i] * esize / 16);
        }
    }
#define VER_GATHER(L)                                                                                                 \
    hipLaunchKernelGGL(k_lattice_gather<L>, dim3(blocks), dim3(256), 0, st, (const uint4*)src, (uint4*)col, tl, stride_v, \
                       B, Zr, Zs, H, W, CV, cb)
    if (layout == 0) VER_GATHER(0);
    else if (layout == 1) VER_GATHER(1);
    else if (layout == 2) VER_GATHER(2);
    else VER_GATHER(3);
#undef VER_GATHER
    return ver_check_launch("ver_lattice_gather");
}

extern "C" int ver_lattice_scatter(const void* gcol, void* gsrc, const int* taps, const long* col_offset,
                                   long col_stride, int ntaps, int B, int Zr, int Zs, int H, int W, int C, int layout,
                                   int dtype, void* stream) {
    int rc = check_lattice(gcol, gsrc, B, Zs, H, W, C, dtype);
    if (rc) return rc;
    rc = check_layout("ver_lattice_scatter", layout, Zr, Zs, H, W);
    if (rc) return rc;
    const int esize = dtype == VER_BF16 ? 2 : 4;
    TapListEx tl;
    rc = fill_taps_ex(tl, taps, col_offset, ntaps, col_stride, C, esize);
    if (rc) return rc;
    if (B == 0) return VER_OK;
    const int CV = C * esize / 16;
    const long total = (long)B * Zs * H * W * CV;
    const long stride_v = col_stride * esize / 16;
    VER_REQUIRE(CV <= 256, VER_EUNSUPPORTED, "ver_lattice_scatter: more than 256 16-byte vectors per lattice position");
    const long sc_groups = ((long)B * Zs * H * W + (256 / CV < 4 ? 256 / CV : 4) - 1) / (256 / CV < 4 ? 256 / CV : 4);
    const unsigned sc_blocks = (unsigned)(sc_groups < 256L * 32 ? sc_groups : 256L * 32);
    (void)total;
    hipStream_t st = (hipStream_t)stream;
#define VER_SCATTER(BF, L)                                                                                         \
    hipLaunchKernelGGL((k_lattice_scatter<BF, L>), dim3(sc_blocks), dim3(256), 0, st, (const uint4*)gcol, \
                       (uint4*)gsrc, tl, stride_v, B, Zr, Zs, H, W, CV)
    if (dtype == VER_BF16) {
        if (layout == 0) VER_SCATTER(true, 0);
        else if (layout == 1) VER_SCATTER(true, 1);
        else if (layout == 2) VER_SCATTER(true, 2);
        else VER_SCATTER(true, 3);
    } else {
        if (layout == 0) VER_SCATTER(false, 0);
        else if (layout == 1) VER_SCATTER(false, 1);
        else if (layout == 2) VER_SCATTER(false, 2);
        else VER_SCATTER(false, 3);
    }
#undef VER_SCATTER
    return ver_check_launch("ver_lattice_scatter");
}
