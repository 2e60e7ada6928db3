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
// blocks so that each output parity class reads ONE column range of a shared 27-tap matrix), and the
// source may be PLANAR: four planes [B,Z,H/2,W/2,C] holding the positions (2y'+pm, 2x'+pn) of the
// combined (H, W) lattice, plane index 2*pm + pn -- the layout the previous layer's four class
// GEMMs leave behind, so the lattice is never interleaved.
struct TapListEx {
    int n;
    signed char dz[32], dy[32], dx[32];
    int off[32];        // column offset of the tap block, in 16-byte vectors
};

template <bool PLANAR>
__device__ __forceinline__ long lattice_index(int b, int z, int y, int x, int B, int Z, int H, int W) {
    if (PLANAR) {
        const int plane = ((y & 1) << 1) | (x & 1);
        return ((((long)plane * B + b) * Z + z) * (H >> 1) + (y >> 1)) * (W >> 1) + (x >> 1);
    }
    return (((long)b * Z + z) * H + y) * W + x;
}

// One workgroup walks rows; inside a row the (tap, vector) index advances without divisions
// (the first version decoded every 16-byte vector from a flat index: five runtime divisions each).
template <bool PLANAR>
__global__ __launch_bounds__(256) void k_lattice_gather(const uint4* __restrict__ src, uint4* __restrict__ col,
                                                        TapListEx taps, long stride_v, int B, int Z, int H, int W,
                                                        int CV) {
    const long rows = (long)B * Z * H * W;
    const int per_row = taps.n * CV;
    const int t0 = (int)threadIdx.x / CV, v0 = (int)threadIdx.x % CV;
    const int dt = 256 / CV, dv = 256 % CV;
    for (long row = blockIdx.x; row < rows; row += gridDim.x) {
        long r = row;
        const int x = (int)(r % W);
        r /= W;
        const int y = (int)(r % H);
        r /= H;
        const int z = (int)(r % Z);
        const int b = (int)(r / Z);
        uint4* dst = col + row * stride_v;
        int t = t0, v = v0;
        for (int i = threadIdx.x; i < per_row; i += 256) {
            const int sz = z + taps.dz[t], sy = y + taps.dy[t], sx = x + taps.dx[t];
            uint4 val = make_uint4(0u, 0u, 0u, 0u);
            if (sz >= 0 && sz < Z && sy >= 0 && sy < H && sx >= 0 && sx < W)
                val = src[lattice_index<PLANAR>(b, sz, sy, sx, B, Z, H, W) * CV + v];
            dst[taps.off[t] + v] = val;
            t += dt;
            v += dv;
            if (v >= CV) {
                v -= CV;
                ++t;
            }
        }
    }
}

template <bool BF16, bool PLANAR>
__global__ __launch_bounds__(256) void k_lattice_scatter(const uint4* __restrict__ gcol, uint4* __restrict__ gsrc,
                                                         TapListEx taps, long stride_v, int B, int Z, int H, int W,
                                                         int CV) {
    // one thread = one 16-byte vector of the source gradient, enumerated in STORAGE order
    const long total = (long)B * Z * H * W * CV;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int v = (int)(i % CV);
        long r = i / CV;
        int b, z, y, x;
        if (PLANAR) {
            const int Wh = W >> 1, Hh = H >> 1;
            const int xh = (int)(r % Wh);
            long q = r / Wh;
            const int yh = (int)(q % Hh);
            q /= Hh;
            z = (int)(q % Z);
            q /= Z;
            b = (int)(q % B);
            const int plane = (int)(q / B);
            y = 2 * yh + (plane >> 1);
            x = 2 * xh + (plane & 1);
        } else {
            x = (int)(r % W);
            long q = r / W;
            y = (int)(q % H);
            q /= H;
            z = (int)(q % Z);
            b = (int)(q / Z);
        }
        float acc[BF16 ? 8 : 4];
#pragma unroll
        for (int j = 0; j < (BF16 ? 8 : 4); ++j) acc[j] = 0.0f;
        for (int t = 0; t < taps.n; ++t) {
            const int oz = z - taps.dz[t], oy = y - taps.dy[t], ox = x - taps.dx[t];
            if (oz < 0 || oz >= Z || oy < 0 || oy >= H || ox < 0 || ox >= W) continue;
            const uint4 g = gcol[((((long)b * Z + oz) * H + oy) * W + ox) * stride_v + taps.off[t] + v];
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
        gsrc[i] = o;
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
    VER_REQUIRE(taps && col_offset && ntaps > 0 && ntaps <= 32, VER_EINVAL, "ver_lattice: 1..32 taps with offsets required");
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
unsigned lattice_blocks(long total) {
    return (unsigned)((total + 255) / 256 < 256L * 32 ? (total + 255) / 256 : 256L * 32);
}
}  // namespace

extern "C" int ver_lattice_gather(const void* src, void* col, const int* taps, const long* col_offset, long col_stride,
                                  int ntaps, int B, int Z, int H, int W, int C, int planar, int dtype, void* stream) {
    int rc = check_lattice(src, col, B, Z, H, W, C, dtype);
    if (rc) return rc;
    VER_REQUIRE(!planar || (H % 2 == 0 && W % 2 == 0), VER_EINVAL, "ver_lattice_gather: planar source needs even H, W");
    const int esize = dtype == VER_BF16 ? 2 : 4;
    TapListEx tl;
    rc = fill_taps_ex(tl, taps, col_offset, ntaps, col_stride, C, esize);
    if (rc) return rc;
    if (B == 0) return VER_OK;
    const int CV = C * esize / 16;
    const long rows = (long)B * Z * H * W;
    const long stride_v = col_stride * esize / 16;
    const unsigned blocks = (unsigned)(rows < 256L * 32 ? rows : 256L * 32);
    if (planar)
        hipLaunchKernelGGL(k_lattice_gather<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                           (const uint4*)src, (uint4*)col, tl, stride_v, B, Z, H, W, CV);
    else
        hipLaunchKernelGGL(k_lattice_gather<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                           (const uint4*)src, (uint4*)col, tl, stride_v, B, Z, H, W, CV);
    return ver_check_launch("ver_lattice_gather");
}

extern "C" int ver_lattice_scatter(const void* gcol, void* gsrc, const int* taps, const long* col_offset,
                                   long col_stride, int ntaps, int B, int Z, int H, int W, int C, int planar, int dtype,
                                   void* stream) {
    int rc = check_lattice(gcol, gsrc, B, Z, H, W, C, dtype);
    if (rc) return rc;
    VER_REQUIRE(!planar || (H % 2 == 0 && W % 2 == 0), VER_EINVAL, "ver_lattice_scatter: planar source needs even H, W");
    const int esize = dtype == VER_BF16 ? 2 : 4;
    TapListEx tl;
    rc = fill_taps_ex(tl, taps, col_offset, ntaps, col_stride, C, esize);
    if (rc) return rc;
    if (B == 0) return VER_OK;
    const int CV = C * esize / 16;
    const long total = (long)B * Z * H * W * CV;
    const long stride_v = col_stride * esize / 16;
    hipStream_t st = (hipStream_t)stream;
#define VER_SCATTER(BF, PL)                                                                                         \
    hipLaunchKernelGGL((k_lattice_scatter<BF, PL>), dim3(lattice_blocks(total)), dim3(256), 0, st, (const uint4*)gcol, \
                       (uint4*)gsrc, tl, stride_v, B, Z, H, W, CV)
    if (dtype == VER_BF16) {
        if (planar) VER_SCATTER(true, true); else VER_SCATTER(true, false);
    } else {
        if (planar) VER_SCATTER(false, true); else VER_SCATTER(false, false);
    }
#undef VER_SCATTER
    return ver_check_launch("ver_lattice_scatter");
}
