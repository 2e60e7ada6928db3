// Occupancy post-processing on the device (include/ver_ops.h: ver_occ_predict).
//
// Reference: VoxelFormerOccupancyHead.get_occupancy_prediction, focal-loss branch
// (dense_heads/voxelformer_occupancy_head.py:1505-1540):
//     p = sigmoid(logits [N, C]);  p = cat(p, threshold column);  cls = argmax(p, -1);
//     idx = where(cls < C);  pairs = stack(idx, cls[idx])                  -> int64 [K, 2], ascending voxel index
// As torch ops on the GPU that is five passes and an [N, C+1] temporary; here: one classification pass per row
// (sigmoid in fp32, first-occurrence arg-max exactly as torch.argmax orders ties, NaN counts as the maximum), an
// exclusive scan over 1024-row blocks and one ordered compaction pass.  Integer results: bit-exact by construction
// wherever the fp32 sigmoids of a row are distinct.
#include "ver_common.h"

namespace {
constexpr int kRowsPerBlock = 1024;

template <bool BF16>
__device__ __forceinline__ int classify_row(const void* logits, long row, int C, float thr) {
    int best = 0;
    float pb = 0.0f;
    for (int c0 = 0; c0 < C; c0 += 8) {
        float x[8];
        if (BF16) {
            const uint4 t = reinterpret_cast<const uint4*>(logits)[(row * C + c0) / 8];
            const uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                x[2 * j] = __uint_as_float(w[j] << 16);
                x[2 * j + 1] = __uint_as_float(w[j] & 0xffff0000u);
            }
        } else {
            const float4 a = reinterpret_cast<const float4*>(logits)[(row * C + c0) / 4];
            const float4 b = reinterpret_cast<const float4*>(logits)[(row * C + c0) / 4 + 1];
            x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w;
            x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float p = __fdiv_rn(1.0f, __fadd_rn(1.0f, expf(-x[j])));
            const bool take = (c0 + j == 0) || (p > pb) || (isnan(p) && !isnan(pb));
            if (take) {
                best = c0 + j;
                pb = p;
            }
        }
    }
    // the threshold is the LAST column: it wins only when strictly greater than every class probability
    if (!isnan(pb) && thr > pb) best = C;
    return best;
}

template <bool BF16>
__global__ __launch_bounds__(256) void k_occ_count(const void* __restrict__ logits, long N, int C, float thr,
                                                   int* __restrict__ block_count) {
    __shared__ int wsum[4];
    const long base = (long)blockIdx.x * kRowsPerBlock;
    int mine = 0;
#pragma unroll
    for (int j = 0; j < kRowsPerBlock / 256; ++j) {
        const long row = base + j * 256 + threadIdx.x;
        if (row < N) mine += classify_row<BF16>(logits, row, C, thr) < C ? 1 : 0;
    }
    const unsigned long long b = __ballot(mine & 1), b2 = __ballot(mine & 2), b4 = __ballot(mine & 4);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = __popcll(b) + 2 * __popcll(b2) + 4 * __popcll(b4);
    __syncthreads();
    if (threadIdx.x == 0) block_count[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// exclusive scan of the block counts in place (one workgroup); *total = number of occupied rows
__global__ __launch_bounds__(1024) void k_occ_scan(int* __restrict__ block_count, int nblocks, int64_t* __restrict__ total) {
    __shared__ long carry;
    __shared__ int part[1024];
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int b0 = 0; b0 < nblocks; b0 += 1024) {
        const int i = b0 + threadIdx.x;
        const int v = i < nblocks ? block_count[i] : 0;
        part[threadIdx.x] = v;
        __syncthreads();
        for (int d = 1; d < 1024; d <<= 1) {                 // Hillis-Steele inclusive scan
            const int add = threadIdx.x >= d ? part[threadIdx.x - d] : 0;
            __syncthreads();
            part[threadIdx.x] += add;
            __syncthreads();
        }
        const long c = carry;
        if (i < nblocks) block_count[i] = (int)(c + part[threadIdx.x] - v);
        __syncthreads();
        if (threadIdx.x == 1023) carry = c + part[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry;
}

template <bool BF16>
__global__ __launch_bounds__(256) void k_occ_emit(const void* __restrict__ logits, long N, int C, float thr,
                                                  const int* __restrict__ block_off, int64_t* __restrict__ pairs) {
    __shared__ int wsum[4];
    const long base = (long)blockIdx.x * kRowsPerBlock;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
    long off = block_off[blockIdx.x];
    for (int j = 0; j < kRowsPerBlock / 256; ++j) {            // slabs of 256 consecutive rows: ascending output order
        const long row = base + j * 256 + threadIdx.x;
        const int cls = row < N ? classify_row<BF16>(logits, row, C, thr) : C;
        const bool occ = cls < C;
        const unsigned long long bal = __ballot(occ);
        if (lane == 0) wsum[wave] = __popcll(bal);
        __syncthreads();
        int before = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            if (w < wave) before += wsum[w];
            tot += wsum[w];
        }
        if (occ) {
            const long k = off + before + __popcll(bal & below);
            pairs[2 * k] = row;
            pairs[2 * k + 1] = cls;
        }
        off += tot;
        __syncthreads();
    }
}
}  // namespace

extern "C" long ver_occ_predict_blocks(long N) { return N <= 0 ? 0 : (N + kRowsPerBlock - 1) / kRowsPerBlock; }

extern "C" int ver_occ_predict(const void* logits, int dtype, long N, int C, float threshold, int32_t* block_work,
                               int64_t* pairs, int64_t* count, void* stream) {
    VER_REQUIRE(N >= 0 && C > 0, VER_EINVAL, "ver_occ_predict: bad shape N=%ld C=%d", N, C);
    VER_REQUIRE(C % 8 == 0, VER_EUNSUPPORTED, "ver_occ_predict: class count %d is not a multiple of 8", C);
    VER_REQUIRE(dtype == VER_F32 || dtype == VER_BF16, VER_EINVAL, "ver_occ_predict: dtype %d", dtype);
    VER_REQUIRE(count, VER_EINVAL, "ver_occ_predict: null count");
    hipStream_t st = (hipStream_t)stream;
    if (N == 0) {
        hipError_t e = hipMemsetAsync(count, 0, sizeof(int64_t), st);
        return e == hipSuccess ? VER_OK : ver_fail(VER_ELAUNCH, "ver_occ_predict: %s", hipGetErrorString(e));
    }
    VER_REQUIRE(logits && block_work && pairs, VER_EINVAL, "ver_occ_predict: null pointer argument");
    VER_REQUIRE(((uintptr_t)logits & 15) == 0, VER_EINVAL, "ver_occ_predict: logits must be 16-byte aligned");
    const long nb = ver_occ_predict_blocks(N);
    VER_REQUIRE(nb < (1L << 31) && N < (1L << 31), VER_EUNSUPPORTED, "ver_occ_predict: more than 2^31 rows");
    if (dtype == VER_BF16) hipLaunchKernelGGL(k_occ_count<true>, dim3((unsigned)nb), dim3(256), 0, st, logits, N, C, threshold, block_work);
    else hipLaunchKernelGGL(k_occ_count<false>, dim3((unsigned)nb), dim3(256), 0, st, logits, N, C, threshold, block_work);
    int rc = ver_check_launch("ver_occ_predict/count");
    if (rc) return rc;
    hipLaunchKernelGGL(k_occ_scan, dim3(1), dim3(1024), 0, st, block_work, (int)nb, count);
    rc = ver_check_launch("ver_occ_predict/scan");
    if (rc) return rc;
    if (dtype == VER_BF16) hipLaunchKernelGGL(k_occ_emit<true>, dim3((unsigned)nb), dim3(256), 0, st, logits, N, C, threshold, block_work, pairs);
    else hipLaunchKernelGGL(k_occ_emit<false>, dim3((unsigned)nb), dim3(256), 0, st, logits, N, C, threshold, block_work, pairs);
    return ver_check_launch("ver_occ_predict/emit");
}
