// Error plumbing of the C ABI (include/ver_ops.h).
#include <cstdarg>
#include <cstdio>
#include "ver_common.h"

static thread_local char g_err[512] = "";

int ver_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int ver_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return ver_fail(VER_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
    return VER_OK;
}

__global__ __launch_bounds__(256) void k_zero_words(uint32_t* __restrict__ p, size_t n, int vec) {
    const size_t n4 = vec ? n >> 2 : 0;                      // 16-byte stores when the buffer is aligned for them
    for (size_t i = blockIdx.x * 256UL + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) reinterpret_cast<uint4*>(p)[i] = make_uint4(0, 0, 0, 0);
    for (size_t i = (n4 << 2) + blockIdx.x * 256UL + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = 0;
}

int ver_zero_async(void* ptr, size_t bytes, void* stream) {
    if (bytes == 0) return VER_OK;
    if (!ptr || (bytes & 3) || ((uintptr_t)ptr & 3)) return ver_fail(VER_EINVAL, "ver_zero_async: %zu bytes at %p (4-byte units)", bytes, ptr);
    const size_t n = bytes >> 2;
    const int vec = ((uintptr_t)ptr & 15) == 0;
    size_t grid = ((vec ? n / 4 : n) + 255) / 256;
    grid = grid > 4096 ? 4096 : (grid ? grid : 1);
    hipLaunchKernelGGL(k_zero_words, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, (uint32_t*)ptr, n, vec);
    return ver_check_launch("ver_zero_async");
}

extern "C" int ver_abi_version(void) { return VER_ABI_VERSION; }
extern "C" const char* ver_last_error(void) { return g_err; }
