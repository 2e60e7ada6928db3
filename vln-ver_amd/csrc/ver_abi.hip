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

extern "C" int ver_abi_version(void) { return VER_ABI_VERSION; }
extern "C" const char* ver_last_error(void) { return g_err; }
