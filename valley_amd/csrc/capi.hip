// Error plumbing and ABI version for libvalley_hip.so.
#include "common.hpp"
#include "../../include/valley_hip.h"
#include <stdarg.h>
#include <stdio.h>

static thread_local char g_err[512] = "";

void vly_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int vly_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        vly_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return -(1000 + (int)e);
    }
    return 0;
}

extern "C" int vly_abi_version(void) { return VLY_ABI_VERSION; }
extern "C" const char* vly_last_error(void) { return g_err; }
extern "C" int vly_storage_dtype(void) { return VLY_FP16 ? VLY_STORAGE_FP16 : VLY_STORAGE_BF16; }
