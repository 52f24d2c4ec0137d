// C-ABI plumbing shared by every entry point: error text, version, device probe.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "common.h"

thread_local char g_svae_err[512] = {0};

int svae_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_svae_err, sizeof(g_svae_err), fmt, ap);
    va_end(ap);
    return code;
}

extern "C" const char* svae_last_error(void) { return g_svae_err; }

extern "C" int svae_abi_version(void) { return 1; }

// 0 when the current device is a gfx950 part this library was compiled for.
extern "C" int svae_check_device(void) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return svae_fail((int)e, "svae_check_device: %s", hipGetErrorString(e));
    hipDeviceProp_t p;
    e = hipGetDeviceProperties(&p, dev);
    if (e != hipSuccess) return svae_fail((int)e, "svae_check_device: %s", hipGetErrorString(e));
    if (strncmp(p.gcnArchName, "gfx950", 6) != 0)
        return svae_fail(-2, "svae_check_device: built for gfx950, device is %s", p.gcnArchName);
    // the library's set-up call for this device: look up the static scratch of svae_masked_topk_static here, so
    // that entry point has nothing left to set up when it is called from code that must not touch the runtime
    return svae_scratch_setup(nullptr);
}
