// Error plumbing and device query for libm324.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include "common.h"

static thread_local char g_err[512] = "";

void m324_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int m324_abi_version(void) { return 9; }

extern "C" int m324_last_error(char* buf, int n) {
    if (!buf || n <= 0) return (int)strlen(g_err);
    strncpy(buf, g_err, (size_t)n - 1);
    buf[n - 1] = 0;
    return (int)strlen(buf);
}

extern "C" int m324_device_info(char* name, int n) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) M324_FAIL(M324_ERR_HIP, "hipGetDevice: %s", hipGetErrorString(e));
    hipDeviceProp_t p;
    e = hipGetDeviceProperties(&p, dev);
    if (e != hipSuccess) M324_FAIL(M324_ERR_HIP, "hipGetDeviceProperties: %s", hipGetErrorString(e));
    if (name && n > 0) {
        strncpy(name, p.gcnArchName, (size_t)n - 1);
        name[n - 1] = 0;
    }
    return p.multiProcessorCount;
}
