// Error plumbing and device query for libm324.
#include <limits.h>
#include <stdarg.h>
#include <stdlib.h>
#include <atomic>
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

extern "C" int m324_abi_version(void) { return 22; }

// ---- tunables (see common.h): environment read once at load, then m324_set_tunable only
namespace {
struct TunDef { const char* env; int dflt; };
const TunDef kTun[m324::TUN_COUNT] = {
    {"M324_GEMM", 0}, {"M324_GEMM_TN", 0}, {"M324_XCD", 3}, {"M324_ATTN_NW", 0}, {"M324_ATTN_FLAT", 1},
    {"M324_ATTN_OCC", 0}, {"M324_ATTN_NQ2", 0}, {"M324_ATTN_BWD_NW", 0}, {"M324_ATTN_EXP", 0}, {"M324_LN_ROWS", 2}, {"M324_GEMM_PERSIST", 1}, {"M324_ATTN_PWG", 1}, {"M324_QKV_RING", 1}, {"M324_NT_MB", 128}, {"M324_HP", 6}};
std::atomic<int> g_tun[m324::TUN_COUNT];
int parse_tun(const char* e) { return (e[0] == 'v' || e[0] == 'V') ? atoi(e + 1) : atoi(e); }
struct TunInit {
    TunInit() {
        for (int i = 0; i < m324::TUN_COUNT; ++i) {
            const char* e = getenv(kTun[i].env);
            g_tun[i].store(e && e[0] ? parse_tun(e) : kTun[i].dflt, std::memory_order_relaxed);
        }
    }
} g_tun_init;
}  // namespace

int m324::tunable(int which) { return g_tun[which].load(std::memory_order_relaxed); }

extern "C" int m324_set_tunable(const char* name, int value) {
    M324_REQUIRE(name, "m324_set_tunable: null name");
    for (int i = 0; i < m324::TUN_COUNT; ++i)
        if (!strcmp(name, kTun[i].env)) {
            g_tun[i].store(value == INT_MIN ? kTun[i].dflt : value, std::memory_order_relaxed);
            return M324_OK;
        }
    M324_FAIL(M324_ERR_INVALID, "m324_set_tunable: unknown switch %s", name);
}

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
