// Shared device/host helpers for libm324 (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/m324.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

typedef uint16_t bf16_t;   // storage type of a bf16 element in HBM / LDS

// ---- error plumbing (never throw / abort across the C ABI) -------------------------------
void m324_set_error(const char* fmt, ...);
#define M324_FAIL(code, ...)            \
    do {                                \
        m324_set_error(__VA_ARGS__);    \
        return (code);                  \
    } while (0)
#define M324_REQUIRE(cond, ...)                                   \
    do {                                                          \
        if (!(cond)) M324_FAIL(M324_ERR_INVALID, __VA_ARGS__);    \
    } while (0)
#define M324_CHECK_LAUNCH(name)                                                             \
    do {                                                                                    \
        hipError_t e_ = hipGetLastError();                                                  \
        if (e_ != hipSuccess) M324_FAIL(M324_ERR_HIP, "%s: %s", name, hipGetErrorString(e_)); \
    } while (0)

// ---- tunables: A/B switches of the kernel choosers.  Read from the environment ONCE, when the library is loaded
// (M324_GEMM=v10, M324_ATTN_NW=8, ...); afterwards only m324_set_tunable() changes them (labs / tests).  No launch
// path calls getenv().
namespace m324 {
enum Tunable { TUN_GEMM = 0, TUN_GEMM_TN, TUN_XCD, TUN_ATTN_NW, TUN_ATTN_FLAT, TUN_ATTN_OCC, TUN_ATTN_NQ2, TUN_ATTN_BWD_NW,
               TUN_ATTN_EXP, TUN_COUNT };
int tunable(int which);          // 0 = "not set" for every switch except TUN_XCD (default 3) / TUN_ATTN_FLAT (default 1)
}  // namespace m324

// ---- bf16 <-> f32 (round-to-nearest-even, NaN preserved) ---------------------------------
__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// native conversions: one v_cvt_pk_bf16_f32 (RNE, NaN-preserving) per pair
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    f32x2 f = {lo, hi};
    bf16x2_t b = __builtin_convertvector(f, bf16x2_t);
    return *reinterpret_cast<uint32_t*>(&b);
}
__device__ __forceinline__ bf16_t f32_to_bf16(float f) { return (bf16_t)(pack_bf16x2(f, 0.f) & 0xffffu); }

template <typename T> struct Elem;
template <> struct Elem<float> {
    static constexpr int PER16 = 4;   // elements per 16-byte chunk
    __device__ static __forceinline__ float load(const float* p) { return *p; }
    __device__ static __forceinline__ void store(float* p, float v) { *p = v; }
};
template <> struct Elem<bf16_t> {
    static constexpr int PER16 = 8;
    __device__ static __forceinline__ float load(const bf16_t* p) { return bf16_to_f32(*p); }
    __device__ static __forceinline__ void store(bf16_t* p, float v) { *p = f32_to_bf16(v); }
};

// exact (erf) GELU, nn.GELU() default -- reference model/transformer.py:58
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

static inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }
