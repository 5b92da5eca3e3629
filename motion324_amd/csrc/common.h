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
               TUN_ATTN_EXP, TUN_LN_ROWS, TUN_GEMM_PERSIST, TUN_ATTN_PWG, TUN_QKV_RING, TUN_NT_MB, TUN_HP, TUN_COUNT };
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

// eight consecutive values per lane: one 16-byte (bf16) or two 16-byte (fp32) accesses
__device__ __forceinline__ void load8(const bf16_t* p, float (&v)[8]) {
    const uint4 u = *reinterpret_cast<const uint4*>(p);
    v[0] = __uint_as_float(u.x << 16); v[1] = __uint_as_float(u.x & 0xFFFF0000u);
    v[2] = __uint_as_float(u.y << 16); v[3] = __uint_as_float(u.y & 0xFFFF0000u);
    v[4] = __uint_as_float(u.z << 16); v[5] = __uint_as_float(u.z & 0xFFFF0000u);
    v[6] = __uint_as_float(u.w << 16); v[7] = __uint_as_float(u.w & 0xFFFF0000u);
}
__device__ __forceinline__ void load8(const float* p, float (&v)[8]) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void store8(bf16_t* p, const float (&v)[8]) {
    *reinterpret_cast<uint4*>(p) = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
}
__device__ __forceinline__ void store8(float* p, const float (&v)[8]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
// exact (erf) GELU, nn.GELU() default -- reference model/transformer.py:58
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// Sum over the 64 lanes, delivered to every lane.  The four steps inside a row of 16 lanes are DPP adds (v_add_f32 with a
// quad_perm / row_half_mirror / row_mirror operand: no LDS crossbar, ~1 issue slot each); only the two steps across rows go
// through ds_bpermute.  A butterfly like the xor shuffles it replaces (every lane ends with the same, fixed-order sum:
// neighbours first, then quads, halves, rows), six dependent LDS round trips shorter.
template <int CTRL>
__device__ __forceinline__ float m324_dpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float wave_sum(float v) {
    v += m324_dpp<0xB1>(v);          // quad_perm [1, 0, 3, 2]   (lane ^ 1)
    v += m324_dpp<0x4E>(v);          // quad_perm [2, 3, 0, 1]   (lane ^ 2)
    v += m324_dpp<0x141>(v);         // row_half_mirror          (the other quad of the 8)
    v += m324_dpp<0x140>(v);         // row_mirror               (the other half of the 16)
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

// two independent sums at once: the steps of both butterflies interleave, so one's shuffle latency hides the other's
__device__ __forceinline__ void wave_sum2(float& a, float& b) {
    a += m324_dpp<0xB1>(a);  b += m324_dpp<0xB1>(b);
    a += m324_dpp<0x4E>(a);  b += m324_dpp<0x4E>(b);
    a += m324_dpp<0x141>(a); b += m324_dpp<0x141>(b);
    a += m324_dpp<0x140>(a); b += m324_dpp<0x140>(b);
    const float a16 = __shfl_xor(a, 16, 64), b16 = __shfl_xor(b, 16, 64);
    a += a16; b += b16;
    const float a32 = __shfl_xor(a, 32, 64), b32 = __shfl_xor(b, 32, 64);
    a += a32; b += b32;
}

static inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }
