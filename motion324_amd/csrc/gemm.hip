// m324_gemm: C = epilogue(A[M,K] . W[N,K]^T) on gfx950 matrix cores.
//
// Both operands are K-major (nn.Linear stores W as [out, in]), so A rows feed the MFMA A operand and
// W rows feed the B operand with the same 16-byte-per-lane fragment load.
//
// Tile: 128 x 128 x (128 bytes of K) per 256-thread workgroup; 4 waves in a 2 x 2 grid, each wave a
// 64 x 64 output block = 2 x 2 MFMA 32x32 accumulators (64 AGPR/VGPR).  K-tile = 64 bf16 or 32 fp32,
// i.e. the LDS image has the same byte geometry in both precisions:
//     row r (0..127) at byte r*128, its 16-byte chunk c (0..7) stored at chunk c ^ ((r >> 1) & 7).
// The XOR makes the fragment read (32 lanes = 32 consecutive rows, same logical chunk) hit 16
// distinct 16-byte slots of the 256-byte LDS bank row per 16-lane group -> conflict-free
// ds_read_b128 (bf16) and 2-way ds_read_b64 (fp32 parity mode).
// Pipeline: global -> registers (issued before the MFMAs of the current tile) -> LDS (after them),
// double-buffered LDS, one barrier per K-tile.
#include <stdlib.h>
#include <type_traits>
#include "common.h"

namespace {

constexpr int BM = 128, BN = 128, ROWB = 128;           // ROWB: bytes of K per LDS row
constexpr int TILE_BYTES = BM * ROWB;                   // 16 KiB per operand per stage

struct Epilogue {
    const float* bias;
    const float* gamma;
    const float* residual;
    long ldr;
    int res_rows;
    int act;
    int row_gin, row_gout, row_off;
    long strideA, strideW, strideC;
    void* aux;
    long ldaux;
    int aux_mode;
    bf16_t* qkv_out[3];          // M324_AUX_QKV_HEADS: head-major q, k, v
    const float* qkv_w[2];       // RMSNorm weights of q, k (or null)
    float qkv_eps, qkv_qscale;
    int qkv_L, qkv_H;
};

// GELU for the bf16 path: erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7 absolute, far below the
// bf16 rounding of the result); the fp32 parity path keeps erff.  ~12 VALU + 2 transcendental ops
// instead of ocml's branchy erff -- the fc1 epilogue (128x128 GELUs per workgroup) is otherwise as
// long as its whole K = 768 main loop.
__device__ __forceinline__ float gelu_fast(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float erfc_z = p * t * __builtin_amdgcn_exp2f(-1.4426950408889634f * z * z);   // 1 - erf(z), z >= 0
    const float phi = x >= 0.f ? 1.0f - 0.5f * erfc_z : 0.5f * erfc_z;                       // Phi(x)
    return x * phi;
}

// Two GELUs per instruction stream for the bf16 epilogue: erf(z) = z P(z^2) on |z| <= 3 (odd minimax polynomial, 9
// coefficients, |erf error| <= 1.7e-5; beyond the clamp erf(3) = 0.99998 stands in for 1), evaluated with packed fp32
// FMAs (v_pk_fma_f32) and no transcendental.  |GELU error| <= 7e-5 absolute -- below the bf16 rounding step of every
// output larger than 0.02 -- at about a third of the issue slots of the rcp/exp form above.
typedef float f32x2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2v gelu_poly2(f32x2v x) {
    f32x2v z = x * 0.70710678118654752440f;
    z.x = __builtin_amdgcn_fmed3f(z.x, -3.0f, 3.0f);
    z.y = __builtin_amdgcn_fmed3f(z.y, -3.0f, 3.0f);
    const f32x2v t = z * z;
    f32x2v p = (f32x2v)(4.074096087e-08f);
    p = __builtin_elementwise_fma(p, t, (f32x2v)(-1.944782217e-06f));
    p = __builtin_elementwise_fma(p, t, (f32x2v)(4.105993727e-05f));
    p = __builtin_elementwise_fma(p, t, (f32x2v)(-5.110323815e-04f));
    p = __builtin_elementwise_fma(p, t, (f32x2v)(4.235408041e-03f));
    p = __builtin_elementwise_fma(p, t, (f32x2v)(-2.510281415e-02f));
    p = __builtin_elementwise_fma(p, t, (f32x2v)(1.110792751e-01f));
    p = __builtin_elementwise_fma(p, t, (f32x2v)(-3.753148415e-01f));
    p = __builtin_elementwise_fma(p, t, (f32x2v)(1.128268421e+00f));
    const f32x2v hx = x * 0.5f;
    return __builtin_elementwise_fma(hx, z * p, hx);
}

template <typename TOUT>
__device__ __forceinline__ void apply_gelu4(float4& v) {
    if constexpr (sizeof(TOUT) == 2) {
        f32x2v a = {v.x, v.y}, b = {v.z, v.w};
        a = gelu_poly2(a);
        b = gelu_poly2(b);
        v = make_float4(a.x, a.y, b.x, b.y);
    } else {
        v = make_float4(gelu_erf(v.x), gelu_erf(v.y), gelu_erf(v.z), gelu_erf(v.w));
    }
}

// d gelu(z) / dz = Phi(z) + z phi(z).  fp32 outputs: erff / expf; bf16 outputs: the polynomial erf above and exp2.
template <typename TOUT>
__device__ __forceinline__ float gelu_grad(float z) {
    if constexpr (sizeof(TOUT) == 2) {
        const float t = __builtin_amdgcn_fmed3f(z * 0.70710678118654752440f, -3.0f, 3.0f), t2 = t * t;
        float p = 4.074096087e-08f;
        p = fmaf(p, t2, -1.944782217e-06f); p = fmaf(p, t2, 4.105993727e-05f); p = fmaf(p, t2, -5.110323815e-04f);
        p = fmaf(p, t2, 4.235408041e-03f); p = fmaf(p, t2, -2.510281415e-02f); p = fmaf(p, t2, 1.110792751e-01f);
        p = fmaf(p, t2, -3.753148415e-01f); p = fmaf(p, t2, 1.128268421e+00f);
        const float cdf = fmaf(0.5f * t, p, 0.5f);
        const float pdf = 0.39894228040143267794f * __builtin_amdgcn_exp2f(-0.72134752044448170368f * z * z);
        return fmaf(z, pdf, cdf);
    } else {
        const float cdf = 0.5f * (1.0f + erff(z * 0.70710678118654752440f));
        const float pdf = 0.39894228040143267794f * expf(-0.5f * z * z);
        return cdf + z * pdf;
    }
}

template <typename TOUT>
__device__ __forceinline__ float4 load4_out(const TOUT* p) {
    if constexpr (sizeof(TOUT) == 2) {
        const uint2 u = *reinterpret_cast<const uint2*>(p);
        return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xFFFF0000u), __uint_as_float(u.y << 16),
                           __uint_as_float(u.y & 0xFFFF0000u));
    } else {
        return *reinterpret_cast<const float4*>(p);
    }
}

template <typename TOUT>
__device__ __forceinline__ float apply_gelu(float v) {
    if constexpr (sizeof(TOUT) == 2) return gelu_fast(v);
    else return gelu_erf(v);
}

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * ROWB + ((chunk ^ ((row >> 1) & 7)) << 4); }


// One K-tile of MFMAs for a wave's 64 x 64 block (2 x 2 accumulators of 32 x 32).
// SWAP = false: acc[i][j] = A_i . W_j^T  (C layout: lane = column n, registers = rows m)
// SWAP = true : acc[i][j] = W_j . A_i^T  (C layout: lane = row m, registers = 4-runs of columns n) --
//               the transposed accumulator lets the epilogue read/write 4 consecutive columns per lane.
template <typename TIN, bool SWAP = false>
__device__ __forceinline__ void mma_tile(const unsigned char* sa, const unsigned char* sb, int arow0, int brow0, int hi,
                                         f32x16 (&acc)[2][2]) {
    if constexpr (sizeof(TIN) == 2) {
        // bf16: 4 k-steps of 16; lane (row, hi) supplies k = ks*16 + hi*8 .. +7 = chunk ks*2 + hi
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 af[2], bfr[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                af[i] = *reinterpret_cast<const bf16x8*>(sa + lds_off(arow0 + i * 32, ks * 2 + hi));
                bfr[i] = *reinterpret_cast<const bf16x8*>(sb + lds_off(brow0 + i * 32, ks * 2 + hi));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = SWAP ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0)
                                     : __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
    } else {
        // f32 parity mode: chunk c holds k = 4c..4c+3; lane hi reads the 8 bytes at hi*8 of it
        // (k = 4c+2hi, 4c+2hi+1) and feeds them to two 32x32x2 MFMAs.  A and W use the same map,
        // so each k is contracted exactly once.
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            f32x2 af[2], bfr[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                af[i] = *reinterpret_cast<const f32x2*>(sa + lds_off(arow0 + i * 32, c) + hi * 8);
                bfr[i] = *reinterpret_cast<const f32x2*>(sb + lds_off(brow0 + i * 32, c) + hi * 8);
            }
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = SWAP ? __builtin_amdgcn_mfma_f32_32x32x2f32(bfr[j][e], af[i][e], acc[i][j], 0, 0, 0)
                                         : __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bfr[j][e], acc[i][j], 0, 0, 0);
        }
    }
}

// Epilogue of a wave's 64 x 64 block.  32x32 C layout: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
// Residual reads are issued as one batch of 16 unconditional loads per 32x32 block (indices clamped
// into range) so they overlap instead of paying one HBM round trip per element; stores are predicated.
// ACT: 0 none, 1 GELU, -1 decide at run time.  RES: 0 no residual / no row remap, 1 residual[m] (same row),
// 2 generic (row-modulo residual and/or output row remap), -1 decide at run time.
template <typename TOUT, int ACT, int RES>
__device__ __forceinline__ void store_tile_out(const f32x16 (&acc)[2][2], TOUT* C, long ldc, int M, int N, int mw, int nw,
                                               int l31, int hi, const Epilogue& ep) {
    const bool has_res = RES == 0 ? false : (RES == 1 ? true : ep.residual != nullptr);
    const bool res_mod = (RES == 0 || RES == 1) ? false : (ep.res_rows > 0 && ep.res_rows < M);
    const bool remap = (RES == 0 || RES == 1) ? false : ep.row_gin > 0;
    const bool gelu = ACT < 0 ? ep.act == M324_ACT_GELU : ACT == 1;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = nw + j * 32 + l31;
        const bool nok = n < N;
        const int nc = nok ? n : N - 1;
        const float bias = ep.bias ? ep.bias[nc] : 0.f;
        const float gamma = ep.gamma ? ep.gamma[nc] : 1.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int mb = mw + i * 32 + 4 * hi;
            float res[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) res[r] = 0.f;
            if (has_res) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    int mc = min(mb + (r & 3) + 8 * (r >> 2), M - 1);
                    if (res_mod) mc %= ep.res_rows;
                    res[r] = ep.residual[(long)mc * ep.ldr + nc];
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mb + (r & 3) + 8 * (r >> 2);
                float v = acc[i][j][r] + bias;
                if (gelu) v = apply_gelu<TOUT>(v);
                v = fmaf(v, gamma, res[r]);
                long orow = m;
                if (remap) orow = (long)(m / ep.row_gin) * ep.row_gout + (m % ep.row_gin) + ep.row_off;
                if (nok && m < M) Elem<TOUT>::store(C + orow * ldc + n, v);
            }
        }
    }
}


// SWAPPED accumulators (all LDS-DMA kernels): lane = output row m (32 rows per block), registers = columns
// n = 8*g + 4*hi + e (g = r >> 2, e = r & 3): every lane owns runs of 4 consecutive columns.  Needs N % 4 == 0.
template <typename TOUT>
__device__ __forceinline__ void store4_out(TOUT* p, float a, float b, float c, float d);
template <>
__device__ __forceinline__ void store4_out<float>(float* p, float a, float b, float c, float d) {
    *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d);
}
template <>
__device__ __forceinline__ void store4_out<bf16_t>(bf16_t* p, float a, float b, float c, float d) {
    *reinterpret_cast<uint2*>(p) = make_uint2(pack_bf16x2(a, b), pack_bf16x2(c, d));
}

// LDS-transposed epilogue.  The swapped accumulator layout gives a lane 4-column runs of ONE row, so a direct store
// instruction touches 32 different rows with 16-32 bytes each: the texture path handles one cache line per cycle and
// the epilogue of a K = 768 GEMM cost as much as half its main loop.  Here every wave bounces each 32 x 64 block
// through a wave-private LDS scratch (32 rows x 68 floats: the 4-float pad makes both the b128 writes -- 8 lanes = 8
// rows -- and the row-contiguous b128 reads conflict-free) and then works on rows: 16 lanes cover the 64 columns of a
// row, so bias / gamma are per-lane constants, residual reads and output stores are whole 128/256-byte lines.
// DS operations of one wave execute in order, so no barrier or wait is needed between the write and read passes.
constexpr int EP_LD = 68;
constexpr int EP_WAVE_FLOATS = 32 * EP_LD;     // 8704 bytes per wave

template <typename TOUT, int ACT, int RES, int MI>
__device__ __forceinline__ void store_tile_lds(const f32x16 (&acc)[MI][2], float* scr, TOUT* C, long ldc, int M, int N, int mw,
                                               int nw, int lane, const Epilogue& ep) {
    const bool has_res = RES == 0 ? false : (RES == 1 ? true : ep.residual != nullptr);
    const bool res_mod = (RES == 0 || RES == 1) ? false : (ep.res_rows > 0 && ep.res_rows < M);
    const bool remap = (RES == 0 || RES == 1) ? false : ep.row_gin > 0;
    const int l31 = lane & 31, hi = lane >> 5;
    const int rr = lane >> 4, cc = (lane & 15) * 4;
    const int n = nw + cc;
    const bool nok = n < N;                    // N % 4 == 0: a lane's 4 columns are all in or all out
    const int ncl = min(n, N - 4);
    const float4 bi = ep.bias ? *reinterpret_cast<const float4*>(ep.bias + ncl) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 ga = ep.gamma ? *reinterpret_cast<const float4*>(ep.gamma + ncl) : make_float4(1.f, 1.f, 1.f, 1.f);
    float* wr = scr + l31 * EP_LD + 4 * hi;
    const float* rd = scr + rr * EP_LD + cc;
    // interior tiles (all but the last row / column of tiles) take a copy without per-store predicates, so the 8 LDS
    // reads and the 8 stores of a block are scheduled as batches instead of read-wait-store chains
    auto body = [&](auto checked) {
        constexpr bool CHECK = decltype(checked)::value;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int mb = mw + i * 32 + rr;       // row of pass p: mb + 4 p
            float4 res[8], az[8];
            if (has_res) {
#pragma unroll
                for (int p = 0; p < 8; ++p) {
                    int mr = CHECK ? min(mb + 4 * p, M - 1) : mb + 4 * p;
                    if (res_mod) mr %= ep.res_rows;
                    res[p] = *reinterpret_cast<const float4*>(ep.residual + (long)mr * ep.ldr + ncl);
                }
            }
            if constexpr (ACT == 3) {
#pragma unroll
                for (int p = 0; p < 8; ++p) {
                    const int mr = CHECK ? min(mb + 4 * p, M - 1) : mb + 4 * p;
                    az[p] = load4_out<TOUT>(static_cast<const TOUT*>(ep.aux) + (long)mr * ep.ldaux + ncl);
                }
            }
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<float4*>(wr + j * 32 + 8 * g) =
                        make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
            float4 v[8];
#pragma unroll
            for (int p = 0; p < 8; ++p) v[p] = *reinterpret_cast<const float4*>(rd + p * 4 * EP_LD);
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                float4 x = v[p];
                x.x += bi.x; x.y += bi.y; x.z += bi.z; x.w += bi.w;
                const int m = mb + 4 * p;
                if (ACT == 2 && (!CHECK || (m < M && nok)))
                    store4_out<TOUT>(static_cast<TOUT*>(ep.aux) + (long)m * ep.ldaux + n, x.x, x.y, x.z, x.w);
                if (ACT == 1 || ACT == 2) apply_gelu4<TOUT>(x);
                if (ep.gamma) { x.x *= ga.x; x.y *= ga.y; x.z *= ga.z; x.w *= ga.w; }
                if (has_res) { x.x += res[p].x; x.y += res[p].y; x.z += res[p].z; x.w += res[p].w; }
                if constexpr (ACT == 3) {
                    x.x *= gelu_grad<TOUT>(az[p].x); x.y *= gelu_grad<TOUT>(az[p].y);
                    x.z *= gelu_grad<TOUT>(az[p].z); x.w *= gelu_grad<TOUT>(az[p].w);
                }
                if (!CHECK || (m < M && nok)) {
                    long orow = m;
                    if (remap) orow = (long)(m / ep.row_gin) * ep.row_gout + (m % ep.row_gin) + ep.row_off;
                    store4_out<TOUT>(C + orow * ldc + n, x.x, x.y, x.z, x.w);
                }
            }
        }
    };
    // bf16 interior tiles without residual / row remap: 8 columns per lane, so a row is 8 lanes x 16 bytes and one
    // store instruction writes 8 whole 128-byte lines (half as many store instructions as the 4-column form)
    auto body8 = [&]() {
        const int r8 = lane >> 3, c8 = (lane & 7) * 8;
        const int n8 = nw + c8;
        float4 b0 = make_float4(0.f, 0.f, 0.f, 0.f), b1 = b0, g0 = make_float4(1.f, 1.f, 1.f, 1.f), g1 = g0;
        if (ep.bias) { b0 = *reinterpret_cast<const float4*>(ep.bias + n8); b1 = *reinterpret_cast<const float4*>(ep.bias + n8 + 4); }
        if (ep.gamma) { g0 = *reinterpret_cast<const float4*>(ep.gamma + n8); g1 = *reinterpret_cast<const float4*>(ep.gamma + n8 + 4); }
        const float* rd8 = scr + r8 * EP_LD + c8;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<float4*>(wr + j * 32 + 8 * g) =
                        make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
            float4 v0[4], v1[4], z0[4], z1[4];
            if constexpr (ACT == 3) {
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const TOUT* zp = static_cast<const TOUT*>(ep.aux) + (long)(mw + i * 32 + p * 8 + r8) * ep.ldaux + n8;
                    z0[p] = load4_out<TOUT>(zp);
                    z1[p] = load4_out<TOUT>(zp + 4);
                }
            }
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                v0[p] = *reinterpret_cast<const float4*>(rd8 + p * 8 * EP_LD);
                v1[p] = *reinterpret_cast<const float4*>(rd8 + p * 8 * EP_LD + 4);
            }
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                float4 x = v0[p], y = v1[p];
                x.x += b0.x; x.y += b0.y; x.z += b0.z; x.w += b0.w;
                y.x += b1.x; y.y += b1.y; y.z += b1.z; y.w += b1.w;
                const long m = mw + i * 32 + p * 8 + r8;
                if constexpr (ACT == 2)
                    *reinterpret_cast<uint4*>(static_cast<TOUT*>(ep.aux) + m * ep.ldaux + n8) =
                        make_uint4(pack_bf16x2(x.x, x.y), pack_bf16x2(x.z, x.w), pack_bf16x2(y.x, y.y), pack_bf16x2(y.z, y.w));
                if (ACT == 1 || ACT == 2) { apply_gelu4<TOUT>(x); apply_gelu4<TOUT>(y); }
                if (ep.gamma) {
                    x.x *= g0.x; x.y *= g0.y; x.z *= g0.z; x.w *= g0.w;
                    y.x *= g1.x; y.y *= g1.y; y.z *= g1.z; y.w *= g1.w;
                }
                if constexpr (ACT == 3) {
                    x.x *= gelu_grad<TOUT>(z0[p].x); x.y *= gelu_grad<TOUT>(z0[p].y);
                    x.z *= gelu_grad<TOUT>(z0[p].z); x.w *= gelu_grad<TOUT>(z0[p].w);
                    y.x *= gelu_grad<TOUT>(z1[p].x); y.y *= gelu_grad<TOUT>(z1[p].y);
                    y.z *= gelu_grad<TOUT>(z1[p].z); y.w *= gelu_grad<TOUT>(z1[p].w);
                }
                *reinterpret_cast<uint4*>(C + m * ldc + n8) = make_uint4(pack_bf16x2(x.x, x.y), pack_bf16x2(x.z, x.w),
                                                                         pack_bf16x2(y.x, y.y), pack_bf16x2(y.z, y.w));
            }
        }
    };
    // fused q|k|v projection (M324_AUX_QKV_HEADS): this wave's 64 columns are one head of q (which = 0), k (1) or v (2).
    // 8 columns per lane: a row's 64 values sit in the 8 lanes that share lane >> 3, so the per-head RMSNorm is three
    // xor-shuffles inside the group, and a (token, head) row leaves as 8 x 16 bytes = one 128-byte line.
    auto body_qkv = [&]() {
        const int r8 = lane >> 3, c8 = (lane & 7) * 8;
        const int hc = ep.qkv_H * 64, which = nw / hc, head = (nw % hc) >> 6;
        const int n8 = nw + c8;
        float4 b0 = make_float4(0.f, 0.f, 0.f, 0.f), b1 = b0, w0 = make_float4(1.f, 1.f, 1.f, 1.f), w1 = w0;
        if (ep.bias) { b0 = *reinterpret_cast<const float4*>(ep.bias + n8); b1 = *reinterpret_cast<const float4*>(ep.bias + n8 + 4); }
        const bool norm = which < 2 && ep.qkv_w[which] != nullptr;
        if (norm) {
            w0 = *reinterpret_cast<const float4*>(ep.qkv_w[which] + c8);
            w1 = *reinterpret_cast<const float4*>(ep.qkv_w[which] + c8 + 4);
        }
        const float post = which == 0 ? ep.qkv_qscale : 1.0f;
        bf16_t* const base = ep.qkv_out[which] + (long)head * ep.qkv_L * 64 + c8;
        const float* rd8 = scr + r8 * EP_LD + c8;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<float4*>(wr + j * 32 + 8 * g) =
                        make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
            float4 v0[4], v1[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                v0[p] = *reinterpret_cast<const float4*>(rd8 + p * 8 * EP_LD);
                v1[p] = *reinterpret_cast<const float4*>(rd8 + p * 8 * EP_LD + 4);
            }
            // token of pass p: m = mw + 32 i + r8 + 8 p  ->  (batch, position); one division per block, then steps of 8
            const int m0r = mw + i * 32 + r8;
            int bb = m0r / ep.qkv_L, ll = m0r - bb * ep.qkv_L;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                float4 x = v0[p], y = v1[p];
                x.x += b0.x; x.y += b0.y; x.z += b0.z; x.w += b0.w;
                y.x += b1.x; y.y += b1.y; y.z += b1.z; y.w += b1.w;
                float rs = post;
                if (norm) {
                    float ss = x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w + y.x * y.x + y.y * y.y + y.z * y.z + y.w * y.w;
                    ss += __shfl_xor(ss, 1, 64); ss += __shfl_xor(ss, 2, 64); ss += __shfl_xor(ss, 4, 64);
                    rs *= rsqrtf(ss * (1.0f / 64.0f) + ep.qkv_eps);
                }
                x.x *= rs * w0.x; x.y *= rs * w0.y; x.z *= rs * w0.z; x.w *= rs * w0.w;
                y.x *= rs * w1.x; y.y *= rs * w1.y; y.z *= rs * w1.z; y.w *= rs * w1.w;
                if (m0r + 8 * p < M && n8 < N)
                    *reinterpret_cast<uint4*>(base + (((long)bb * ep.qkv_H) * ep.qkv_L + ll) * 64) =
                        make_uint4(pack_bf16x2(x.x, x.y), pack_bf16x2(x.z, x.w), pack_bf16x2(y.x, y.y), pack_bf16x2(y.z, y.w));
                ll += 8;
                while (ll >= ep.qkv_L) { ll -= ep.qkv_L; ++bb; }
            }
        }
    };
    const bool interior = mw + MI * 32 <= M && nw + 64 <= N;
    if constexpr (ACT == 4) {
        body_qkv();
        return;
    }
    if constexpr (sizeof(TOUT) == 2 && RES == 0 && ACT != 4) {
        if (interior && (ldc & 7) == 0 && (reinterpret_cast<uintptr_t>(C) & 15) == 0 &&
            (!ep.bias || (reinterpret_cast<uintptr_t>(ep.bias) & 15) == 0) &&
            (ACT < 2 || ((ep.ldaux & 7) == 0 && (reinterpret_cast<uintptr_t>(ep.aux) & 15) == 0))) {
            body8();
            return;
        }
    }
    if (interior) body(std::false_type{});
    else body(std::true_type{});
}

template <typename TIN, typename TOUT>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const TIN* __restrict__ A, long lda, const TIN* __restrict__ W,
                                                      long ldw, TOUT* C, long ldc, int M, int N, int K,
                                                      Epilogue ep) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[4 * TILE_BYTES];   // A0 B0 A1 B1
    constexpr int EPC = Elem<TIN>::PER16;        // elements per 16-byte chunk
    constexpr int BK = ROWB / sizeof(TIN);       // 64 (bf16) or 32 (f32)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, hi = lane >> 5;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;

    // staging assignment: 4 chunks of A and 4 of W per thread per K-tile
    int srow[4], schk[4];
    const TIN* ga[4];
    const TIN* gb[4];
    bool aval[4], bval[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int id = tid + 256 * i;
        srow[i] = id >> 3;
        schk[i] = id & 7;
        int am = m0 + srow[i];
        aval[i] = am < M;
        ga[i] = A + (long)(aval[i] ? am : 0) * lda + schk[i] * EPC;
        int bn = n0 + srow[i];
        bval[i] = bn < N;
        gb[i] = W + (long)(bval[i] ? bn : 0) * ldw + schk[i] * EPC;
    }
    uint4 ra[4], rb[4];
    auto load_tile = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ra[i] = aval[i] ? *reinterpret_cast<const uint4*>(ga[i] + (long)kt * BK) : make_uint4(0, 0, 0, 0);
            rb[i] = bval[i] ? *reinterpret_cast<const uint4*>(gb[i] + (long)kt * BK) : make_uint4(0, 0, 0, 0);
        }
    };
    auto store_tile = [&](int buf) {
        unsigned char* sa = smem + buf * 2 * TILE_BYTES;
        unsigned char* sb = sa + TILE_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int o = lds_off(srow[i], schk[i]);
            *reinterpret_cast<uint4*>(sa + o) = ra[i];
            *reinterpret_cast<uint4*>(sb + o) = rb[i];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = K / BK;
    load_tile(0);
    store_tile(0);
    __syncthreads();

    const int arow0 = wm * 64 + l31, brow0 = wn * 64 + l31;
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
        if (more) load_tile(kt + 1);
        const unsigned char* sa = smem + (kt & 1) * 2 * TILE_BYTES;
        const unsigned char* sb = sa + TILE_BYTES;
        mma_tile<TIN>(sa, sb, arow0, brow0, hi, acc);
        if (more) store_tile((kt + 1) & 1);
        __syncthreads();
    }

    store_tile_out<TOUT, -1, -1>(acc, C, ldc, M, N, m0 + wm * 64, n0 + wn * 64, l31, hi, ep);
}


// ------------------------------------------------------------------------------------------------
// v2: same tile / LDS image / MFMA schedule, but the K-tiles are staged by LDS-DMA
// (global_load_lds_dwordx4: HBM -> LDS without a VGPR round trip and without the ds_write pass that
// made v1 LDS-bound: 32 KiB of ds_write_b128 per K-tile at ~79 B/clk/CU is ~415 cycles against 512
// cycles of MFMA).  A wave-instruction writes 1 KiB = 8 tile rows linearly (LDS address = wave-uniform
// base + lane * 16), so the XOR swizzle is applied to the SOURCE address: lane l fills row
// r = 8*g + (l >> 3), slot p = l & 7, which must hold logical chunk c = p ^ ((r >> 1) & 7).
// Rows past M (or N) are clamped to the last valid row instead of zero-filled: an output row depends
// only on its own A row / W row, and rows >= M, columns >= N are never stored.
// Double-buffered; the barrier at the top of iteration kt both publishes tile kt (hipcc drains the
// LDS-DMA queue, vmcnt(0), before s_barrier) and retires every wave's reads of the buffer that tile
// kt+1 is about to overwrite.
typedef __attribute__((address_space(3))) void lds_ptr_t;
typedef __attribute__((address_space(1))) const void glb_ptr_t;

template <typename TIN, typename TOUT, int ACT, int RES>
__global__ __launch_bounds__(256, 2) void gemm_glds_kernel(const TIN* __restrict__ A, long lda, const TIN* __restrict__ W,
                                                           long ldw, TOUT* C, long ldc, int M, int N, int K, Epilogue ep,
                                                           int ntn, int xcd_remap) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[4 * TILE_BYTES];   // A0 B0 A1 B1
    constexpr int EPC = Elem<TIN>::PER16;
    constexpr int BK = ROWB / sizeof(TIN);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, hi = lane >> 5;
    // 1-D grid.  Workgroup b runs on XCD b % 8 (observed dispatch order; used for speed only): give every
    // XCD a contiguous range of logical tiles = whole A row-panels with all their column tiles, so a panel
    // is pulled into ONE XCD's L2 and re-used by its ntn column tiles instead of being fetched by all eight.
    int lid = blockIdx.x;
    if (xcd_remap & 1) {
        const int nb = gridDim.x, q = nb >> 3, r = nb & 7, x = lid & 7, loc = lid >> 3;
        lid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + loc;
    }
    const int m0 = (lid / ntn) * BM, n0 = (lid % ntn) * BN;
    A += (long)blockIdx.y * ep.strideA;      // batched launch (split-K weight gradients): blockIdx.y = problem index
    W += (long)blockIdx.y * ep.strideW;
    C += (long)blockIdx.y * ep.strideC;

    // this wave stages row groups g = wave*4 + i (8 rows each) of both operands
    const TIN* ga[4];
    const TIN* gb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (wave * 4 + i) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        ga[i] = A + (long)min(m0 + r, M - 1) * lda + c * EPC;
        gb[i] = W + (long)min(n0 + r, N - 1) * ldw + c * EPC;
    }
    auto issue_tile = [&](int kt, int buf) {
        unsigned char* sa = smem + buf * 2 * TILE_BYTES + wave * 4096;
        unsigned char* sb = sa + TILE_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_global_load_lds((glb_ptr_t*)(ga[i] + (long)kt * BK), (lds_ptr_t*)(sa + i * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_ptr_t*)(gb[i] + (long)kt * BK), (lds_ptr_t*)(sb + i * 1024), 16, 0, 0);
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = K / BK;
    const int arow0 = wm * 64 + l31, brow0 = wn * 64 + l31;
    issue_tile(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // tile kt landed (never left to __syncthreads()'s fence)
        __syncthreads();
        if (kt + 1 < nk) issue_tile(kt + 1, (kt + 1) & 1);
        const unsigned char* sa = smem + (kt & 1) * 2 * TILE_BYTES;
        mma_tile<TIN, true>(sa, sa + TILE_BYTES, arow0, brow0, hi, acc);
    }
    __syncthreads();                            // all waves are done reading the stages: reuse them as epilogue scratch
    store_tile_lds<TOUT, ACT, RES, 2>(acc, reinterpret_cast<float*>(smem) + wave * EP_WAVE_FLOATS, C, ldc, M, N, m0 + wm * 64,
                                      n0 + wn * 64, lane, ep);
}


// ------------------------------------------------------------------------------------------------
// v5: 256 x 256 tile, 8 waves as 2 (M) x 4 (N), each wave a 128 x 64 block = 4 x 2 accumulators
// (128 VGPR).  Per K-tile a wave issues 32 MFMAs (1024 matrix-pipe cycles) against 24 fragment reads,
// and the workgroup moves 64 KiB by LDS-DMA per 8.4 MFLOP -- half the LDS traffic per FLOP of the
// 128-wide tiles and twice the work per barrier.  Two LDS stages of 64 KiB (one workgroup per CU).
// Used where 256-wide column tiles quantise well (N % 256 == 0) and the grid still fills the chip.
constexpr int BM5 = 256, BN5 = 256;
constexpr int STAGE5 = (BM5 + BN5) * ROWB;   // 64 KiB

template <typename TIN, typename TOUT, int ACT, int RES>
__global__ __launch_bounds__(512, 2) void gemm_glds5_kernel(const TIN* __restrict__ A, long lda, const TIN* __restrict__ W,
                                                            long ldw, TOUT* C, long ldc, int M, int N, int K, Epilogue ep,
                                                            int ntn, int xcd_remap) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * STAGE5];
    constexpr int EPC = Elem<TIN>::PER16;
    constexpr int BK = ROWB / sizeof(TIN);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int l31 = lane & 31, hi = lane >> 5;
    int lid = blockIdx.x;
    if (xcd_remap & 1) {
        const int nb = gridDim.x, q = nb >> 3, r = nb & 7, x = lid & 7, loc = lid >> 3;
        lid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + loc;
    }
    const int m0 = (lid / ntn) * BM5, n0 = (lid % ntn) * BN5;

    // staging: 32 row groups of 8 rows per operand; wave w takes groups w*4 .. w*4+3 of A and of W
    const TIN* ga[4];
    const TIN* gb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (wave * 4 + i) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        ga[i] = A + (long)min(m0 + r, M - 1) * lda + c * EPC;
        gb[i] = W + (long)min(n0 + r, N - 1) * ldw + c * EPC;
    }
    auto issue_tile = [&](int kt, int buf) {
        unsigned char* sa = smem + buf * STAGE5 + wave * 4096;
        unsigned char* sb = sa + BM5 * ROWB;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_global_load_lds((glb_ptr_t*)(ga[i] + (long)kt * BK), (lds_ptr_t*)(sa + i * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_ptr_t*)(gb[i] + (long)kt * BK), (lds_ptr_t*)(sb + i * 1024), 16, 0, 0);
        }
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = K / BK;
    const int arow0 = wm * 128 + l31, brow0 = wn * 64 + l31;
    issue_tile(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // tile kt landed (never left to __syncthreads()'s fence)
        __syncthreads();
        if (kt + 1 < nk) issue_tile(kt + 1, (kt + 1) & 1);
        const unsigned char* sa = smem + (kt & 1) * STAGE5;
        const unsigned char* sb = sa + BM5 * ROWB;
        if constexpr (sizeof(TIN) == 2) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                bf16x8 af[4], bfr[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) bfr[j] = *reinterpret_cast<const bf16x8*>(sb + lds_off(brow0 + j * 32, ks * 2 + hi));
#pragma unroll
                for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const bf16x8*>(sa + lds_off(arow0 + i * 32, ks * 2 + hi));
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                f32x2 af[4], bfr[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) bfr[j] = *reinterpret_cast<const f32x2*>(sb + lds_off(brow0 + j * 32, c) + hi * 8);
#pragma unroll
                for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const f32x2*>(sa + lds_off(arow0 + i * 32, c) + hi * 8);
#pragma unroll
                for (int e = 0; e < 2; ++e)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bfr[j][e], af[i][e], acc[i][j], 0, 0, 0);
            }
        }
    }
    __syncthreads();
    store_tile_lds<TOUT, ACT, RES, 4>(acc, reinterpret_cast<float*>(smem) + wave * EP_WAVE_FLOATS, C, ldc, M, N, m0 + wm * 128,
                                      n0 + wn * 64, lane, ep);
}


// ------------------------------------------------------------------------------------------------
// 64-byte-row half-tile image shared by the pipelined kernel: one 32 KiB LDS slot = A[256][32] + W[256][32] (bf16),
// 16-byte chunks swizzled c ^ ((row >> 2) & 3) (applied to the LDS-DMA source address and to the fragment reads).
constexpr int ROWB6 = 64;                        // bytes of K per LDS row in a half-tile
constexpr int PART6 = 256 * ROWB6;               // 16 KiB: one operand's half-tile
constexpr int SLOT6 = 2 * PART6;                 // 32 KiB
__device__ __forceinline__ int lds_off6(int row, int chunk) { return row * ROWB6 + ((chunk ^ ((row >> 2) & 3)) << 4); }

#define M324_BARRIER()                             \
    do {                                           \
        asm volatile("s_barrier" ::: "memory");    \
        __builtin_amdgcn_sched_barrier(0);         \
    } while (0)

// ------------------------------------------------------------------------------------------------
// v7: software-pipelined 256 x 256 kernel (bf16), 8 waves as 2 (M) x 4 (N), each wave a 128 x 64 block = 4 x 2
// accumulators.  K is streamed in half-tiles of 32 through a 4-slot ring (128 KiB, one workgroup per CU) with ONE
// barrier per half-tile, and the overlap is done inside each wave: while the 8 MFMAs of a k-step run, the 6 fragment
// reads of the next k-step and the wave's LDS-DMA pieces of half-tile h+3 are issued between them (pinned with
// sched_group_barrier), so the matrix pipe is fed without relying on the SIMD partner's phase.
//   barrier h (top of iteration h): every wave's pieces of half-tile h+1 have landed (counted vmcnt: only the 4 pieces
//   of h+2 may still fly) and every wave has finished reading half-tile h-1, whose slot now receives half-tile h+3.
//   Fragments of (h+1, k-step 0) are fetched during (h, k-step 1), i.e. before barrier h+1: MFMAs restart immediately.
//   The steady state is branch-free: past the end of K the last half-tile is fetched again into a free slot (an L2
//   hit nobody reads), so the wait is always vmcnt(4); all pieces are drained before the ring becomes epilogue scratch.
// Measured (tools/gemm_lab, 4096^3): MFMA-only loop 1825 TF/s, + fragment reads 1490, + LDS-DMA issue 1316, all 1190.
template <typename TOUT, int ACT, int RES>
__global__ __launch_bounds__(512, 2) void gemm_pipe_kernel(const bf16_t* __restrict__ A, long lda, const bf16_t* __restrict__ W,
                                                           long ldw, TOUT* C, long ldc, int M, int N, int K, Epilogue ep, int ntn,
                                                           int xcd_remap) {
    constexpr int RING = 4, PPW = 4;             // ring slots; LDS-DMA pieces per wave per half-tile (2 of A, 2 of W)
    __shared__ __attribute__((aligned(1024))) unsigned char smem[RING * SLOT6];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int l31 = lane & 31, hi = lane >> 5;
    int lid = blockIdx.x;
    if (xcd_remap & 1) {
        const int nb = gridDim.x, q = nb >> 3, r = nb & 7, x = lid & 7, loc = lid >> 3;
        lid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + loc;
    }
    const int m0 = (lid / ntn) * BM5, n0 = (lid % ntn) * BN5;

    // LDS-DMA: a wave-instruction fills 16 rows x 64 B; wave w moves row groups 2w, 2w+1 of A and of W
    const bf16_t* ga[2];
    const bf16_t* gb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = (wave * 2 + i) * 16 + (lane >> 2);
        const int c = ((lane & 3) ^ ((r >> 2) & 3)) * 8;
        ga[i] = A + (long)min(m0 + r, M - 1) * lda + c;
        gb[i] = W + (long)min(n0 + r, N - 1) * ldw + c;
    }
    auto issue_a = [&](int h, int slot) {
        unsigned char* sa = smem + slot * SLOT6 + wave * 2048;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((glb_ptr_t*)(ga[i] + (long)h * 32), (lds_ptr_t*)(sa + i * 1024), 16, 0, 0);
    };
    auto issue_b = [&](int h, int slot) {
        unsigned char* sb = smem + slot * SLOT6 + PART6 + wave * 2048;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((glb_ptr_t*)(gb[i] + (long)h * 32), (lds_ptr_t*)(sb + i * 1024), 16, 0, 0);
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int NH = K / 32;
    const int aoff = lds_off6(wm * 128 + l31, hi), boff = PART6 + lds_off6(wn * 64 + l31, hi);
    bf16x8 fa[2][4], fb[2][2];
    auto load_frags = [&](int set, int slot, int ks) {
        const unsigned char* base = smem + slot * SLOT6;
        const int x = ks << 5;          // k-step 1 = chunk + 2 = byte offset ^ 32 (the swizzle only touches bits 4-5)
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[set][j] = *reinterpret_cast<const bf16x8*>(base + ((boff + j * 2048) ^ x));
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[set][i] = *reinterpret_cast<const bf16x8*>(base + ((aoff + i * 2048) ^ x));
    };
    auto mma8 = [&](int set) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[set][j], fa[set][i], acc[i][j], 0, 0, 0);
    };
    // k-step schedule (sched_group_barrier: 0x008 MFMA, 0x020 VMEM read, 0x100 DS read): the 6 fragment reads of the next
    // k-step ride on the first 6 MFMAs (landed when that k-step starts), the 2 LDS-DMA pieces on the last two.  LDS-DMA
    // instructions also match the DS mask: they come after the reads in program order, so the DS groups take the reads.
#define M324_SG(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
    auto sched_kstep = [&]() {
        M324_SG(0x008, 1); M324_SG(0x100, 1); M324_SG(0x008, 1); M324_SG(0x100, 1);
        M324_SG(0x008, 1); M324_SG(0x100, 1); M324_SG(0x008, 1); M324_SG(0x100, 1);
        M324_SG(0x008, 1); M324_SG(0x100, 1); M324_SG(0x008, 1); M324_SG(0x100, 1);
        M324_SG(0x008, 1); M324_SG(0x020, 1); M324_SG(0x008, 1); M324_SG(0x020, 1);
    };
#define M324_WAIT_PIECES(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")

    for (int h = 0; h < RING - 1; ++h) {
        issue_a(h < NH ? h : NH - 1, h);
        issue_b(h < NH ? h : NH - 1, h);
    }
    M324_WAIT_PIECES(2 * PPW);                   // half-tile 0 landed
    M324_BARRIER();
    load_frags(0, 0, 0);
    int slot = 0;                                // h % RING
    for (int h = 0; h < NH; ++h) {
        const int nslot = (slot + 1) & 3, fslot = (slot + 3) & 3;
        const int hn = h + 3 < NH ? h + 3 : NH - 1;
        M324_WAIT_PIECES(PPW);                   // half-tile h+1 landed (this wave's pieces)
        M324_BARRIER();
        load_frags(1, slot, 1);
        issue_a(hn, fslot);
        mma8(0);
        sched_kstep();
        load_frags(0, nslot, 0);                 // past the end: stale but valid LDS, never used
        issue_b(hn, fslot);
        mma8(1);
        sched_kstep();
        slot = nslot;
    }
    M324_WAIT_PIECES(0);                         // no LDS-DMA may outlive the main loop: the ring becomes scratch
#undef M324_WAIT_PIECES
#undef M324_SG
    M324_BARRIER();
    store_tile_lds<TOUT, ACT, RES, 4>(acc, reinterpret_cast<float*>(smem) + wave * EP_WAVE_FLOATS, C, ldc, M, N, m0 + wm * 128,
                                      n0 + wn * 64, lane, ep);
}

// ------------------------------------------------------------------------------------------------
// v9: skinny GEMM for M <= 64 (the 64 latent tokens of the shape encoder: every projection of the 4 point-transformer
// blocks and of the encoder cross-attention at B = 1).  A 128 x 128 tile kernel runs these on N / 128 = 6..24 CUs with the
// whole K loop serial (17-48 us for 0.1-0.3 GFLOP).  Here a workgroup owns 32 output columns, its 8 waves split K in
// interleaved 64-element chunks, operands go global -> registers directly in MFMA fragment order (no LDS: every
// element is used once per workgroup), and the eight partial 64 x 32 blocks are summed through LDS in a fixed order.
// k-assignment inside a chunk: lane half `hi` owns k = 32 hi .. 32 hi + 31 (four 16-byte fragments = four MFMA steps);
// A and W use the same assignment, so every k is contracted exactly once (only the summation order differs from the
// tile kernels).
constexpr int SK_LD = 36;      // floats per partial row: 32 + 4 pad (conflict-free b128 writes: 8 lanes = 8 rows)
constexpr int SK_WAVES = 8;

template <typename TOUT, int ACT>
__global__ __launch_bounds__(512) void gemm_skinny_kernel(const bf16_t* __restrict__ A, long lda, const bf16_t* __restrict__ W,
                                                          long ldw, TOUT* C, long ldc, int M, int N, int K, Epilogue ep) {
    __shared__ __attribute__((aligned(16))) float part[SK_WAVES * 64 * SK_LD];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int n0 = blockIdx.x * 32;
    const bf16_t* a0 = A + (long)min(l31, M - 1) * lda + hi * 32;
    const bf16_t* a1 = A + (long)min(32 + l31, M - 1) * lda + hi * 32;
    const bf16_t* w0 = W + (long)min(n0 + l31, N - 1) * ldw + hi * 32;
    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const int nchunk = K / 64;
    bf16x8 fa[3][2][4], fw[3][4];
    auto load = [&](int set, int c) {
        const long k = (long)c * 64;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            fw[set][s] = *reinterpret_cast<const bf16x8*>(w0 + k + s * 8);
            fa[set][0][s] = *reinterpret_cast<const bf16x8*>(a0 + k + s * 8);
            fa[set][1][s] = *reinterpret_cast<const bf16x8*>(a1 + k + s * 8);
        }
    };
    auto mma = [&](int set) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < 2; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[set][s], fa[set][i][s], acc[i], 0, 0, 0);
    };
    // three chunks of this wave in flight (the kernel is latency-bound: 24-96 workgroups stream all of W);
    // the trip is unrolled by 3 so that the register sets rotate without dynamic indexing
    constexpr int ST = SK_WAVES;
    int c = wave;
    if (c < nchunk) load(0, c);
    if (c + ST < nchunk) load(1, c + ST);
    for (; c < nchunk; c += 3 * ST) {
        if (c + 2 * ST < nchunk) load(2, c + 2 * ST);
        mma(0);
        if (c + ST < nchunk) {
            if (c + 3 * ST < nchunk) load(0, c + 3 * ST);
            mma(1);
        }
        if (c + 2 * ST < nchunk) {
            if (c + 4 * ST < nchunk) load(1, c + 4 * ST);
            mma(2);
        }
    }
    // partial blocks -> LDS ([wave][row][col], swapped accumulator layout: lane = row, registers = 4-column runs)
    float* mine = part + wave * 64 * SK_LD;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<float4*>(mine + (i * 32 + l31) * SK_LD + 8 * g + 4 * hi) =
                make_float4(acc[i][4 * g], acc[i][4 * g + 1], acc[i][4 * g + 2], acc[i][4 * g + 3]);
    __syncthreads();
    // thread t: row t / 8, columns (t % 8) * 4 .. + 3; fixed summation order wave 0..7
    const int row = tid >> 3, cc = (tid & 7) * 4;
    float4 x = *reinterpret_cast<const float4*>(part + row * SK_LD + cc);
#pragma unroll
    for (int w = 1; w < SK_WAVES; ++w) {
        const float4 p = *reinterpret_cast<const float4*>(part + (w * 64 + row) * SK_LD + cc);
        x.x += p.x; x.y += p.y; x.z += p.z; x.w += p.w;
    }
    const int n = n0 + cc;
    if (row >= M || n >= N) return;            // N % 4 == 0
    long orow = row;
    if (ep.row_gin > 0) orow = (long)(row / ep.row_gin) * ep.row_gout + (row % ep.row_gin) + ep.row_off;
    const int rrow = (ep.res_rows > 0 && ep.res_rows < M) ? row % ep.res_rows : row;
    if (ep.bias) {
        const float4 b = *reinterpret_cast<const float4*>(ep.bias + n);
        x.x += b.x; x.y += b.y; x.z += b.z; x.w += b.w;
    }
    if (ACT == 1) apply_gelu4<TOUT>(x);
    if (ep.gamma) {
        const float4 g = *reinterpret_cast<const float4*>(ep.gamma + n);
        x.x *= g.x; x.y *= g.y; x.z *= g.z; x.w *= g.w;
    }
    if (ep.residual) {
        const float4 r = *reinterpret_cast<const float4*>(ep.residual + (long)rrow * ep.ldr + n);
        x.x += r.x; x.y += r.y; x.z += r.z; x.w += r.w;
    }
    store4_out<TOUT>(C + orow * ldc + n, x.x, x.y, x.z, x.w);
}

// ------------------------------------------------------------------------------------------------
// TN kernel for weight gradients: C[n, j] = sum_m X[m, n] * Y[m, j]  (X = dY [M, N], Y = A [M, Kc], both token-major, as
// the forward / dgrad GEMMs leave them).  The contraction index m is the ROW index of both operands, so the MFMA
// fragments (8 consecutive m for one n) are columns of the LDS tile: they are fetched with gfx950's transposing LDS read
// (ds_read_b64_tr_b16: within a 16-lane group lane i supplies 8 bytes of row i >> 2, and lane c receives column c of
// the resulting 4 x 16 block -- measured with tools/tr_lab), two reads per fragment.  This removes the two transpose
// passes per Linear that the NN formulation needed (10 % of the training step).
//   tile 128 (n) x 128 (j), 64 tokens per stage; LDS stage = X[64][128] + Y[64][128] bf16 (256-byte rows), two stages;
//   an LDS-DMA piece = 4 rows x 256 B; slot of (row r, 16-byte chunk c) = c ^ 4 (r & 3): the four rows of a transposed
//   read and the two column blocks of a 32-lane access land in 8 disjoint bank groups;
//   grid.y = split-K slices over the tokens (partials summed by m324_colsum, deterministic); rows past the slice end are
//   zeroed in LDS (X only) before they are contracted.
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;
typedef __attribute__((address_space(3))) s16x4_t lds_s16x4_t;

__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(const bf16_t* __restrict__ X, long ldx, const bf16_t* __restrict__ Y,
                                                         long ldy, float* __restrict__ C, long ldc, int M, int N, int Kc, int ks,
                                                         long strideC, int ntj) {
    constexpr int TROW = 256;                    // bytes per LDS row (128 bf16)
    constexpr int TOP = 64 * TROW;               // one operand of a stage: 16 KiB
    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * 2 * TOP];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int n0 = (blockIdx.x / ntj) * 128, j0 = (blockIdx.x % ntj) * 128;
    const int mbeg = blockIdx.y * ks;
    const int mend = blockIdx.y == gridDim.y - 1 ? M : mbeg + ks;
    C += (long)blockIdx.y * strideC;

    // staging: piece p = rows 4p .. 4p+3; wave w moves pieces 4w .. 4w+3 of X and of Y
    const int sr = lane >> 4, sc = (lane & 15) ^ (4 * sr);       // row within the piece, source chunk of this lane's slot
    const bf16_t* gx = X + min(n0 + sc * 8, N - 8);
    const bf16_t* gy = Y + min(j0 + sc * 8, Kc - 8);
    auto issue = [&](int t, int stage) {
        unsigned char* st = smem + stage * 2 * TOP + wave * 4096;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const long row = min(mbeg + t * 64 + (wave * 4 + p) * 4 + sr, mend - 1);
            __builtin_amdgcn_global_load_lds((glb_ptr_t*)(gx + row * ldx), (lds_ptr_t*)(st + p * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_ptr_t*)(gy + row * ldy), (lds_ptr_t*)(st + TOP + p * 1024), 16, 0, 0);
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // transposed fragment reads: lane (group g = lane >> 4, i = lane & 15) supplies row (i >> 2) [+ 8 (g >> 1) + 4 rd + 16 ks]
    // and columns base + 16 (g & 1) + 4 (i & 3) .. + 3 of the wave's 32-column block
    const int g = lane >> 4, li = lane & 15;
    const int frow = (g >> 1) * 8 + (li >> 2);
    auto foff = [&](int colbase) {               // byte offset inside an operand tile of this lane's piece (rd = ks = 0)
        const int col = colbase + 16 * (g & 1) + 4 * (li & 3);
        return frow * TROW + (((col >> 3) ^ (4 * (li >> 2))) << 4) + ((col & 7) << 1);
    };
    int xo[2], yo[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) xo[b] = foff(wm * 64 + b * 32), yo[b] = TOP + foff(wn * 64 + b * 32);
    auto frag = [&](const unsigned char* st, int off, int ks16) {
        const unsigned char* p = st + off + ks16 * (16 * TROW);
        const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)p);
        const s16x4_t hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(p + 4 * TROW));
        const s16x8_t v = __builtin_shufflevector(lo, hi4, 0, 1, 2, 3, 4, 5, 6, 7);
        return __builtin_bit_cast(bf16x8, v);
    };

    const int nt = (mend - mbeg + 63) / 64;
    issue(0, 0);
    for (int t = 0; t < nt; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // tile t landed
        __syncthreads();
        if (t + 1 < nt) issue(t + 1, (t + 1) & 1);
        unsigned char* st = smem + (t & 1) * 2 * TOP;
        const int valid = mend - mbeg - t * 64;                 // rows of this tile inside the slice
        if (valid < 64) {                                       // last tile of the slice: zero the X rows past its end
            for (int e = tid; e < (64 - valid) * 16; e += 256)
                *reinterpret_cast<uint4*>(st + (valid + (e >> 4)) * TROW + ((e & 15) << 4)) = make_uint4(0, 0, 0, 0);
            __syncthreads();
        }
#pragma unroll
        for (int k16 = 0; k16 < 4; ++k16) {
            bf16x8 xf[2], yf[2];
#pragma unroll
            for (int b = 0; b < 2; ++b) xf[b] = frag(st, xo[b], k16), yf[b] = frag(st, yo[b], k16);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(yf[j], xf[i], acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();
    Epilogue ep{};
    store_tile_lds<float, 0, 0, 2>(acc, reinterpret_cast<float*>(smem) + wave * EP_WAVE_FLOATS, C, ldc, N, Kc, n0 + wm * 64,
                                   j0 + wn * 64, lane, ep);
}

// TN kernel, 256 x 256 tile on the v7 pipeline (used when the slices are whole 32-token half-tiles and the output is at
// least one 256 x 256 tile): half the LDS-DMA pieces per FLOP of the 128 x 128 kernel above, which is texture-path-bound.
// Ring slot = X[32 m][256 n] + Y[32 m][256 j] (512-byte rows, 16 KiB each); an LDS-DMA piece = 2 rows x 512 B; slot of
// (row r, 16-byte chunk c) = c ^ 4 (r & 3).  Per k-step a wave reads 6 fragments = 12 transposing reads for 8 MFMAs.
__global__ __launch_bounds__(512, 2) void gemm_tn_pipe_kernel(const bf16_t* __restrict__ X, long ldx, const bf16_t* __restrict__ Y,
                                                              long ldy, float* __restrict__ C, long ldc, int M, int N, int Kc,
                                                              int ks, long strideC, int ntj) {
    constexpr int TROW = 512, TOP = 32 * TROW;   // bytes per LDS row; one operand of a slot (16 KiB)
    constexpr int RING = 4, PPW = 4;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[RING * 2 * TOP];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int n0 = (blockIdx.x / ntj) * 256, j0 = (blockIdx.x % ntj) * 256;
    const int mbeg = blockIdx.y * ks;
    const int mend = blockIdx.y == gridDim.y - 1 ? M : mbeg + ks;      // (mend - mbeg) % 32 == 0 (host-checked)
    C += (long)blockIdx.y * strideC;

    // staging: piece p = rows 2p, 2p+1 of the half-tile; wave w moves pieces 2w, 2w+1 (rows 4w .. 4w+3) of X and of Y
    const int sr = lane >> 5;                    // row within the piece
    const bf16_t* gx[2];
    const bf16_t* gy[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int r = (wave * 2 + p) * 2 + sr;   // row within the half-tile
        const int c = (lane & 31) ^ (4 * (r & 3));
        gx[p] = X + (long)(mbeg + r) * ldx + min(n0 + c * 8, N - 8);
        gy[p] = Y + (long)(mbeg + r) * ldy + min(j0 + c * 8, Kc - 8);
    }
    auto issue_x = [&](int h, int slot) {
        unsigned char* st = smem + slot * 2 * TOP + wave * 2048;
#pragma unroll
        for (int p = 0; p < 2; ++p)
            __builtin_amdgcn_global_load_lds((glb_ptr_t*)(gx[p] + (long)h * 32 * ldx), (lds_ptr_t*)(st + p * 1024), 16, 0, 0);
    };
    auto issue_y = [&](int h, int slot) {
        unsigned char* st = smem + slot * 2 * TOP + TOP + wave * 2048;
#pragma unroll
        for (int p = 0; p < 2; ++p)
            __builtin_amdgcn_global_load_lds((glb_ptr_t*)(gy[p] + (long)h * 32 * ldy), (lds_ptr_t*)(st + p * 1024), 16, 0, 0);
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int g = lane >> 4, li = lane & 15;
    const int frow = (g >> 1) * 8 + (li >> 2);
    auto foff = [&](int colbase) {
        const int col = colbase + 16 * (g & 1) + 4 * (li & 3);
        return frow * TROW + (((col >> 3) ^ (4 * (li >> 2))) << 4) + ((col & 7) << 1);
    };
    int xo[4], yo[2];
#pragma unroll
    for (int b = 0; b < 4; ++b) xo[b] = foff(wm * 128 + b * 32);
#pragma unroll
    for (int b = 0; b < 2; ++b) yo[b] = TOP + foff(wn * 64 + b * 32);
    bf16x8 fx[2][4], fy[2][2];
    auto frag = [&](const unsigned char* p) {
        const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)p);
        const s16x4_t hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(p + 4 * TROW));
        const s16x8_t v = __builtin_shufflevector(lo, hi4, 0, 1, 2, 3, 4, 5, 6, 7);
        return __builtin_bit_cast(bf16x8, v);
    };
    auto load_frags = [&](int set, int slot, int k16) {
        const unsigned char* base = smem + slot * 2 * TOP + k16 * (16 * TROW);
#pragma unroll
        for (int b = 0; b < 2; ++b) fy[set][b] = frag(base + yo[b]);
#pragma unroll
        for (int b = 0; b < 4; ++b) fx[set][b] = frag(base + xo[b]);
    };
    auto mma8 = [&](int set) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fy[set][j], fx[set][i], acc[i][j], 0, 0, 0);
    };
#define M324_SG(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
    auto sched_kstep = [&]() {                   // 12 transposing reads on the first 6 MFMAs, 2 LDS-DMA pieces on the last two
        M324_SG(0x008, 1); M324_SG(0x100, 2); M324_SG(0x008, 1); M324_SG(0x100, 2);
        M324_SG(0x008, 1); M324_SG(0x100, 2); M324_SG(0x008, 1); M324_SG(0x100, 2);
        M324_SG(0x008, 1); M324_SG(0x100, 2); M324_SG(0x008, 1); M324_SG(0x100, 2);
        M324_SG(0x008, 1); M324_SG(0x020, 1); M324_SG(0x008, 1); M324_SG(0x020, 1);
    };
#define M324_WAIT_PIECES(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
    const int NH = (mend - mbeg) / 32;
    for (int h = 0; h < RING - 1; ++h) {
        issue_x(h < NH ? h : NH - 1, h);
        issue_y(h < NH ? h : NH - 1, h);
    }
    M324_WAIT_PIECES(2 * PPW);
    M324_BARRIER();
    load_frags(0, 0, 0);
    int slot = 0;
    for (int h = 0; h < NH; ++h) {
        const int nslot = (slot + 1) & 3, fslot = (slot + 3) & 3;
        const int hn = h + 3 < NH ? h + 3 : NH - 1;
        M324_WAIT_PIECES(PPW);
        M324_BARRIER();
        load_frags(1, slot, 1);
        issue_x(hn, fslot);
        mma8(0);
        sched_kstep();
        load_frags(0, nslot, 0);
        issue_y(hn, fslot);
        mma8(1);
        sched_kstep();
        slot = nslot;
    }
    M324_WAIT_PIECES(0);
#undef M324_WAIT_PIECES
#undef M324_SG
    M324_BARRIER();
    Epilogue ep{};
    store_tile_lds<float, 0, 0, 4>(acc, reinterpret_cast<float*>(smem) + wave * EP_WAVE_FLOATS, C, ldc, N, Kc, n0 + wm * 128,
                                   j0 + wn * 64, lane, ep);
}

// the vectorised epilogue of the LDS-DMA kernel needs 4-column runs to be addressable as float4 / uint2
static bool vec_ok(const m324_gemm_args* a) {
    const int osz = a->out_dtype == M324_BF16 ? 2 : 4;
    auto al = [](const void* p, int b) { return ((uintptr_t)p % b) == 0; };
    return a->N % 4 == 0 && a->N >= 4 && (a->ldc * osz) % (4 * osz) == 0 && al(a->C, 4 * osz) &&
           (!a->residual || (a->ldr % 4 == 0 && al(a->residual, 16))) && (!a->bias || al(a->bias, 16)) &&
           (!a->gamma || al(a->gamma, 16));
}

// XCD-aware tile order (default on; M324_XCD=0 disables)
static int xcd_remap() {
    static const int v = [] { const char* e = getenv("M324_XCD"); return e ? (atoi(e) & 1) : 1; }();
    return v;
}

// Kernel choice.  M324_GEMM=v1|v2|v5|v7|v9 forces a variant (A/B measurements, tests).
static int forced_variant() {      // read per call: lets one process A/B-toggle the variant
    const char* e = getenv("M324_GEMM");
    return (e && e[0] == 'v') ? atoi(e + 1) : 0;
}

static int pick_variant(const m324_gemm_args* a) {
    if (!vec_ok(a)) return 1;
    const int f = forced_variant();
    const bool bf16 = a->in_dtype == M324_BF16;
    if (f == 1 || f == 2 || f == 5) return f;
    if (f == 7) return bf16 ? 7 : 5;
    if (a->M <= 64 && bf16 && !a->aux_mode && (f == 0 || f == 9)) return 9;
    // 256 x 256 tiles halve the LDS-DMA traffic per FLOP: fastest whenever the column count quantises (N % 256 == 0)
    // and the tiles fill most of the 256 CUs in whole rounds; otherwise the 128 x 128 tiles of v2 (two workgroups per
    // CU) balance better.  Measured on the c2 shapes (tools/gemm_lab): v7 wins at >= 0.70 fill, v2 below.
    const long t5 = (long)ceil_div(a->N, BN5) * ceil_div(a->M, BM5);
    const double e5 = (double)t5 / (double)(((t5 + 255) / 256) * 256);
    if (a->N % BN5 == 0 && t5 >= 200 && e5 >= 0.70) return (bf16 && a->K >= 96) ? 7 : 5;
    return 2;
}

template <typename TOUT, int ACT, int RES>
static void launch_pipe(const m324_gemm_args* a, hipStream_t s, const Epilogue& ep) {
    hipLaunchKernelGGL((gemm_pipe_kernel<TOUT, ACT, RES>), dim3(ceil_div(a->N, BN5) * ceil_div(a->M, BM5)), dim3(512), 0, s,
                       (const bf16_t*)a->A, a->lda, (const bf16_t*)a->W, a->ldw, (TOUT*)a->C, a->ldc, a->M, a->N, a->K, ep,
                       ceil_div(a->N, BN5), xcd_remap());
}

template <typename TIN, typename TOUT>
int launch(const m324_gemm_args* a, hipStream_t s) {
    Epilogue ep{a->bias, a->gamma, a->residual, a->ldr, a->res_rows, a->act, a->row_gin, a->row_gout, a->row_off,
                a->strideA, a->strideW, a->strideC, a->aux, a->ldaux, a->aux_mode,
                {(bf16_t*)a->qkv_q, (bf16_t*)a->qkv_k, (bf16_t*)a->qkv_v}, {a->qkv_qw, a->qkv_kw}, a->qkv_eps, a->qkv_qscale,
                a->qkv_L, a->qkv_H};
    dim3 grid(ceil_div(a->N, BN), ceil_div(a->M, BM));
    const int nbatch = a->batch > 1 ? a->batch : 1;
    const int variant = nbatch > 1 ? 2 : pick_variant(a);
    if (variant == 1) {
        hipLaunchKernelGGL((gemm_kernel<TIN, TOUT>), grid, dim3(256), 0, s, (const TIN*)a->A, a->lda, (const TIN*)a->W,
                           a->ldw, (TOUT*)a->C, a->ldc, a->M, a->N, a->K, ep);
    } else if (variant == 9) {
        if (a->act == M324_ACT_GELU)
            hipLaunchKernelGGL((gemm_skinny_kernel<TOUT, 1>), dim3(ceil_div(a->N, 32)), dim3(512), 0, s, (const bf16_t*)a->A,
                               a->lda, (const bf16_t*)a->W, a->ldw, (TOUT*)a->C, a->ldc, a->M, a->N, a->K, ep);
        else
            hipLaunchKernelGGL((gemm_skinny_kernel<TOUT, 0>), dim3(ceil_div(a->N, 32)), dim3(512), 0, s, (const bf16_t*)a->A,
                               a->lda, (const bf16_t*)a->W, a->ldw, (TOUT*)a->C, a->ldc, a->M, a->N, a->K, ep);
    } else {
        const int res = !a->residual && a->row_gin <= 0 ? 0
                        : (a->residual && a->row_gin <= 0 && (a->res_rows <= 0 || a->res_rows >= a->M)) ? 1 : 2;
#define M324_GLDS(ACT, RES)                                                                                              \
    do {                                                                                                                 \
        if (variant == 7)                                                                                                \
            launch_pipe<TOUT, ACT, RES>(a, s, ep);                                                                       \
        else if (variant == 5)                                                                                           \
            hipLaunchKernelGGL((gemm_glds5_kernel<TIN, TOUT, ACT, RES>),                                                 \
                               dim3(ceil_div(a->N, BN5) * ceil_div(a->M, BM5)), dim3(512), 0, s, (const TIN*)a->A,       \
                               a->lda, (const TIN*)a->W, a->ldw, (TOUT*)a->C, a->ldc, a->M, a->N, a->K, ep,              \
                               ceil_div(a->N, BN5), xcd_remap());                                                        \
        else                                                                                                             \
            hipLaunchKernelGGL((gemm_glds_kernel<TIN, TOUT, ACT, RES>), dim3(grid.x * grid.y, nbatch), dim3(256), 0, s,  \
                               (const TIN*)a->A, a->lda, (const TIN*)a->W, a->ldw, (TOUT*)a->C, a->ldc, a->M, a->N,      \
                               a->K, ep, (int)grid.x, xcd_remap());                                                      \
    } while (0)
        if (a->aux_mode == M324_AUX_QKV_HEADS) {
            if constexpr (sizeof(TOUT) == 2) M324_GLDS(4, 0);
        } else if (a->aux_mode == M324_AUX_STORE_PREACT) {
            M324_GLDS(2, 0);
        } else if (a->aux_mode == M324_AUX_MUL_GELU_GRAD) {
            M324_GLDS(3, 0);
        } else if (a->act == M324_ACT_GELU) {
            if (res == 0) M324_GLDS(1, 0); else M324_GLDS(1, 2);
        } else {
            if (res == 0) M324_GLDS(0, 0); else if (res == 1) M324_GLDS(0, 1); else M324_GLDS(0, 2);
        }
#undef M324_GLDS
    }
    M324_CHECK_LAUNCH("m324_gemm");
    return M324_OK;
}

}  // namespace

extern "C" int m324_gemm_tn(const void* X, long ldx, const void* Y, long ldy, float* C, long ldc, int M, int N, int Kc,
                            int slices, long strideC, void* stream) {
    M324_REQUIRE(X && Y && C, "m324_gemm_tn: null pointer");
    M324_REQUIRE(M > 0 && N >= 8 && Kc >= 8 && N % 8 == 0 && Kc % 8 == 0, "m324_gemm_tn: bad sizes M=%d N=%d Kc=%d", M, N, Kc);
    M324_REQUIRE(ldx >= N && ldy >= Kc && ldx % 8 == 0 && ldy % 8 == 0 && ldc >= Kc && ldc % 4 == 0 &&
                     ((uintptr_t)X % 16) == 0 && ((uintptr_t)Y % 16) == 0 && ((uintptr_t)C % 16) == 0,
                 "m324_gemm_tn: operands must be 16-byte aligned with leading dimensions that are multiples of 8");
    M324_REQUIRE(slices >= 1 && slices <= 65535 && (slices == 1 || strideC >= (long)N * ldc), "m324_gemm_tn: bad slicing");
    const int ks = ((M + slices - 1) / slices + 63) / 64 * 64;          // tokens per slice, whole 64-row tiles
    M324_REQUIRE((long)ks * (slices - 1) < M, "m324_gemm_tn: %d slices leave an empty slice for M=%d", slices, M);
    // 256 x 256 tiles on the pipelined kernel when every slice is whole 32-token half-tiles and the output fills the tiles
    // (M324_GEMM_TN=128 forces the small kernel, read per call)
    const char* ftn = getenv("M324_GEMM_TN");
    const bool big = !(ftn && atoi(ftn) == 128) && M % 32 == 0 && N % 256 == 0 && Kc % 256 == 0 && M / slices >= 256;
    if (big) {
        const int ntj = Kc / 256;
        hipLaunchKernelGGL(gemm_tn_pipe_kernel, dim3((N / 256) * ntj, slices), dim3(512), 0, (hipStream_t)stream,
                           (const bf16_t*)X, ldx, (const bf16_t*)Y, ldy, C, ldc, M, N, Kc, ks, strideC, ntj);
    } else {
        const int ntj = ceil_div(Kc, 128);
        hipLaunchKernelGGL(gemm_tn_kernel, dim3(ceil_div(N, 128) * ntj, slices), dim3(256), 0, (hipStream_t)stream,
                           (const bf16_t*)X, ldx, (const bf16_t*)Y, ldy, C, ldc, M, N, Kc, ks, strideC, ntj);
    }
    M324_CHECK_LAUNCH("m324_gemm_tn");
    return M324_OK;
}

extern "C" int m324_gemm(const m324_gemm_args* a, void* stream) {
    M324_REQUIRE(a && a->A && a->W && (a->C || a->aux_mode == M324_AUX_QKV_HEADS), "m324_gemm: null pointer");
    M324_REQUIRE(a->M > 0 && a->N > 0 && a->K > 0, "m324_gemm: empty problem M=%d N=%d K=%d", a->M, a->N, a->K);
    const int bk = a->in_dtype == M324_BF16 ? 64 : 32;
    M324_REQUIRE(a->K % bk == 0, "m324_gemm: K=%d must be a multiple of %d", a->K, bk);
    const int esz = a->in_dtype == M324_BF16 ? 2 : 4;
    M324_REQUIRE((a->lda * esz) % 16 == 0 && (a->ldw * esz) % 16 == 0 && ((uintptr_t)a->A % 16) == 0 &&
                     ((uintptr_t)a->W % 16) == 0,
                 "m324_gemm: A/W rows must be 16-byte aligned");
    M324_REQUIRE(a->lda >= a->K && a->ldw >= a->K && a->ldc >= a->N, "m324_gemm: leading dimension too small");
    M324_REQUIRE(!a->residual || a->ldr >= a->N, "m324_gemm: ldr too small");
    M324_REQUIRE(a->batch <= 1 || (vec_ok(a) && !a->residual && a->batch <= 65535),
                 "m324_gemm: a batched launch needs a vectorisable, residual-free problem");
    M324_REQUIRE(a->aux_mode >= 0 && a->aux_mode <= 3, "m324_gemm: aux_mode %d", a->aux_mode);
    if (a->aux_mode == M324_AUX_QKV_HEADS) {
        M324_REQUIRE(a->in_dtype == M324_BF16 && a->out_dtype == M324_BF16 && a->qkv_q && a->qkv_k && a->qkv_v && a->qkv_H > 0 &&
                         a->qkv_L > 0 && a->N == 3 * a->qkv_H * 64 && a->M % a->qkv_L == 0 && !a->residual && !a->gamma &&
                         a->act == M324_ACT_NONE && a->row_gin <= 0 && a->batch <= 1 && a->M > 64 && vec_ok(a),
                     "m324_gemm: M324_AUX_QKV_HEADS needs a plain bf16 [B*L, 3*H*64] projection with M > 64");
        M324_REQUIRE(((uintptr_t)a->qkv_q % 16) == 0 && ((uintptr_t)a->qkv_k % 16) == 0 && ((uintptr_t)a->qkv_v % 16) == 0 &&
                         (!a->qkv_qw || ((uintptr_t)a->qkv_qw % 16) == 0) && (!a->qkv_kw || ((uintptr_t)a->qkv_kw % 16) == 0),
                     "m324_gemm: misaligned qkv outputs / norm weights");
    } else if (a->aux_mode) {
        const int osz = a->out_dtype == M324_BF16 ? 2 : 4;
        M324_REQUIRE(a->aux && a->ldaux >= a->N && a->ldaux % 4 == 0 && ((uintptr_t)a->aux % (4 * osz)) == 0 && vec_ok(a) &&
                         a->row_gin <= 0 && a->batch <= 1,
                     "m324_gemm: aux operand needs a vectorisable, un-remapped, un-batched problem with ldaux >= N");
        M324_REQUIRE(a->aux_mode != M324_AUX_STORE_PREACT || a->act == M324_ACT_GELU,
                     "m324_gemm: M324_AUX_STORE_PREACT only makes sense with an activation");
        M324_REQUIRE(a->aux_mode != M324_AUX_MUL_GELU_GRAD || a->act == M324_ACT_NONE,
                     "m324_gemm: M324_AUX_MUL_GELU_GRAD excludes an activation");
        M324_REQUIRE(!a->residual, "m324_gemm: the aux operand excludes a residual");
    }
    hipStream_t s = (hipStream_t)stream;
    if (a->in_dtype == M324_BF16 && a->out_dtype == M324_BF16) return launch<bf16_t, bf16_t>(a, s);
    if (a->in_dtype == M324_BF16 && a->out_dtype == M324_F32) return launch<bf16_t, float>(a, s);
    if (a->in_dtype == M324_F32 && a->out_dtype == M324_F32) return launch<float, float>(a, s);
    M324_FAIL(M324_ERR_UNSUPPORTED, "m324_gemm: unsupported dtype pair in=%d out=%d", a->in_dtype, a->out_dtype);
}
