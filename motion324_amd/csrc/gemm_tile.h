// Shared by the GEMM translation units (gemm.hip, gemm_ring4.hip, gemm_pp.hip): epilogue descriptor, GELU forms, LDS swizzles, the
// MFMA k-tile of the 128 x 128 kernels and the LDS-transposed epilogue every LDS-DMA kernel ends with.
#pragma once
#include <utility>
#include <stdlib.h>
#include <type_traits>
#include "common.h"

namespace m324 {
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
    int qkv_vt;                  // M324_AUX_QKV_HEADS_VT: qkv_out[2] is Vt [B, H, 64, L] (key quarters of every 16 in the order 0,2,1,3)
    int stream;                  // bf16 outputs without residual: store non-temporal (host: output larger than the MALL keeps)
    int res_out;                 // the residual has the OUTPUT's dtype (host: residual aliases a bf16 C -- a bf16 residual stream updated in place)
    // LayerNorm folded around the GEMM (bf16 inference; m324.h "LayerNorm fold").  Consumer side: A holds the raw stream and W
    // the weight with the LayerNorm's scale folded in; the epilogue turns acc into rstd (acc - mean colsum[n]).  Producer
    // side: the epilogue leaves per-row (sum, M2) of every 64-column block of the values it stores, and (fp32 outputs) a
    // bf16 copy of them -- the next GEMM's A operand.
    const float2* rowstat;       // [M] (rstd, -rstd * mean) of the A rows, or null
    const float* colsum;         // [N] sum over k of W[n, k]
    float2* stats;               // [N / 64][M] (sum, sum of squared deviations from the block mean), or null
    bf16_t* copy;                // [M, ldcopy] bf16 copy of the stored values (fp32 outputs), or null
    long ldcopy;
    int ncb;                     // > 0: rowstat is the producer's UNMERGED table [ncb][M] (sum, M2 per 64-column block); the consumer merges its rows' blocks
    float eps;                   // ... with this LayerNorm epsilon (what m324_rowstats_finish does in a launch of its own)
};

// gemm_ring4.hip (own translation unit: built WITHOUT -amdgpu-mfma-vgpr-form, its 256 accumulators live in AGPRs).
// act_code: the ACT template value (0 none, 1 GELU, 2 GELU + pre-activation, 3 x gelu', 4 q|k|v heads); res_code: RES.
int launch_ring4(const m324_gemm_args* a, hipStream_t s, const Epilogue& ep, int act_code, int res_code, int xcd_remap, int variant);
// gemm_hp.hip (v15: the hand-placed K = 768 stream, previous tile's epilogue inside the main loop); hp_ok: does it take this GEMM
bool hp_ok(const m324_gemm_args* a, const Epilogue& ep, int act_code, int res_code);
int launch_hp(const m324_gemm_args* a, hipStream_t s, const Epilogue& ep, int act_code, int res_code, int xcd_remap);
int hp_grid(const m324_gemm_args* a);
}  // namespace m324

namespace {


constexpr int BM = 128, BN = 128, ROWB = 128;           // ROWB: bytes of K per LDS row
constexpr int TILE_BYTES = BM * ROWB;                   // 16 KiB per operand per stage

using m324::Epilogue;

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

// Two GELUs per instruction stream for the bf16 epilogue: GELU(x) = x (1/2 + u Q(u^2)), u = clamp(x, +-3 sqrt 2), with
// u Q(u^2) = erf(u / sqrt 2) / 2 as an odd minimax polynomial (9 coefficients, |erf error| <= 1.7e-5; beyond the clamp
// erf(3) = 0.99998 stands in for 1), evaluated with packed fp32 FMAs (v_pk_fma_f32) and no transcendental: 13 VALU
// instructions per pair.  |GELU error| <= 6e-5 absolute -- below the bf16 rounding step of every output larger than 0.02.
typedef float f32x2v __attribute__((ext_vector_type(2)));
#define M324_GELU_Q8 5.626603458e-11f
#define M324_GELU_Q7 -5.371752709e-09f
#define M324_GELU_Q6 2.268262506e-07f
#define M324_GELU_Q5 -5.646163474e-06f
#define M324_GELU_Q4 9.359017959e-05f
#define M324_GELU_Q3 -1.109398132e-03f
#define M324_GELU_Q2 9.818113584e-03f
#define M324_GELU_Q1 -6.634691738e-02f
#define M324_GELU_Q0 3.989031257e-01f
#define M324_GELU_CLAMP 4.2426405f
// N independent pairs, one Horner step of every pair before the next step of any: a dependent v_pk_fma_f32 needs a wait
// state, and the compiler otherwise emits the pairs one after another with an s_nop between all 11 dependent steps.
// GRAD: d receives gelu'(x) = Phi(x) + x phi(x) as well -- Phi is the polynomial's own 1/2 + u Q(u^2), phi(x) = exp2(-x^2 log2(e) / 2) /
// sqrt(2 pi): four more instructions per value, one of them transcendental (M324_AUX_STORE_GELU_GRAD: the training forward leaves the
// derivative instead of the pre-activation, so that the dgrad epilogue of the Linear behind the GELU multiplies instead of evaluating erf + exp).
template <int N, bool GRAD = false>
__device__ __forceinline__ void gelu_poly2n(f32x2v (&x)[N], f32x2v* d = nullptr) {
    f32x2v u[N], t[N], p[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        u[i].x = __builtin_amdgcn_fmed3f(x[i].x, -M324_GELU_CLAMP, M324_GELU_CLAMP);
        u[i].y = __builtin_amdgcn_fmed3f(x[i].y, -M324_GELU_CLAMP, M324_GELU_CLAMP);
    }
#pragma unroll
    for (int i = 0; i < N; ++i) t[i] = u[i] * u[i];
#pragma unroll
    for (int i = 0; i < N; ++i) p[i] = __builtin_elementwise_fma((f32x2v)(M324_GELU_Q8), t[i], (f32x2v)(M324_GELU_Q7));
#define M324_GELU_STEP(C)          \
    _Pragma("unroll") for (int i = 0; i < N; ++i) p[i] = __builtin_elementwise_fma(p[i], t[i], (f32x2v)(C));
    M324_GELU_STEP(M324_GELU_Q6)
    M324_GELU_STEP(M324_GELU_Q5)
    M324_GELU_STEP(M324_GELU_Q4)
    M324_GELU_STEP(M324_GELU_Q3)
    M324_GELU_STEP(M324_GELU_Q2)
    M324_GELU_STEP(M324_GELU_Q1)
    M324_GELU_STEP(M324_GELU_Q0)
#undef M324_GELU_STEP
#pragma unroll
    for (int i = 0; i < N; ++i) p[i] = __builtin_elementwise_fma(u[i], p[i], (f32x2v)(0.5f));
    if constexpr (GRAD) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const f32x2v xx = x[i] * x[i], xs = x[i] * (f32x2v)(0.39894228040143267794f);
            f32x2v e;
            e.x = __builtin_amdgcn_exp2f(-0.72134752044448170368f * xx.x);
            e.y = __builtin_amdgcn_exp2f(-0.72134752044448170368f * xx.y);
            d[i] = __builtin_elementwise_fma(xs, e, p[i]);
        }
    }
#pragma unroll
    for (int i = 0; i < N; ++i) x[i] = x[i] * p[i];
}
__device__ __forceinline__ f32x2v gelu_poly2(f32x2v x) {
    f32x2v v[1] = {x};
    gelu_poly2n<1>(v);
    return v[0];
}

template <typename TOUT>
__device__ __forceinline__ void apply_gelu4(float4& v) {
    if constexpr (sizeof(TOUT) == 2) {
        f32x2v a[2] = {{v.x, v.y}, {v.z, v.w}};
        gelu_poly2n<2>(a);
        v = make_float4(a[0].x, a[0].y, a[1].x, a[1].y);
    } else {
        v = make_float4(gelu_erf(v.x), gelu_erf(v.y), gelu_erf(v.z), gelu_erf(v.w));
    }
}
template <typename TOUT>
__device__ __forceinline__ void apply_gelu8(float4& v, float4& w) {
    if constexpr (sizeof(TOUT) == 2) {
        f32x2v a[4] = {{v.x, v.y}, {v.z, v.w}, {w.x, w.y}, {w.z, w.w}};
        gelu_poly2n<4>(a);
        v = make_float4(a[0].x, a[0].y, a[1].x, a[1].y);
        w = make_float4(a[2].x, a[2].y, a[3].x, a[3].y);
    } else {
        apply_gelu4<TOUT>(v);
        apply_gelu4<TOUT>(w);
    }
}

// v <- gelu(v), d <- gelu'(v) (both from the fp32 value in front of the activation)
template <typename TOUT>
__device__ __forceinline__ void apply_gelu4_grad(float4& v, float4& d) {
    if constexpr (sizeof(TOUT) == 2) {
        f32x2v a[2] = {{v.x, v.y}, {v.z, v.w}}, g[2];
        gelu_poly2n<2, true>(a, g);
        v = make_float4(a[0].x, a[0].y, a[1].x, a[1].y);
        d = make_float4(g[0].x, g[0].y, g[1].x, g[1].y);
    } else {
        auto one = [](float z, float& dz) {
            const float cdf = 0.5f * (1.0f + erff(z * 0.70710678118654752440f));
            dz = cdf + z * 0.39894228040143267794f * expf(-0.5f * z * z);
            return z * cdf;
        };
        v = make_float4(one(v.x, d.x), one(v.y, d.y), one(v.z, d.z), one(v.w, d.w));
    }
}
template <typename TOUT>
__device__ __forceinline__ void apply_gelu8_grad(float4& v, float4& w, float4& dv, float4& dw) {
    if constexpr (sizeof(TOUT) == 2) {
        f32x2v a[4] = {{v.x, v.y}, {v.z, v.w}, {w.x, w.y}, {w.z, w.w}}, g[4];
        gelu_poly2n<4, true>(a, g);
        v = make_float4(a[0].x, a[0].y, a[1].x, a[1].y);
        w = make_float4(a[2].x, a[2].y, a[3].x, a[3].y);
        dv = make_float4(g[0].x, g[0].y, g[1].x, g[1].y);
        dw = make_float4(g[2].x, g[2].y, g[3].x, g[3].y);
    } else {
        apply_gelu4_grad<TOUT>(v, dv);
        apply_gelu4_grad<TOUT>(w, dw);
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

// Workgroup -> output tile.  Hardware deals consecutive workgroup ids round-robin over the 8 XCDs (id & 7), each with
// its own 4 MiB L2.  mode bit 0: XCD x works a CONTIGUOUS range of the tile order, so an A row panel is pulled into one
// L2 and shared by its ntn column tiles instead of being fetched by all eight.  mode bit 1 (weights larger than an L2):
// the tile order itself becomes 4 row groups x 2 column groups, one group per XCD -- every XCD then keeps HALF of W
// resident and streams a quarter of A, instead of cycling all of W through its L2 once per row panel.
__device__ __forceinline__ void tile_of(int bid, int nblocks, int ntm, int ntn, int mode, int& tm, int& tn) {
    int p = bid;
    tm = tn = 0;
    if (mode & 1) {
        const int q = nblocks >> 3, r = nblocks & 7, x = bid & 7, loc = bid >> 3;
        p = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + loc;
    }
    if (!(mode & 2)) {
        tm = p / ntn;
        tn = p - tm * ntn;
        return;
    }
    const int rq = ntm >> 2, rr = ntm & 3, c0 = (ntn + 1) >> 1;
    int rs = 0;
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
        const int rows = rq + (rg < rr);
#pragma unroll
        for (int cg = 0; cg < 2; ++cg) {
            const int cols = cg ? ntn - c0 : c0, size = rows * cols;
            if (p >= 0 && p < size) {
                const int rl = p / cols;
                tm = rs + rl;
                tn = (cg ? c0 : 0) + (p - rl * cols);
            }
            p -= size;                                   // p goes negative once its group is found
        }
        rs += rows;
    }
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
                    res[r] = ep.res_out ? Elem<TOUT>::load(reinterpret_cast<const TOUT*>(ep.residual) + (long)mc * ep.ldr + nc)
                                        : ep.residual[(long)mc * ep.ldr + nc];
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


// Sum over the aligned group of 8 (16) consecutive lanes, delivered to every lane of the group: v_add_f32 with a DPP operand
// (quad_perm butterflies, then row_half_mirror / row_mirror) -- no LDS crossbar traffic, unlike ds_bpermute shuffles.
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float group8_sum(float v) {
    v += dpp_f<0xB1>(v);         // quad_perm [1, 0, 3, 2]
    v += dpp_f<0x4E>(v);         // quad_perm [2, 3, 0, 1]
    v += dpp_f<0x141>(v);        // row_half_mirror: lane i <- lane 7 - i of its group of 8
    return v;
}
__device__ __forceinline__ float group16_sum(float v) {
    v = group8_sum(v);
    v += dpp_f<0x140>(v);        // row_mirror: lane i <- lane 15 - i
    return v;
}
// LayerNorm fold, consumer side: v = rstd * acc - rstd * mean * colsum + bias  (rs = (rstd, -rstd * mean))
__device__ __forceinline__ void ln_fold4(float4& v, const float2 rs, const float4& cs, const float4& bi) {
    v.x = fmaf(rs.x, v.x, fmaf(rs.y, cs.x, bi.x)); v.y = fmaf(rs.x, v.y, fmaf(rs.y, cs.y, bi.y));
    v.z = fmaf(rs.x, v.z, fmaf(rs.y, cs.z, bi.z)); v.w = fmaf(rs.x, v.w, fmaf(rs.y, cs.w, bi.w));
}
// producer side: (sum, M2) of the 64 columns of a row that `group` lanes hold (8 lanes x 8 values or 16 lanes x 4)
// P independent rows at once, one reduction step of every row before the next step of any: a DPP add needs wait states
// behind the VALU write of its source, and a wave that is alone on its SIMD (the 4-wave kernels) otherwise walks each
// row's chain of ~25 dependent instructions to the end before it starts the next row's.
template <int P, int CTRL>
__device__ __forceinline__ void dpp_step(float (&v)[P]) {
    float t[P];
#pragma unroll
    for (int p = 0; p < P; ++p) t[p] = dpp_f<CTRL>(v[p]);
#pragma unroll
    for (int p = 0; p < P; ++p) v[p] += t[p];
}
template <int P, bool WIDE>          // WIDE: 16 lanes per row (4 values each); else 8 lanes (8 values each)
__device__ __forceinline__ void group_sum_batch(float (&v)[P]) {
    dpp_step<P, 0xB1>(v);
    dpp_step<P, 0x4E>(v);
    dpp_step<P, 0x141>(v);
    if (WIDE) dpp_step<P, 0x140>(v);
}
// (sum, M2) of the 64 columns of P rows; x[p] (and y[p]) are this lane's 4 (8) values of row p
template <int P>
__device__ __forceinline__ void ln_stats4_batch(const float4 (&x)[P], float2 (&st)[P]) {
    float s[P], q[P];
#pragma unroll
    for (int p = 0; p < P; ++p) s[p] = (x[p].x + x[p].y) + (x[p].z + x[p].w);
    group_sum_batch<P, true>(s);
#pragma unroll
    for (int p = 0; p < P; ++p) {
        const float bm = s[p] * (1.0f / 64.0f);
        const float a = x[p].x - bm, b = x[p].y - bm, c = x[p].z - bm, d = x[p].w - bm;
        q[p] = fmaf(a, a, fmaf(b, b, fmaf(c, c, d * d)));
    }
    group_sum_batch<P, true>(q);
#pragma unroll
    for (int p = 0; p < P; ++p) st[p] = make_float2(s[p], q[p]);
}
template <int P>
__device__ __forceinline__ void ln_stats8_batch(const float4 (&x)[P], const float4 (&y)[P], float2 (&st)[P]) {
    float s[P], q[P];
#pragma unroll
    for (int p = 0; p < P; ++p) s[p] = ((x[p].x + x[p].y) + (x[p].z + x[p].w)) + ((y[p].x + y[p].y) + (y[p].z + y[p].w));
    group_sum_batch<P, false>(s);
#pragma unroll
    for (int p = 0; p < P; ++p) {
        const float bm = s[p] * (1.0f / 64.0f);
        const float a = x[p].x - bm, b = x[p].y - bm, c = x[p].z - bm, d = x[p].w - bm;
        const float e = y[p].x - bm, f = y[p].y - bm, g = y[p].z - bm, h = y[p].w - bm;
        q[p] = fmaf(a, a, fmaf(b, b, fmaf(c, c, d * d))) + fmaf(e, e, fmaf(f, f, fmaf(g, g, h * h)));
    }
    group_sum_batch<P, false>(q);
#pragma unroll
    for (int p = 0; p < P; ++p) st[p] = make_float2(s[p], q[p]);
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

// Large bf16 outputs without residual (the decoder's MLP hidden: 403 MB) are stored NON-TEMPORAL: such an output is a
// stream nobody finds in a cache again, and letting it allocate in the 4 MiB L2 of its XCD evicts the W panels every
// workgroup of the XCD re-reads (tools/gemm_lab, 65536 x 3072 x 768: 330 -> 291 us).  Outputs the 256 MiB MALL can
// keep for the consumer (the trunk's 64 MB hidden: the fc2 GEMM that reads it lost what fc1 gained), the fp32 residual
// stream (read-modify-write of the same lines) and the head-major q|k|v keep plain stores; Epilogue::stream is the
// host's decision (m324_gemm: more than 128 MiB of output).
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store16(void* p, bool stream, unsigned a, unsigned b, unsigned c, unsigned d) {
    if (stream) __builtin_nontemporal_store((u32x4v){a, b, c, d}, reinterpret_cast<u32x4v*>(p));
    else *reinterpret_cast<uint4*>(p) = make_uint4(a, b, c, d);
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
// LayerNorm-fold instantiations (ACTX & 8) append wave-private tables to the scratch: the 128 rows' (rstd, -rstd mean) for
// the consumer side -- fetched ONCE per tile, before the main loop where the kernel can, instead of a
// dependent 8-byte global load per 8 rows in the middle of the epilogue (measured: +7 us on a 60 us GEMM) -- and the
// colsum of its 64 columns.
constexpr int EP_LN_FLOATS = 512;
constexpr int ep_wave_floats(int actx) { return (actx & 8) ? EP_WAVE_FLOATS + EP_LN_FLOATS : EP_WAVE_FLOATS; }
// Consumer side: the table entry of one row per lane; kernels call this before their prologue's LDS-DMA (older than every piece,
// so the counted vmcnt waits of the main loop retire it for free).
// ep.ncb > 0: the table is the producer's unmerged one and the consumer merges the blocks of its rows between issuing its
// prologue's LDS-DMA and the main loop (ln_finish: the table loads are older than every piece, the counted wait the compiler puts
// in front of the merge leaves the pieces in flight).  Chan's update, block after block (blocks of 64 values):
//   mean += (mean_b - mean) / (b + 1),  M2 += M2_b + (mean_b - mean)^2 * 64 b / (b + 1).
// MI = 2 (128 x 128 tiles, 64 rows per wave): a lane fetches / merges its row alone (up to 16 table entries in flight).
// MI = 4 (v10: 128 rows per wave, the same rows under the four waves of a row of waves, 255 registers): wave wn fetches / merges
// rows 32 wn .. 32 wn + 31 (unmerged table: the lane halves take half of the blocks each and combine); the epilogue exchanges the
// 4 x 32 rows through one LDS table per row of waves (LnPreT::table) behind a workgroup barrier -- three registers held across
// the main loop instead of five.
constexpr int LN_MAX_NCB = 16;
typedef float ln_f32x16 __attribute__((ext_vector_type(16)));
typedef float ln_f32x8 __attribute__((ext_vector_type(8)));
template <int MI>
struct LnPreT {
    float2 rs;                   // (rstd, -rstd mean) of row mw + lane (MI = 2) / mw + slot + (lane & 31) (MI = 4)
    float cs;                    // colsum of column nw + lane
    // ncb > 0: the row's (half of the) block entries until ln_finish -- vector VALUES, not an array member: an array's
    // `#pragma unroll` loops are unrolled after the pass that turns arrays into registers, and the entries went through scratch
    std::conditional_t<(MI > 2), ln_f32x8, ln_f32x16> raw_sum, raw_m2;
    int table_off;               // MI = 4: the row of waves' shared (rstd, -rstd mean) table in LDS, in floats from the wave's own scratch
    int slot;                    // ... and this wave's first row in it
};
// one block more (cnt blocks merged so far); `take` false leaves the pair as it is (no branch: the lane halves of MI = 4 differ)
__device__ __forceinline__ void ln_chan_step(const float2 pb, float inv_cnt1, float w, bool take, float& mean, float& m2) {
    const float d = pb.x * (1.0f / 64.0f) - mean;                   // inv_cnt1 = 1 / (cnt + 1), w = 64 cnt / (cnt + 1)
    const float nm = fmaf(d, inv_cnt1, mean);
    const float n2 = m2 + fmaf(d * d, w, pb.y);
    mean = take ? nm : mean;
    m2 = take ? n2 : m2;
}
// The two halves of a row's blocks (h blocks = 64 h values each), merged one after the other by ln_chan_step, combined: Chan's
// pairwise form for equal counts, d^2 n n / 2n = 32 h d^2.  EVERY consumer merges a row this way -- the lane halves of the 256-wide
// kernels hold one half each, the 128-wide kernels and ln_row_direct walk both -- so that the statistics, and with them the
// product, do not depend on the schedule the chooser picked (a graph branch with half the rows may pick another one).
__device__ __forceinline__ float2 ln_combine_halves(float mean_a, float m2_a, float mean_b, float m2_b, int h, int ncb, float eps) {
    const float d = mean_b - mean_a;
    const float m2 = (m2_a + m2_b) + (d * d) * (32.0f * (float)h);
    const float mean = fmaf(0.5f, d, mean_a);
    const float rstd = rsqrtf(m2 / (64.0f * (float)ncb) + eps);
    return make_float2(rstd, -rstd * mean);
}
// entry (block, row) of the unmerged table: uniform base + uniform block offset + ONE 32-bit per-lane offset (64-bit per-lane
// pointers, one per block, were hoisted out of the tile loop and spilled).  Host: the table is smaller than 4 GiB.
__device__ __forceinline__ float2 ln_entry(const char* table, unsigned block_bytes, int block, unsigned lane_off) {
    return *reinterpret_cast<const float2*>(table + (size_t)((unsigned)block * block_bytes) + lane_off);
}
// MI = 4: `table` is the row of waves' LDS table, wn the wave's column (it owns rows 32 wn .. 32 wn + 31 of the 128).  Unmerged
// table: the lower lane half reads blocks 0 .. h - 1, the upper h .. ncb - 1, h = ncb / 2 (the host sends odd counts -- no width
// of the product -- through m324_rowstats_finish).
// IN_LOOP (MI = 4): the call sits inside v10's persistent tile loop (below); false: in front of it, where plain loads are safe and
// their latency runs beside the ring prologue's pieces.
template <int ACTX, int MI, bool IN_LOOP = true>
__device__ __forceinline__ void ln_prefetch(const Epilogue& ep, int M, int N, int mw, int nw, int lane, LnPreT<MI>& pre, int wn = 0,
                                            int table_off = 0) {
    pre.rs = make_float2(0.f, 0.f);
    pre.cs = 0.f;
    pre.table_off = table_off;
    pre.slot = 32 * wn;
    if constexpr (MI > 2 && IN_LOOP) asm volatile("" : "+v"(lane));      // tile-local: see the opaque copies below
    if constexpr ((ACTX & 8) != 0) {
        pre.cs = ep.colsum[min(nw + lane, N - 1)];
        const int row = MI > 2 ? min(mw + 32 * wn + (lane & 31), M - 1) : min(mw + lane, M - 1);
        if (ep.ncb <= 0) {
            pre.rs = ep.rowstat[row];
        } else if constexpr (MI > 2 && !IN_LOOP) {
            const char* const tbl = reinterpret_cast<const char*>(ep.rowstat);
            const int h = ep.ncb >> 1;
            const unsigned bb = (unsigned)M * 8u, off = (unsigned)((lane >> 5) ? h : 0) * bb + (unsigned)row * 8u;
#pragma unroll
            for (int i = 0; i < LN_MAX_NCB / 2; ++i) {
                const float2 p = ln_entry(tbl, bb, min(i, h - 1), off);
                pre.raw_sum[i] = p.x, pre.raw_m2[i] = p.y;
            }
        } else if constexpr (MI > 2) {
            // v10 calls this once per tile inside its persistent loop with 250 registers live.  Written as compiler-visible loads, the
            // (tile-invariant) per-block addresses were hoisted out of that loop into registers it does not have and came back
            // from scratch one by one; so: eight loads from ONE uniform base + 32-bit lane offsets and their wait in one block.
            // vmcnt(0): at both call sites nothing else this wave has in flight is wanted later than these entries.
            const char* tbl = reinterpret_cast<const char*>(ep.rowstat);
            unsigned bb = (unsigned)M * 8u;
            int h = ep.ncb >> 1;                            // host: an even block count
            // opaque copies: nothing below may be hoisted out of the tile loop (the loop has neither the scalar nor the vector
            // registers for tile-invariant addresses; hoisted, they went to VGPR lanes and from there to scratch)
            asm volatile("" : "+s"(tbl), "+s"(bb), "+s"(h));
            const unsigned off = (unsigned)((lane >> 5) ? h : 0) * bb + (unsigned)row * 8u;
            unsigned o[LN_MAX_NCB / 2];                     // per-block lane offsets: vector registers are free behind the epilogue
#pragma unroll
            for (int i = 0; i < LN_MAX_NCB / 2; ++i) o[i] = off + (unsigned)min(i, h - 1) * bb;
            f32x2 e0, e1, e2, e3, e4, e5, e6, e7;
            asm volatile("global_load_dwordx2 %0, %8, %16\n\t"
                         "global_load_dwordx2 %1, %9, %16\n\t"
                         "global_load_dwordx2 %2, %10, %16\n\t"
                         "global_load_dwordx2 %3, %11, %16\n\t"
                         "global_load_dwordx2 %4, %12, %16\n\t"
                         "global_load_dwordx2 %5, %13, %16\n\t"
                         "global_load_dwordx2 %6, %14, %16\n\t"
                         "global_load_dwordx2 %7, %15, %16\n\t"
                         "s_waitcnt vmcnt(0)"
                         : "=&v"(e0), "=&v"(e1), "=&v"(e2), "=&v"(e3), "=&v"(e4), "=&v"(e5), "=&v"(e6), "=&v"(e7)
                         : "v"(o[0]), "v"(o[1]), "v"(o[2]), "v"(o[3]), "v"(o[4]), "v"(o[5]), "v"(o[6]), "v"(o[7]), "s"(tbl)
                         : "memory");
            pre.raw_sum = ln_f32x8{e0[0], e1[0], e2[0], e3[0], e4[0], e5[0], e6[0], e7[0]};
            pre.raw_m2 = ln_f32x8{e0[1], e1[1], e2[1], e3[1], e4[1], e5[1], e6[1], e7[1]};
        } else {
            // entries 0 .. 7: the first half of the row's blocks (0 .. h - 1), entries 8 .. 15: the second half (h .. ncb - 1) --
            // both halves then merge with compile-time step constants (ln_finish)
            const char* const tbl = reinterpret_cast<const char*>(ep.rowstat);
            const int h = ep.ncb >> 1;                      // host: an even block count
#pragma unroll
            for (int i = 0; i < LN_MAX_NCB; ++i) {
                const int blk = (i < LN_MAX_NCB / 2 ? 0 : h) + min(i & (LN_MAX_NCB / 2 - 1), h - 1);
                const float2 p = ln_entry(tbl, (unsigned)M * 8u, blk, (unsigned)row * 8u);
                pre.raw_sum[i] = p.x, pre.raw_m2[i] = p.y;
            }
        }
    }
}
template <int ACTX, int MI>
__device__ __forceinline__ void ln_finish(const Epilogue& ep, int lane, LnPreT<MI>& pre) {
    if constexpr ((ACTX & 8) != 0) {
        if (ep.ncb <= 0) return;
        float mean = 0.f, m2 = 0.f;
        if constexpr (MI > 2) {
            const int h = ep.ncb >> 1;                      // both lane halves merge h blocks (host: an even count)
            mean = pre.raw_sum[0] * (1.0f / 64.0f), m2 = pre.raw_m2[0];
#pragma unroll
            for (int i = 1; i < LN_MAX_NCB / 2; ++i)
                ln_chan_step(make_float2(pre.raw_sum[i], pre.raw_m2[i]), 1.0f / (float)(i + 1), 64.0f * (float)i / (float)(i + 1), i < h, mean, m2);
            // the other half's (mean, M2); lanes < 32 hold the first half of the blocks and are the ones whose pair is used
            const float omean = __shfl_xor(mean, 32, 64), om2 = __shfl_xor(m2, 32, 64);
            pre.rs = (lane < 32) ? ln_combine_halves(mean, m2, omean, om2, h, ep.ncb, ep.eps) : ln_combine_halves(omean, om2, mean, m2, h, ep.ncb, ep.eps);
            return;
        } else {
            // both halves (ln_prefetch: entries 0 .. 7 and 8 .. 15), each as the 256-wide kernels' lane halves merge theirs
            const int h = ep.ncb >> 1;
            constexpr int HB = LN_MAX_NCB / 2;
            float mean_b = pre.raw_sum[HB] * (1.0f / 64.0f), m2_b = pre.raw_m2[HB];
            mean = pre.raw_sum[0] * (1.0f / 64.0f), m2 = pre.raw_m2[0];
#pragma unroll
            for (int i = 1; i < HB; ++i) {
                ln_chan_step(make_float2(pre.raw_sum[i], pre.raw_m2[i]), 1.0f / (float)(i + 1), 64.0f * (float)i / (float)(i + 1), i < h, mean, m2);
                ln_chan_step(make_float2(pre.raw_sum[HB + i], pre.raw_m2[HB + i]), 1.0f / (float)(i + 1), 64.0f * (float)i / (float)(i + 1), i < h, mean_b,
                             m2_b);
            }
            pre.rs = ln_combine_halves(mean, m2, mean_b, m2_b, h, ep.ncb, ep.eps);
            return;
        }
    }
}
// the rows a kernel without prefetch needs (v11 / v12: no folded consumer of the product reaches them; tests do)
__device__ __forceinline__ float2 ln_row_direct(const Epilogue& ep, int M, int row) {
    if (ep.ncb <= 0) return ep.rowstat[row];
    if ((ep.ncb & 1) == 0) {                                 // the consumers' common arithmetic: two halves, then ln_combine_halves
        const int h = ep.ncb >> 1;
        float mean[2], m2[2];
        for (int half = 0; half < 2; ++half) {
            const float2 p0 = ep.rowstat[(long)(half * h) * M + row];
            mean[half] = p0.x * (1.0f / 64.0f), m2[half] = p0.y;
            for (int i = 1; i < h; ++i)
                ln_chan_step(ep.rowstat[(long)(half * h + i) * M + row], 1.0f / (float)(i + 1), 64.0f * (float)i / (float)(i + 1), true, mean[half], m2[half]);
        }
        return ln_combine_halves(mean[0], m2[0], mean[1], m2[1], h, ep.ncb, ep.eps);
    }
    float mean = 0.f, m2 = 0.f;
    for (int b = 0; b < ep.ncb; ++b)
        ln_chan_step(ep.rowstat[(long)b * M + row], 1.0f / (float)(b + 1), 64.0f * (float)b / (float)(b + 1), true, mean, m2);
    const float rstd = rsqrtf(m2 / (64.0f * (float)ep.ncb) + ep.eps);
    return make_float2(rstd, -rstd * mean);
}

// ACTX: the low bits are the activation / aux code (0 none, 1 GELU, 2 GELU + pre-activation, 3 x gelu', 4 q|k|v heads,
// 5 N3 head); bits 3-5 compile the LayerNorm-fold paths in: 8 = consumer (ep.rowstat / ep.colsum), 16 = producer statistics
// (ep.stats), 32 = bf16 twin of an fp32 output (ep.copy).  Template bits, not run-time flags: a run-time `if` inside the
// unrolled passes splits them into basic blocks, and the loads / stores of a block are then no longer issued as batches
// (measured: +6 us on a 56 us GEMM whose fold work is 2 FMAs per element); the instantiations without the bits are
// untouched.
// The fp32 residual values of a wave's MI x 32 x 64 block in the row-owning layout of store_tile_lds's `body` (lane -> row
// mw + 32 i + (lane >> 4) + 4 p, columns nw + 4 (lane & 15) ..): round 6, single-round kernels (v12: one 256 x 128 tile per CU)
// fetch them BEFORE the main loop.  Their epilogue otherwise pays MI dependent HBM round trips at the very end, when all 246
// workgroups of the launch reach it at once (the residual IS the output stream: the loads of row block i + 1 cannot be hoisted
// above the stores of block i), while HBM sits idle during the main loop, whose operands come out of the L2s.
// (32-float vector VALUES per row block, not a float4 array: an array indexed by the row-block loop is still a memory object when
// the register-promotion pass runs -- the loops are unrolled later -- and went to scratch; LnPreT above met the same.)
typedef float res_f32x32 __attribute__((ext_vector_type(32)));
template <int MI>
struct ResPre {
    res_f32x32 v0, v1, v2, v3;                      // row blocks 0 .. 3 (MI <= 4): element 4 p + e = pass p, column e
    __device__ __forceinline__ float4 get(int i, int p) const {
        const res_f32x32 v = i == 0 ? v0 : (i == 1 ? v1 : (i == 2 ? v2 : v3));
        return make_float4(v[4 * p], v[4 * p + 1], v[4 * p + 2], v[4 * p + 3]);
    }
};
template <int MI>
__device__ __forceinline__ void res_prefetch(const Epilogue& ep, int M, int N, int mw, int nw, int lane, ResPre<MI>& pr) {
    static_assert(MI <= 4, "ResPre holds four row blocks");
    const int rr = lane >> 4, ncl = min(nw + (lane & 15) * 4, N - 4);
    auto block = [&](int i) {
        res_f32x32 v;
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const float4 x = *reinterpret_cast<const float4*>(ep.residual + (long)min(mw + i * 32 + rr + 4 * p, M - 1) * ep.ldr + ncl);
            v[4 * p] = x.x, v[4 * p + 1] = x.y, v[4 * p + 2] = x.z, v[4 * p + 3] = x.w;
        }
        return v;
    };
    pr.v0 = block(0);
    if (MI > 1) pr.v1 = block(1);
    if (MI > 2) pr.v2 = block(2);
    if (MI > 3) pr.v3 = block(3);
    asm volatile("" ::: "memory");                  // the loads are issued HERE: nothing below may be scheduled above them, nor they below
}

template <typename TOUT, int ACTX, int RES, int MI>
__device__ __forceinline__ void store_tile_lds(const f32x16 (&acc)[MI][2], float* scr, TOUT* C, long ldc, int M, int N, int mw,
                                               int nw, int lane, const Epilogue& ep, const LnPreT<MI>* pre = nullptr,
                                               const ResPre<MI>* pres = nullptr, bool use_pres = false) {
    constexpr int ACT = ACTX & 7;
    constexpr bool fold = (ACTX & 8) != 0, STATS = (ACTX & 16) != 0, COPY = (ACTX & 32) != 0;
    // The 256-wide kernels call this once per tile from a persistent loop that runs at the register limit.  Lane-derived
    // addresses / row indices of the epilogue are tile-invariant, so the compiler hoists them out of that loop -- into registers
    // the main loop does not have: they were spilled, and every scratch reload in here brings an `s_waitcnt vmcnt(0)` that drains
    // the epilogue's own stores and the next tile's LDS-DMA chunks in flight (nine drains per tile in the statistics producer).
    // An opaque copy of the lane index makes all of it tile-local: a few dozen VALU instructions per tile instead.
    if constexpr (MI > 2) asm volatile("" : "+v"(lane));
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
    const bool wave_in = nw + 64 <= N;                       // producer side needs whole 64-column blocks (host: N % 64 == 0)
    float2* rsl = reinterpret_cast<float2*>(scr + EP_WAVE_FLOATS);         // [128] (rstd, -rstd mean) of rows mw .. (consumers)
    float* const csl = scr + EP_WAVE_FLOATS + 256;                          // [64] colsum of columns nw ..
    if (fold) {                                              // DS operations of a wave execute in order: no barrier needed
        csl[lane] = pre ? pre->cs : ep.colsum[min(nw + lane, N - 1)];
        if (MI > 2 && pre) {                                 // the four waves of a row of waves fetched / merged 32 rows each (ln_prefetch)
            // opaque offset / lane: addresses into the shared table are rebuilt per tile -- as tile-invariants of v10's persistent
            // loop they were spilled, and a scratch reload in the epilogue waits out the next tile's chunks in flight (vmcnt)
            int toff = pre->table_off, lf = lane;
            asm volatile("" : "+s"(toff), "+v"(lf));
            rsl = reinterpret_cast<float2*>(scr + toff);
            if (lf < 32) rsl[pre->slot + lf] = pre->rs;
            // every wave of the workgroup runs this epilogue.  (The marker exempts this barrier from tools/audit_barriers.py's "vmcnt
            // wait in front of every barrier" rule: it orders writes / reads of epilogue scratch only; the LDS-DMA chunks in flight
            // belong to the next tile, which waits for them itself.)
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier ; m324-audit: epilogue scratch only" ::: "memory");
        } else {
            rsl[lane] = pre ? pre->rs : ln_row_direct(ep, M, min(mw + lane, M - 1));
            if (MI > 2) rsl[64 + lane] = ln_row_direct(ep, M, min(mw + 64 + lane, M - 1));
        }
    }
    float2* const stats_wave = (STATS && wave_in) ? ep.stats + (long)(nw >> 6) * M : nullptr;
    // interior tiles (all but the last row / column of tiles) take a copy without per-store predicates, so the 8 LDS
    // reads and the 8 stores of a block are scheduled as batches instead of read-wait-store chains
    auto body = [&](auto checked) {
        constexpr bool CHECK = decltype(checked)::value;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int mb = mw + i * 32 + rr;       // row of pass p: mb + 4 p
            float4 res[8], az[8];
            float2 rs[8];
            const float4 cs = fold ? *reinterpret_cast<const float4*>(csl + cc) : make_float4(0.f, 0.f, 0.f, 0.f);
            float4 xs[8];                                    // STATS: the stored values of the 8 passes
            if (fold) {
#pragma unroll
                for (int p = 0; p < 8; ++p) rs[p] = rsl[i * 32 + rr + 4 * p];
            }
            if (has_res && pres && use_pres) {      // (pres: a compile-time constant per call site; the values stay in registers)
#pragma unroll
                for (int p = 0; p < 8; ++p) res[p] = pres->get(i, p);
            } else if (has_res) {
                // broadcast residual (the decoder's out-projection: the same 2048 point rows under every frame): ONE
                // division per 32-row block, then a compare-subtract per pass (an integer modulo per row cost this
                // epilogue 25 us of the 115 us GEMM); periods shorter than a block keep the per-row form
                const bool wrap = res_mod && ep.res_rows >= 32;
                const int mb0 = wrap ? mb % ep.res_rows : mb;
#pragma unroll
                for (int p = 0; p < 8; ++p) {
                    int mr;
                    if (wrap) {
                        mr = mb0 + 4 * p;
                        mr = mr >= ep.res_rows ? mr - ep.res_rows : mr;
                    } else {
                        mr = CHECK ? min(mb + 4 * p, M - 1) : mb + 4 * p;
                        if (res_mod) mr %= ep.res_rows;
                    }
                    res[p] = ep.res_out ? load4_out<TOUT>(reinterpret_cast<const TOUT*>(ep.residual) + (long)mr * ep.ldr + ncl)
                                        : *reinterpret_cast<const float4*>(ep.residual + (long)mr * ep.ldr + ncl);
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
                if (fold) ln_fold4(x, rs[p], cs, bi);
                else { x.x += bi.x; x.y += bi.y; x.z += bi.z; x.w += bi.w; }
                const int m = mb + 4 * p;
                if constexpr (ACT == 2) {
                    float4 keep = x;                        // the pre-activation, or (M324_AUX_STORE_GELU_GRAD) the activation's derivative there
                    if (ep.aux_mode == M324_AUX_STORE_GELU_GRAD) apply_gelu4_grad<TOUT>(x, keep);
                    else apply_gelu4<TOUT>(x);
                    if (!CHECK || (m < M && nok))
                        store4_out<TOUT>(static_cast<TOUT*>(ep.aux) + (long)m * ep.ldaux + n, keep.x, keep.y, keep.z, keep.w);
                }
                if (ACT == 1) apply_gelu4<TOUT>(x);
                if (ep.gamma) { x.x *= ga.x; x.y *= ga.y; x.z *= ga.z; x.w *= ga.w; }
                if (has_res) { x.x += res[p].x; x.y += res[p].y; x.z += res[p].z; x.w += res[p].w; }
                if constexpr (ACT == 3) {
                    if (ep.aux_mode == M324_AUX_MUL) {       // aux already holds the factor (the forward's M324_AUX_STORE_GELU_GRAD)
                        x.x *= az[p].x; x.y *= az[p].y; x.z *= az[p].z; x.w *= az[p].w;
                    } else {
                        x.x *= gelu_grad<TOUT>(az[p].x); x.y *= gelu_grad<TOUT>(az[p].y);
                        x.z *= gelu_grad<TOUT>(az[p].z); x.w *= gelu_grad<TOUT>(az[p].w);
                    }
                }
                if constexpr (STATS) xs[p] = x;
                if (!CHECK || (m < M && nok)) {
                    long orow = m;
                    if (remap) orow = (long)(m / ep.row_gin) * ep.row_gout + (m % ep.row_gin) + ep.row_off;
                    store4_out<TOUT>(C + orow * ldc + n, x.x, x.y, x.z, x.w);
                    if constexpr (COPY) store4_out<bf16_t>(ep.copy + (long)m * ep.ldcopy + n, x.x, x.y, x.z, x.w);
                }
            }
            if constexpr (STATS) {
                // every lane takes part in the DPP sums; all 16 lanes of a row end up with its pair, lane g == p keeps it
                // (no branch), so that lanes g < 8 hold rows rr + 4 g: one store per 32-row block, 32 lanes x 8 bytes
                float2 st[8], keep = make_float2(0.f, 0.f);
                ln_stats4_batch<8>(xs, st);
#pragma unroll
                for (int p = 0; p < 8; ++p) {
                    const bool mine = (lane & 15) == p;
                    keep.x = mine ? st[p].x : keep.x;
                    keep.y = mine ? st[p].y : keep.y;
                }
                const int m = mb + 4 * (lane & 15);
                if (stats_wave && (lane & 15) < 8 && (!CHECK || m < M)) stats_wave[m] = keep;
            }
        }
    };
    // bf16 interior tiles without row remap (residual: fp32, or the bf16 stream itself): 8 columns per lane, so a row is 8 lanes x 16 bytes and one
    // store instruction writes 8 whole 128-byte lines (half as many store instructions as the 4-column form)
    auto body8 = [&]() {
        const int r8 = lane >> 3, c8 = (lane & 7) * 8;
        const int n8 = nw + c8;
        float4 b0 = make_float4(0.f, 0.f, 0.f, 0.f), b1 = b0, g0 = make_float4(1.f, 1.f, 1.f, 1.f), g1 = g0;
        if (ep.bias) { b0 = *reinterpret_cast<const float4*>(ep.bias + n8); b1 = *reinterpret_cast<const float4*>(ep.bias + n8 + 4); }
        if (ep.gamma) { g0 = *reinterpret_cast<const float4*>(ep.gamma + n8); g1 = *reinterpret_cast<const float4*>(ep.gamma + n8 + 4); }
        float4 cs0 = make_float4(0.f, 0.f, 0.f, 0.f), cs1 = cs0;
        if (fold) { cs0 = *reinterpret_cast<const float4*>(csl + c8); cs1 = *reinterpret_cast<const float4*>(csl + c8 + 4); }
        const float* rd8 = scr + r8 * EP_LD + c8;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            float4 xa[4], ya[4];                             // STATS: the stored values of the 4 passes
            float2 rs[4];
            if (fold) {
#pragma unroll
                for (int p = 0; p < 4; ++p) rs[p] = rsl[i * 32 + p * 8 + r8];
            }
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<float4*>(wr + j * 32 + 8 * g) =
                        make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
            float4 v0[4], v1[4], z0[4], z1[4], ra[4], rb[4];
            if constexpr (ACT == 3) {
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const TOUT* zp = static_cast<const TOUT*>(ep.aux) + (long)(mw + i * 32 + p * 8 + r8) * ep.ldaux + n8;
                    z0[p] = load4_out<TOUT>(zp);
                    z1[p] = load4_out<TOUT>(zp + 4);
                }
            }
            if constexpr (RES != 0) {
                // residual rows of the 4 passes: same row, or row modulo res_rows (one division per block, then steps of 8);
                // 16 bytes per lane when the residual is the bf16 stream itself, two float4 when it is fp32
                int mr = mw + i * 32 + r8;
                if (res_mod) mr %= ep.res_rows;
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    if (ep.res_out) {
                        const TOUT* rp = reinterpret_cast<const TOUT*>(ep.residual) + (long)mr * ep.ldr + n8;
                        ra[p] = load4_out<TOUT>(rp);
                        rb[p] = load4_out<TOUT>(rp + 4);
                    } else {
                        const float* rp = ep.residual + (long)mr * ep.ldr + n8;
                        ra[p] = *reinterpret_cast<const float4*>(rp);
                        rb[p] = *reinterpret_cast<const float4*>(rp + 4);
                    }
                    mr += 8;
                    if (res_mod) { while (mr >= ep.res_rows) mr -= ep.res_rows; }
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
                if (fold) { ln_fold4(x, rs[p], cs0, b0); ln_fold4(y, rs[p], cs1, b1); }
                else {
                    x.x += b0.x; x.y += b0.y; x.z += b0.z; x.w += b0.w;
                    y.x += b1.x; y.y += b1.y; y.z += b1.z; y.w += b1.w;
                }
                const long m = mw + i * 32 + p * 8 + r8;
                if constexpr (ACT == 2) {
                    float4 kx = x, ky = y;                  // the pre-activation, or (M324_AUX_STORE_GELU_GRAD) the activation's derivative there
                    if (ep.aux_mode == M324_AUX_STORE_GELU_GRAD) apply_gelu8_grad<TOUT>(x, y, kx, ky);
                    else apply_gelu8<TOUT>(x, y);
                    *reinterpret_cast<uint4*>(static_cast<TOUT*>(ep.aux) + m * ep.ldaux + n8) =
                        make_uint4(pack_bf16x2(kx.x, kx.y), pack_bf16x2(kx.z, kx.w), pack_bf16x2(ky.x, ky.y), pack_bf16x2(ky.z, ky.w));
                }
                if (ACT == 1) apply_gelu8<TOUT>(x, y);
                if (ep.gamma) {
                    x.x *= g0.x; x.y *= g0.y; x.z *= g0.z; x.w *= g0.w;
                    y.x *= g1.x; y.y *= g1.y; y.z *= g1.z; y.w *= g1.w;
                }
                if constexpr (RES != 0) {
                    x.x += ra[p].x; x.y += ra[p].y; x.z += ra[p].z; x.w += ra[p].w;
                    y.x += rb[p].x; y.y += rb[p].y; y.z += rb[p].z; y.w += rb[p].w;
                }
                if constexpr (ACT == 3) {
                    if (ep.aux_mode == M324_AUX_MUL) {       // aux already holds the factor (the forward's M324_AUX_STORE_GELU_GRAD)
                        x.x *= z0[p].x; x.y *= z0[p].y; x.z *= z0[p].z; x.w *= z0[p].w;
                        y.x *= z1[p].x; y.y *= z1[p].y; y.z *= z1[p].z; y.w *= z1[p].w;
                    } else {
                        x.x *= gelu_grad<TOUT>(z0[p].x); x.y *= gelu_grad<TOUT>(z0[p].y);
                        x.z *= gelu_grad<TOUT>(z0[p].z); x.w *= gelu_grad<TOUT>(z0[p].w);
                        y.x *= gelu_grad<TOUT>(z1[p].x); y.y *= gelu_grad<TOUT>(z1[p].y);
                        y.z *= gelu_grad<TOUT>(z1[p].z); y.w *= gelu_grad<TOUT>(z1[p].w);
                    }
                }
                if constexpr (STATS) { xa[p] = x; ya[p] = y; }
                store16(C + m * ldc + n8, ep.stream != 0, pack_bf16x2(x.x, x.y), pack_bf16x2(x.z, x.w), pack_bf16x2(y.x, y.y), pack_bf16x2(y.z, y.w));
            }
            if constexpr (STATS) {                          // lane (r8, g) keeps the pair of row r8 + 8 g: lanes g < 4 store
                float2 st[4], keep = make_float2(0.f, 0.f);
                ln_stats8_batch<4>(xa, ya, st);
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const bool mine = (lane & 7) == p;
                    keep.x = mine ? st[p].x : keep.x;
                    keep.y = mine ? st[p].y : keep.y;
                }
                if (stats_wave && (lane & 7) < 4) stats_wave[mw + i * 32 + r8 + 8 * (lane & 7)] = keep;
            }
        }
    };
    // fused q|k|v projection (M324_AUX_QKV_HEADS): this wave's 64 columns are one head of q (which = 0), k (1) or v (2).
    // 8 columns per lane: a row's 64 values sit in the 8 lanes that share lane >> 3, so the per-head RMSNorm is three
    // xor-shuffles inside the group, and a (token, head) row leaves as 8 x 16 bytes = one 128-byte line.
    auto body_qkv = [&]() {
        const int r8 = lane >> 3, c8 = (lane & 7) * 8;
        // a projection may carry only the k|v or only the q part (cross-attention: NULL outputs in front shift `which`)
        const int hc = ep.qkv_H * 64, which = nw / hc + (ep.qkv_out[0] ? 0 : 1), head = (nw % hc) >> 6;
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
        float4 cs0 = make_float4(0.f, 0.f, 0.f, 0.f), cs1 = cs0;
        if (fold) { cs0 = *reinterpret_cast<const float4*>(csl + c8); cs1 = *reinterpret_cast<const float4*>(csl + c8 + 4); }
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<float4*>(wr + j * 32 + 8 * g) =
                        make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
            if (which == 2 && ep.qkv_vt) {
                // V of this head, transposed: four lanes share a column d of the 32 x 64 block, lane (d = (lane >> 2) + 16 p,
                // c = lane & 3) gathers the 8 tokens of 16-byte chunk c (b32 reads, two-way bank conflicts) and a store
                // instruction writes 16 Vt rows x 64 contiguous bytes (one lane per column wrote 64 rows x 16 bytes: four times
                // as many partial-line writes).  Token quarters of each 16-group go out in the order 0, 2, 1, 3 (what the
                // attention's P^T fragments contract): chunk c holds tokens base .. base+3 and base+8 .. base+11,
                // base = 16 (c >> 1) + 4 (c & 1).  qkv_L % 128 == 0 (host-checked): a 32-token block never straddles a
                // batch and Vt has no padding.
                const int m0b = mw + i * 32;
                if (m0b < M && nw + 64 <= N) {
                    const int bb = m0b / ep.qkv_L, ll = m0b - bb * ep.qkv_L;
                    const int cq = lane & 3, base = 16 * (cq >> 1) + 4 * (cq & 1);
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        const int d = (lane >> 2) + 16 * p;
                        const float bd = ep.bias ? ep.bias[nw + d] : 0.f;
                        const float cd = fold ? csl[d] : 0.f;
                        float u[8];
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            if (fold) {
                                const float2 ra = rsl[i * 32 + base + k], rb = rsl[i * 32 + base + 8 + k];
                                u[k] = fmaf(ra.x, scr[(base + k) * EP_LD + d], fmaf(ra.y, cd, bd));
                                u[4 + k] = fmaf(rb.x, scr[(base + 8 + k) * EP_LD + d], fmaf(rb.y, cd, bd));
                            } else {
                                u[k] = scr[(base + k) * EP_LD + d] + bd;
                                u[4 + k] = scr[(base + 8 + k) * EP_LD + d] + bd;
                            }
                        }
                        bf16_t* dst = ep.qkv_out[2] + (((long)bb * ep.qkv_H + head) * 64 + d) * (long)ep.qkv_L + ll + cq * 8;
                        *reinterpret_cast<uint4*>(dst) =
                            make_uint4(pack_bf16x2(u[0], u[1]), pack_bf16x2(u[2], u[3]), pack_bf16x2(u[4], u[5]), pack_bf16x2(u[6], u[7]));
                    }
                }
                continue;
            }
            float4 v0[4], v1[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                v0[p] = *reinterpret_cast<const float4*>(rd8 + p * 8 * EP_LD);
                v1[p] = *reinterpret_cast<const float4*>(rd8 + p * 8 * EP_LD + 4);
            }
            // token of pass p: m = mw + 32 i + r8 + 8 p  ->  (batch, position); one division per block, then steps of 8
            const int m0r = mw + i * 32 + r8;
            int bb = m0r / ep.qkv_L, ll = m0r - bb * ep.qkv_L;
            float2 rst[4];
            if (fold) {
#pragma unroll
                for (int p = 0; p < 4; ++p) rst[p] = rsl[i * 32 + r8 + 8 * p];
            }
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                float4 x = v0[p], y = v1[p];
                if (fold) { ln_fold4(x, rst[p], cs0, b0); ln_fold4(y, rst[p], cs1, b1); }
                else {
                    x.x += b0.x; x.y += b0.y; x.z += b0.z; x.w += b0.w;
                    y.x += b1.x; y.y += b1.y; y.z += b1.z; y.w += b1.w;
                }
                float rs = post;
                if (norm) {
                    float ss = x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w + y.x * y.x + y.y * y.y + y.z * y.z + y.w * y.w;
                    ss = group8_sum(ss);                      // DPP adds (no LDS crossbar): the 8 lanes of the row
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
    // M324_AUX_N3 (the head: Linear -> GELU -> Linear 768 -> 3): the GELU output is not stored at all; every wave contracts its
    // 64 columns with the three rows of the last layer's weight and writes 3 partial sums per row into part[column block][M][3]
    // (8 consecutive rows = 96 contiguous bytes per store group); m324_n3_finish adds the N / 64 column blocks in a fixed order.
    auto body_n3 = [&]() {
        const int r8 = lane >> 3, c8 = (lane & 7) * 8;
        const int n8 = nw + c8;
        float4 b0 = make_float4(0.f, 0.f, 0.f, 0.f), b1 = b0;
        if (ep.bias) { b0 = *reinterpret_cast<const float4*>(ep.bias + n8); b1 = *reinterpret_cast<const float4*>(ep.bias + n8 + 4); }
        float4 w0[3], w1[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            w0[j] = *reinterpret_cast<const float4*>(ep.qkv_w[0] + (long)j * N + n8);
            w1[j] = *reinterpret_cast<const float4*>(ep.qkv_w[0] + (long)j * N + n8 + 4);
        }
        const float* rd8 = scr + r8 * EP_LD + c8;
        float* part = static_cast<float*>(ep.aux) + (long)(nw >> 6) * M * 3;
        float4 cs0 = make_float4(0.f, 0.f, 0.f, 0.f), cs1 = cs0;
        if (fold) { cs0 = *reinterpret_cast<const float4*>(csl + c8); cs1 = *reinterpret_cast<const float4*>(csl + c8 + 4); }
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
            float2 rst[4];
            if (fold) {
#pragma unroll
                for (int p = 0; p < 4; ++p) rst[p] = rsl[i * 32 + p * 8 + r8];
            }
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                float4 x = v0[p], y = v1[p];
                if (fold) { ln_fold4(x, rst[p], cs0, b0); ln_fold4(y, rst[p], cs1, b1); }
                else {
                    x.x += b0.x; x.y += b0.y; x.z += b0.z; x.w += b0.w;
                    y.x += b1.x; y.y += b1.y; y.z += b1.z; y.w += b1.w;
                }
                apply_gelu8<TOUT>(x, y);
                float sj[3];
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    float t = x.x * w0[j].x + x.y * w0[j].y + x.z * w0[j].z + x.w * w0[j].w + y.x * w1[j].x + y.y * w1[j].y +
                              y.z * w1[j].z + y.w * w1[j].w;
                    sj[j] = group8_sum(t);
                }
                const long m = mw + i * 32 + p * 8 + r8;
                if ((lane & 7) == 0 && m < M && nw + 64 <= N) {
                    float* o = part + m * 3;
                    o[0] = sj[0]; o[1] = sj[1]; o[2] = sj[2];
                }
            }
        }
    };
    const bool interior = mw + MI * 32 <= M && nw + 64 <= N;
    if constexpr (ACT == 4) {
        body_qkv();
        return;
    }
    if constexpr (ACT == 5) {
        body_n3();
        return;
    }
    if constexpr (sizeof(TOUT) == 2 && ACT != 4) {
        // with a residual: it must be present, without output row remap, and 16-byte addressable in its own dtype
        const bool res8 = RES == 0 || (has_res && !remap && (reinterpret_cast<uintptr_t>(ep.residual) & 15) == 0 &&
                                       (ep.ldr & (ep.res_out ? 7 : 3)) == 0 && (!res_mod || ep.res_rows >= 8));
        if (interior && res8 && (ldc & 7) == 0 && (reinterpret_cast<uintptr_t>(C) & 15) == 0 &&
            (!ep.bias || (reinterpret_cast<uintptr_t>(ep.bias) & 15) == 0) &&
            (ACT < 2 || ((ep.ldaux & 7) == 0 && (reinterpret_cast<uintptr_t>(ep.aux) & 15) == 0))) {
            body8();
            return;
        }
    }
    if (interior) body(std::false_type{});
    else body(std::true_type{});
}

typedef __attribute__((address_space(3))) void lds_ptr_t;
typedef __attribute__((address_space(1))) const void glb_ptr_t;

// LDS-DMA pieces are BUFFER loads (tools/coissue_lab: beside a wave's MFMA stream a buffer_load ... lds piece costs its SIMD ~5
// cycles, a global_load_lds piece with per-lane 64-bit addresses ~65): the resource starts at the tile's first row, the lane
// carries a 32-bit byte offset inside the tile's row panel, the K position rides in the scalar offset.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t dma_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7FFFFFFF, 0x00020000);
}
// A resource without records: every lane of a piece issued through it is out of range -- the piece still counts in vmcnt and
// writes its KiB of LDS (zeros), but fetches nothing.  The ring kernels issue a fixed number of pieces per K-stage (counted vmcnt
// waits, branch-free loop); the look-ahead pieces past the end of K -- 1.5 of a tile's 12 + 1.5 stages at K = 768, 11 % of its
// operand traffic -- go through this resource instead of fetching the last stage again (round 5).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t dma_rsrc_none(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0, 0x00020000);
}
__device__ __forceinline__ void dma_piece(__amdgpu_buffer_rsrc_t rs, unsigned char* lds, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t*)lds, 16, voff, soff, 0, 0);
}

constexpr int BM5 = 256, BN5 = 256;              // tile of the 256-wide kernels (v5, v7, v10, v11)
constexpr int CHUNK10 = 256 * ROWB;              // 32 KiB: one operand's 256 rows x 64 k of one K-stage (v10, v11)

#define M324_BARRIER()                             \
    do {                                           \
        asm volatile("s_barrier" ::: "memory");    \
        __builtin_amdgcn_sched_barrier(0);         \
    } while (0)

}  // namespace
