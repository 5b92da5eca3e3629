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
};

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * ROWB + ((chunk ^ ((row >> 1) & 7)) << 4); }

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
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
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
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bfr[j][e], acc[i][j], 0, 0, 0);
            }
        }
        if (more) store_tile((kt + 1) & 1);
        __syncthreads();
    }

    // ---- epilogue.  32x32 C layout: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    const int res_rows = ep.res_rows > 0 ? ep.res_rows : M;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn * 64 + j * 32 + l31;
        if (n >= N) continue;
        const float bias = ep.bias ? ep.bias[n] : 0.f;
        const float gamma = ep.gamma ? ep.gamma[n] : 1.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (m >= M) continue;
                float v = acc[i][j][r] + bias;
                if (ep.act == M324_ACT_GELU) v = gelu_erf(v);
                v *= gamma;
                if (ep.residual) v += ep.residual[(long)(m % res_rows) * ep.ldr + n];
                long orow = m;
                if (ep.row_gin > 0) orow = (long)(m / ep.row_gin) * ep.row_gout + (m % ep.row_gin) + ep.row_off;
                Elem<TOUT>::store(C + orow * ldc + n, v);
            }
        }
    }
}

template <typename TIN, typename TOUT>
int launch(const m324_gemm_args* a, hipStream_t s) {
    Epilogue ep{a->bias, a->gamma, a->residual, a->ldr, a->res_rows, a->act, a->row_gin, a->row_gout, a->row_off};
    dim3 grid(ceil_div(a->N, BN), ceil_div(a->M, BM));
    hipLaunchKernelGGL((gemm_kernel<TIN, TOUT>), grid, dim3(256), 0, s, (const TIN*)a->A, a->lda, (const TIN*)a->W,
                       a->ldw, (TOUT*)a->C, a->ldc, a->M, a->N, a->K, ep);
    M324_CHECK_LAUNCH("m324_gemm");
    return M324_OK;
}

}  // namespace

extern "C" int m324_gemm(const m324_gemm_args* a, void* stream) {
    M324_REQUIRE(a && a->A && a->W && a->C, "m324_gemm: null pointer");
    M324_REQUIRE(a->M > 0 && a->N > 0 && a->K > 0, "m324_gemm: empty problem M=%d N=%d K=%d", a->M, a->N, a->K);
    const int bk = a->in_dtype == M324_BF16 ? 64 : 32;
    M324_REQUIRE(a->K % bk == 0, "m324_gemm: K=%d must be a multiple of %d", a->K, bk);
    const int esz = a->in_dtype == M324_BF16 ? 2 : 4;
    M324_REQUIRE((a->lda * esz) % 16 == 0 && (a->ldw * esz) % 16 == 0 && ((uintptr_t)a->A % 16) == 0 &&
                     ((uintptr_t)a->W % 16) == 0,
                 "m324_gemm: A/W rows must be 16-byte aligned");
    M324_REQUIRE(a->lda >= a->K && a->ldw >= a->K && a->ldc >= a->N, "m324_gemm: leading dimension too small");
    M324_REQUIRE(!a->residual || a->ldr >= a->N, "m324_gemm: ldr too small");
    hipStream_t s = (hipStream_t)stream;
    if (a->in_dtype == M324_BF16 && a->out_dtype == M324_BF16) return launch<bf16_t, bf16_t>(a, s);
    if (a->in_dtype == M324_BF16 && a->out_dtype == M324_F32) return launch<bf16_t, float>(a, s);
    if (a->in_dtype == M324_F32 && a->out_dtype == M324_F32) return launch<float, float>(a, s);
    M324_FAIL(M324_ERR_UNSUPPORTED, "m324_gemm: unsupported dtype pair in=%d out=%d", a->in_dtype, a->out_dtype);
}
