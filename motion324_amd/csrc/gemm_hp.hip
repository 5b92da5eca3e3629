// m324_gemm, schedule v15: the K = 768 GEMM as ONE wave per SIMD with a hand-placed instruction stream in which the PREVIOUS tile's
// epilogue (LayerNorm fold / bias, erf-polynomial GELU, bf16 pack, stores) is issued between the MFMAs of the current tile's main loop.
//
// Why (profiles/r05_gemm_labs.md): at K = 768 a 256 x 128 tile is 12.3 k cycles of MFMA next to 1200-2000 VALU instructions of
// epilogue per wave; a partner workgroup's epilogue does not hide beside a dense MFMA wave (v14), and hipcc does not place a
// deferred epilogue's instructions between the MFMAs (tools/lab_src/de_lab.hip: every filler added its issue time).  A wave's own
// plain VALU instructions DO issue in the shadow of its MFMAs when they sit between them in program order (tools/issue_lab) -- the
// fact the attention stream (attention_pwg.hip) rests on.  gen_gemm_hp.py writes the stream: schedule, register map, hazard and
// wait-count checks are documented there.
//
// Shapes: bf16 A [M, 768] and W [N, 768] (row-major, K contiguous), bf16 C, N % 128 == 0, optional bias, no residual / gamma / aux;
// epilogues: bias (+ GELU), and the LayerNorm-fold consumer forms of both with a MERGED row-statistics table (ep.ncb == 0).
// Reference: the Linear layers of model/transformer.py:40-66 (MLP fc1 + GELU), :84-144 (projections).
#include "gemm_tile.h"

namespace {

typedef __attribute__((ext_vector_type(4))) int i32x4;

__device__ __forceinline__ i32x4 rsrc_words(const void* base, long bytes) {
    const unsigned long p = (unsigned long)base;
    i32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)p);
    r[1] = __builtin_amdgcn_readfirstlane((int)((p >> 32) & 0xffffu));
    r[2] = __builtin_amdgcn_readfirstlane((int)(bytes > 0x7FFFFFFFl ? 0x7FFFFFFFl : bytes));
    r[3] = 0x00020000;
    return r;
}

constexpr int HP_STAGE = 49152;                   // one ring buffer: X 32 KiB + W 16 KiB
constexpr int HP_TABLE_BYTES = 8192;              // tile table: 48 bytes per entry
constexpr int HP_SCRATCH = 2048;                  // store scratch per wave: 16 rows x 128 bytes
constexpr int HP_MAX_TILES = HP_TABLE_BYTES / 48 - 2;

#define HP_ST_FLAG ""
#define HP_KERNEL gemm_hp_gelu_kernel
#define HP_ASM_INC "gemm_hp_gelu.inc"
#include "gemm_hp_kernel.inl"
#undef HP_KERNEL
#undef HP_ASM_INC
#define HP_KERNEL gemm_hp_fold_gelu_kernel
#define HP_ASM_INC "gemm_hp_fold_gelu.inc"
#include "gemm_hp_kernel.inl"
#undef HP_KERNEL
#undef HP_ASM_INC
#define HP_KERNEL gemm_hp_plain_kernel
#define HP_ASM_INC "gemm_hp_plain.inc"
#include "gemm_hp_kernel.inl"
#undef HP_KERNEL
#undef HP_ASM_INC
#define HP_KERNEL gemm_hp_fold_kernel
#define HP_ASM_INC "gemm_hp_fold.inc"
#include "gemm_hp_kernel.inl"
#undef HP_KERNEL
#undef HP_ASM_INC
// the same streams with nontemporal stores (outputs above M324_NT_MB: ep.stream)
#undef HP_ST_FLAG
#define HP_ST_FLAG " nt"
#define HP_KERNEL gemm_hp_gelu_nt_kernel
#define HP_ASM_INC "gemm_hp_gelu.inc"
#include "gemm_hp_kernel.inl"
#undef HP_KERNEL
#undef HP_ASM_INC
#define HP_KERNEL gemm_hp_fold_gelu_nt_kernel
#define HP_ASM_INC "gemm_hp_fold_gelu.inc"
#include "gemm_hp_kernel.inl"
#undef HP_KERNEL
#undef HP_ASM_INC
#define HP_KERNEL gemm_hp_plain_nt_kernel
#define HP_ASM_INC "gemm_hp_plain.inc"
#include "gemm_hp_kernel.inl"
#undef HP_KERNEL
#undef HP_ASM_INC
#define HP_KERNEL gemm_hp_fold_nt_kernel
#define HP_ASM_INC "gemm_hp_fold.inc"
#include "gemm_hp_kernel.inl"
#undef HP_KERNEL
#undef HP_ASM_INC
#undef HP_ST_FLAG

int hp_cus() {
    static const int n_cu = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        return n - (n & 7);                                  // a multiple of 8: tile t and t + grid share an XCD
    }();
    return n_cu;
}

}  // namespace

namespace m324 {

// what the stream handles; the chooser (gemm.hip pick_variant) asks before it sends a GEMM here
bool hp_ok(const m324_gemm_args* a, const Epilogue& ep, int act_code, int res_code) {
    if (a->K != 768 || a->N % 128 != 0 || a->M < 1 || a->out_dtype != M324_BF16 || res_code != 0) return false;
    if (act_code < 0) return false;
    if (act_code != 0 && act_code != 1 && act_code != 8 && act_code != 9) return false;
    if (ep.gamma || ep.aux || ep.stats || ep.copy || ep.row_gin > 0) return false;
    if ((act_code & 8) && (!ep.rowstat || !ep.colsum || ep.ncb > 0)) return false;
    if ((a->lda & 7) || (a->ldw & 7) || (a->ldc & 7) || ((uintptr_t)a->A & 15) || ((uintptr_t)a->W & 15) || ((uintptr_t)a->C & 15) || ((uintptr_t)ep.bias & 15) ||
        ((uintptr_t)ep.colsum & 15))
        return false;
    if ((long)a->M * a->lda * 2 >= 0x7fffffffl || 256l * a->ldc * 2 >= 0x7fffffffl || (long)a->N * a->ldw * 2 >= 0x7fffffffl) return false;
    const int ntiles = (a->N / 128) * ceil_div(a->M, 256), grid = ntiles < hp_cus() ? ntiles : hp_cus();
    return ceil_div(ntiles, grid) <= HP_MAX_TILES;
}

int launch_hp(const m324_gemm_args* a, hipStream_t s, const Epilogue& ep, int act_code, int res_code, int xcd_remap) {
    if (!hp_ok(a, ep, act_code, res_code)) M324_FAIL(M324_ERR_UNSUPPORTED, "m324_gemm: schedule v15 does not take this GEMM (K = 768, bf16, bias, act=%d res=%d)", act_code, res_code);
    const int ntn = a->N / 128, ntiles = ntn * ceil_div(a->M, 256);
    const dim3 grid(ntiles < hp_cus() ? ntiles : hp_cus());
#define M324_HP(KERNEL)                                                                                                              \
    hipLaunchKernelGGL(KERNEL, grid, dim3(256), 0, s, (const bf16_t*)a->A, a->lda, (const bf16_t*)a->W, a->ldw, (bf16_t*)a->C, a->ldc, a->M, a->N, \
                       ep.bias, ep.colsum, ep.rowstat, ntn, ntiles, xcd_remap)
    switch (act_code + (ep.stream ? 16 : 0)) {
        case 0: M324_HP(gemm_hp_plain_kernel); break;
        case 1: M324_HP(gemm_hp_gelu_kernel); break;
        case 8: M324_HP(gemm_hp_fold_kernel); break;
        case 9: M324_HP(gemm_hp_fold_gelu_kernel); break;
        case 16: M324_HP(gemm_hp_plain_nt_kernel); break;
        case 17: M324_HP(gemm_hp_gelu_nt_kernel); break;
        case 24: M324_HP(gemm_hp_fold_nt_kernel); break;
        default: M324_HP(gemm_hp_fold_gelu_nt_kernel); break;
    }
#undef M324_HP
    return M324_OK;
}

int hp_grid(const m324_gemm_args* a) {
    const int ntiles = (a->N / 128) * ceil_div(a->M, 256);
    return ntiles < hp_cus() ? ntiles : hp_cus();
}

}  // namespace m324
