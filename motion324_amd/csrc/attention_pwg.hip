// attn_pwg_kernel: the long-sequence attention forward of m324_attention (bf16, head_dim 64, pre-scaled Q, transposed
// key-permuted Vt) as ONE wave per SIMD with a hand-placed instruction stream.
//
// Round 4 measurement behind it (tools/issue_lab.cpp): on gfx950 a wave's plain VALU instructions issue in the shadow of its
// OWN MFMAs when they sit between them in program order -- MFMA + 4 v_fma_f32 = 33.5 cycles per MFMA, + 4 plain + 2 v_exp_f32
// = 38-41 -- while a second wave of the SIMD gets one VALU issue per MFMA of a partner that sits on a blocked MFMA (round 3's
// coissue_lab).  The eight-wave kernel in attention.hip leaves the interleaving to the compiler and to cross-wave overlap: 50 %
// matrix-pipe time.  Here a workgroup is 4 waves = 256 queries (64 per wave = two 32-row blocks), each wave owns its SIMD and
// the whole register file (416 registers), and the tile loop is a software pipeline written out by gen_attn_pwg.py:
//     S(t+1) = K(t+1) Q^T - m_ref   ||   P(t) = exp2(S(t)), row sums, bf16 pack, max check of S(t+1)   ||   O += Vt(t) P(t)
// 32 MFMAs per 64-key tile and wave with ~10 issue slots of softmax, LDS fragment reads and LDS-DMA between consecutive MFMAs.
// Same mathematics as attn_bf16_kernel: swapped formulation (a query's scores in one lane pair), lazy reference maximum folded
// into the accumulator's initial value (MR = -m_ref is the C operand of a tile's first MFMA: no per-tile register fill),
// log2-domain LSE, LDS tile images and XOR swizzle of attention.hip, four-stage K / Vt ring filled by buffer_load ... lds.
// Reference: xformers memory_efficient_attention at model/transformer.py:209-214 (global blocks, Pcd_motion.py:401-405).
#include "common.h"

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

#define PWG_KERNEL attn_pwg_kernel
#define PWG_ASM_INC "attn_pwg_asm.inc"
#include "attn_pwg_kernel.inl"
#undef PWG_KERNEL
#undef PWG_ASM_INC

// M324_ATTN_SCORES_BOUNDED: the same pipeline without the lazy reference maximum (no per-lane maxima, no vote, no rescale
// path, the tile's first MFMA starts from the inline constant 0): 96 + 64 instead of 130 + 64 softmax instructions per tile.
#define PWG_KERNEL attn_pwg_bounded_kernel
#define PWG_ASM_INC "attn_pwg_bounded_asm.inc"
#include "attn_pwg_kernel.inl"
#undef PWG_KERNEL
#undef PWG_ASM_INC


}  // namespace

// Called by m324_attention's chooser (attention.hip); returns the launch status through hipGetLastError there.
void m324_attn_pwg_launch(const void* Q, long q_bstride, const void* K, const void* Vt, void* O, long ldo, int B, int H, int Lq, int Lk,
                          float* lse, bool bounded, hipStream_t s) {
    const int Lkp = (Lk + 63) / 64 * 64;
    const int nqt = ceil_div(Lq, 256);
#define PWG_LAUNCH(KERNEL)                                                                                                  \
    hipLaunchKernelGGL(KERNEL, dim3((unsigned)((long)nqt * H * B)), dim3(256), 0, s, (const bf16_t*)Q, q_bstride, (const bf16_t*)K, \
                       (const bf16_t*)Vt, (bf16_t*)O, ldo, H, Lq, Lk, Lkp, lse, nqt)
    if (bounded) PWG_LAUNCH(attn_pwg_bounded_kernel);
    else PWG_LAUNCH(attn_pwg_kernel);
#undef PWG_LAUNCH
}
