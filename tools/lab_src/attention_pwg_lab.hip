// Lab kernels of the one-wave-per-SIMD attention (NOT part of libm324's product build): the timing-only ablation streams that
// motion324_amd/csrc/gen_attn_pwg.py --lab writes (attn_pwg_lab1..9.inc: wrong results, equal launch geometry) and the
// stamped stream (lab8: s_memtime phase sums of one workgroup, written behind the LSE rows).  tools/build_pwg_lab.sh links this
// file into tools/lablibs/libm324_pwglab.so; tools/pwg_check.py --ablate / --trace calls m324_lab_attn_pwg.
#include "../../motion324_amd/csrc/common.h"

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

#define PWG_KERNEL attn_pwg_lab1_kernel
#define PWG_ASM_INC "attn_pwg_lab1.inc"
#include "../../motion324_amd/csrc/attn_pwg_kernel.inl"
#undef PWG_KERNEL
#undef PWG_ASM_INC
#define PWG_KERNEL attn_pwg_lab2_kernel
#define PWG_ASM_INC "attn_pwg_lab2.inc"
#include "../../motion324_amd/csrc/attn_pwg_kernel.inl"
#undef PWG_KERNEL
#undef PWG_ASM_INC
#define PWG_KERNEL attn_pwg_lab3_kernel
#define PWG_ASM_INC "attn_pwg_lab3.inc"
#include "../../motion324_amd/csrc/attn_pwg_kernel.inl"
#undef PWG_KERNEL
#undef PWG_ASM_INC
#define PWG_KERNEL attn_pwg_lab4_kernel
#define PWG_ASM_INC "attn_pwg_lab4.inc"
#include "../../motion324_amd/csrc/attn_pwg_kernel.inl"
#undef PWG_KERNEL
#undef PWG_ASM_INC
#define PWG_KERNEL attn_pwg_lab5_kernel
#define PWG_ASM_INC "attn_pwg_lab5.inc"
#include "../../motion324_amd/csrc/attn_pwg_kernel.inl"
#undef PWG_KERNEL
#undef PWG_ASM_INC
#define PWG_KERNEL attn_pwg_lab6_kernel
#define PWG_ASM_INC "attn_pwg_lab6.inc"
#include "../../motion324_amd/csrc/attn_pwg_kernel.inl"
#undef PWG_KERNEL
#undef PWG_ASM_INC
#define PWG_KERNEL attn_pwg_lab7_kernel
#define PWG_ASM_INC "attn_pwg_lab7.inc"
#include "../../motion324_amd/csrc/attn_pwg_kernel.inl"
#undef PWG_KERNEL
#undef PWG_ASM_INC
#define PWG_KERNEL attn_pwg_lab9_kernel
#define PWG_ASM_INC "attn_pwg_lab9.inc"
#include "../../motion324_amd/csrc/attn_pwg_kernel.inl"
#undef PWG_KERNEL
#undef PWG_ASM_INC

// the stamped stream: the product wrapper plus five phase sums handed out of the asm statement
__global__ __launch_bounds__(256, 1) void attn_pwg_lab8_kernel(const bf16_t* __restrict__ Q, long q_bstride, const bf16_t* __restrict__ K,
                                                          const bf16_t* __restrict__ Vt, bf16_t* __restrict__ O, long ldo, int H, int Lq,
                                                          int Lk, int Lkp, float* __restrict__ lse, int nqt) {
    // four ring stages [K tile 8 KiB | Vt tile 8 KiB]; the only LDS object of the kernel (the asm statement addresses it by value)
    __shared__ __attribute__((aligned(1024))) unsigned char smem[4 * 16384];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    // XCD-aware flat grid as in attention.hip: every XCD walks whole heads
    int qt, h, b;
    {
        const int nb = gridDim.x, q = nb >> 3, r = nb & 7, x = blockIdx.x & 7, loc = blockIdx.x >> 3;
        const int lid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + loc;
        qt = lid % nqt;
        h = (lid / nqt) % H;
        b = lid / (nqt * H);
    }
    const int q0 = (qt * 4 + wave) * 64;
    const bf16_t* Qh = Q + (long)b * q_bstride + (long)h * Lq * 64;
    const bf16_t* Kh = K + ((long)b * H + h) * (long)Lk * 64;
    const bf16_t* Vh = Vt + ((long)b * H + h) * 64 * (long)Lkp;

    const i32x4 rq = rsrc_words(Qh, (long)Lq * 128), rk = rsrc_words(Kh, (long)Lk * 128), rv = rsrc_words(Vh, 64l * Lkp * 2);
    const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) unsigned char*)smem;
    // fragment address of (row l31 of a 32-row block, 16-byte chunk hi) in a [64][128 B] tile whose chunks are XOR-swizzled by the row
    const unsigned ko0 = lds0 + (unsigned)(l31 * 128 + ((hi ^ ((l31 >> 1) & 7)) << 4));
    // LDS-DMA: a wave-instruction fills 8 tile rows; wave w fills row groups 2w, 2w + 1 of the K tile and of the Vt tile
    unsigned vk[2], vv[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int srow = (wave * 2 + i) * 8 + (lane >> 3);
        const int scol = ((lane & 7) ^ ((srow >> 1) & 7)) * 8;
        vk[i] = (unsigned)((srow * 64 + scol) * 2);
        vv[i] = (unsigned)(((long)srow * Lkp + scol) * 2);
    }
    const unsigned wlds = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)wave * 2048u);
    const unsigned qoff0 = (unsigned)(((q0 + l31) * 64 + hi * 8) * 2), qoff1 = qoff0 + 32 * 128;
    // output rows bounce through the wave's 8-KiB block of the ring (two 32-row blocks), XOR-swizzled by the row
    const unsigned escr = lds0 + (unsigned)(wave * 8192 + l31 * 128 + ((hi ^ (l31 & 7)) << 4));
    const int nt = __builtin_amdgcn_readfirstlane((Lk + 63) >> 6);
    const int rem = __builtin_amdgcn_readfirstlane(Lk & 63);        // valid keys of a ragged last tile (0: whole)
    const unsigned hi4 = (unsigned)(hi * 4);
    float lse0, lse1;
    unsigned dbg0, dbg1, dbg2, dbg3, dbg4;
    asm volatile(
#include "attn_pwg_lab8.inc"
        : [lse0] "=&v"(lse0), [lse1] "=&v"(lse1)
          , [dbg0] "=&v"(dbg0), [dbg1] "=&v"(dbg1), [dbg2] "=&v"(dbg2), [dbg3] "=&v"(dbg3), [dbg4] "=&v"(dbg4)
        : [rq] "s"(rq), [rk] "s"(rk), [rv] "s"(rv), [nt] "s"(nt), [rem] "s"(rem), [hi4] "v"(hi4), [wlds] "s"(wlds), [ko0] "v"(ko0), [vk0] "v"(vk[0]), [vk1] "v"(vk[1]),
          [vv0] "v"(vv[0]), [vv1] "v"(vv[1]), [qoff0] "v"(qoff0), [qoff1] "v"(qoff1), [escr] "v"(escr)
        : "memory", "vcc", "scc",
#include "attn_pwg_clobbers_lab.inc"
    );
    if (lse && blockIdx.x == (gridDim.x >> 1) + 3 && lane == 0) {
        unsigned* d = reinterpret_cast<unsigned*>(lse + (long)gridDim.x / nqt * Lq) + wave * 8;
        d[0] = dbg0, d[1] = dbg1, d[2] = dbg2, d[3] = dbg3, d[4] = dbg4, d[5] = (unsigned)nt;
    }
    // whole 128-byte rows out of the wave's block: 8 rows per store instruction
    const int r8 = lane >> 3, c8 = lane & 7;
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const unsigned char* scr = smem + wave * 8192 + n * 4096;
        bf16_t* obase = O + ((long)b * Lq + q0 + n * 32) * ldo + h * 64 + c8 * 8;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int r = p * 8 + r8;
            const uint4 v = *reinterpret_cast<const uint4*>(scr + r * 128 + ((c8 ^ (r & 7)) << 4));
            if (q0 + n * 32 + r < Lq) *reinterpret_cast<uint4*>(obase + (long)r * ldo) = v;
        }
        const int q = q0 + n * 32 + l31;
        if (lse && q < Lq && hi == 0) lse[((long)b * H + h) * Lq + q] = n ? lse1 : lse0;     // log2-domain LSE
    }
}


}  // namespace

extern "C" int m324_lab_attn_pwg(int variant, const void* Q, long q_bstride, const void* K, const void* Vt, void* O, long ldo, int B, int H,
                                 int Lq, int Lk, float* lse, void* stream) {
    const int Lkp = (Lk + 63) / 64 * 64;
    const int nqt = ceil_div(Lq, 256);
    hipStream_t s = (hipStream_t)stream;
#define PWG_LAUNCH(KERNEL)                                                                                                  \
    hipLaunchKernelGGL(KERNEL, dim3((unsigned)((long)nqt * H * B)), dim3(256), 0, s, (const bf16_t*)Q, q_bstride, (const bf16_t*)K, \
                       (const bf16_t*)Vt, (bf16_t*)O, ldo, H, Lq, Lk, Lkp, lse, nqt)
    switch (variant) {
        case 1: PWG_LAUNCH(attn_pwg_lab1_kernel); break;
        case 2: PWG_LAUNCH(attn_pwg_lab2_kernel); break;
        case 3: PWG_LAUNCH(attn_pwg_lab3_kernel); break;
        case 4: PWG_LAUNCH(attn_pwg_lab4_kernel); break;
        case 5: PWG_LAUNCH(attn_pwg_lab5_kernel); break;
        case 6: PWG_LAUNCH(attn_pwg_lab6_kernel); break;
        case 7: PWG_LAUNCH(attn_pwg_lab7_kernel); break;
        case 8: PWG_LAUNCH(attn_pwg_lab8_kernel); break;
        case 9: PWG_LAUNCH(attn_pwg_lab9_kernel); break;
        default: return -1;
    }
#undef PWG_LAUNCH
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
