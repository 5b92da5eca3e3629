// attn_pwg_kernel: the long-sequence attention forward of m324_attention (bf16, head_dim 64, pre-scaled Q, transposed
// key-permuted Vt, Lk % 64 == 0) as ONE wave per SIMD with a hand-placed instruction stream.
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

__global__ __launch_bounds__(256, 1) void attn_pwg_kernel(const bf16_t* __restrict__ Q, long q_bstride, const bf16_t* __restrict__ K,
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
    const int nt = __builtin_amdgcn_readfirstlane(Lk >> 6);
    float lse0, lse1;
    asm volatile(
#include "attn_pwg_asm.inc"
        : [lse0] "=&v"(lse0), [lse1] "=&v"(lse1)
        : [rq] "s"(rq), [rk] "s"(rk), [rv] "s"(rv), [nt] "s"(nt), [wlds] "s"(wlds), [ko0] "v"(ko0), [vk0] "v"(vk[0]), [vk1] "v"(vk[1]),
          [vv0] "v"(vv[0]), [vv1] "v"(vv[1]), [qoff0] "v"(qoff0), [qoff1] "v"(qoff1), [escr] "v"(escr)
        : "memory", "vcc", "scc",
#include "attn_pwg_clobbers.inc"
    );
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

// Called by m324_attention's chooser (attention.hip); returns the launch status through hipGetLastError there.
void m324_attn_pwg_launch(const void* Q, long q_bstride, const void* K, const void* Vt, void* O, long ldo, int B, int H, int Lq, int Lk,
                          float* lse, hipStream_t s) {
    const int Lkp = (Lk + 63) / 64 * 64;
    const int nqt = ceil_div(Lq, 256);
    hipLaunchKernelGGL(attn_pwg_kernel, dim3((unsigned)((long)nqt * H * B)), dim3(256), 0, s, (const bf16_t*)Q, q_bstride,
                       (const bf16_t*)K, (const bf16_t*)Vt, (bf16_t*)O, ldo, H, Lq, Lk, Lkp, lse, nqt);
}
