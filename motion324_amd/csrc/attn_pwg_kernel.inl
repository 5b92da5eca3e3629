// The kernel body of attention_pwg.hip; included once per instruction stream (PWG_KERNEL = kernel name, PWG_ASM_INC = the
// generated stream: attn_pwg_asm.inc for the product; tools/lab_src/attention_pwg_lab.hip includes it with the timing-only ablation streams).
__global__ __launch_bounds__(256, 1) void PWG_KERNEL(const bf16_t* __restrict__ Q, long q_bstride, const bf16_t* __restrict__ K,
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
    asm volatile(
#include PWG_ASM_INC
        : [lse0] "=&v"(lse0), [lse1] "=&v"(lse1)
        : [rq] "s"(rq), [rk] "s"(rk), [rv] "s"(rv), [nt] "s"(nt), [rem] "s"(rem), [hi4] "v"(hi4), [wlds] "s"(wlds), [ko0] "v"(ko0), [vk0] "v"(vk[0]), [vk1] "v"(vk[1]),
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

