// Lab twin of motion324_amd/csrc/gemm_hp_kernel.inl (tools/lab_src/hp_lab.hip): the product wrapper plus HP_TRACE = 1 (the stamped stream's
// debug outputs) and HP_TRACE = 2 (every tile of a workgroup lands on the same 64 KiB: no output traffic; wrong results).
__global__ __launch_bounds__(256, 1) void HP_KERNEL(const bf16_t* __restrict__ A, long lda, const bf16_t* __restrict__ W, long ldw, bf16_t* C, long ldc,
                                                    int M, int N, const float* __restrict__ bias, const float* __restrict__ colsum,
                                                    const float2* __restrict__ rowstat, int ntn, int ntiles, int xcd_remap, unsigned* trace) {
    // three ring buffers [X 256 rows x 128 B | W 128 rows x 128 B] + the workgroup's tile table + 2 KiB of store scratch per wave; the only
    // LDS object of the kernel
    __shared__ __attribute__((aligned(1024))) unsigned char smem[3 * HP_STAGE + HP_TABLE_BYTES + 4 * HP_SCRATCH];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, hi = lane >> 5;
    const int ntm = (M + 255) / 256;
    const int G = gridDim.x;
    const int nt_mine = (ntiles - (int)blockIdx.x + G - 1) / G;          // >= 1: the host launches no more workgroups than tiles

    // tile table: entry 0 = the null tile in front of the first one (C / row-statistics record counts 0: stores dropped, loads return 0),
    // entries 1 .. n = this workgroup's tiles in order, entry n + 1 = the null tile behind the last (X record count 0: LDS-DMA pieces
    // that fetch nothing).  12 words: X lo, X hi, X records, W lo | W hi, C lo, C hi, C records | column byte offset, rowstat lo, hi, records
    unsigned* const tab = reinterpret_cast<unsigned*>(smem + 3 * HP_STAGE);
    for (int e = tid; e < nt_mine + 2; e += 256) {
        const bool real = e >= 1 && e <= nt_mine;
        const int t = (int)blockIdx.x + (real ? e - 1 : 0) * G;
        int tm, tn;
        tile_of(t, ntiles, ntm, ntn, xcd_remap, tm, tn);
        const long mrem = min(256, M - tm * 256);
        const unsigned long xa = (unsigned long)(A + (long)tm * 256 * lda), wa = (unsigned long)(W + (long)tn * 128 * ldw);
#if HP_TRACE == 2          // lab: every tile of a workgroup lands on the same 64 KiB (no output traffic to HBM; wrong results)
        const unsigned long ca = (unsigned long)(C + (long)(blockIdx.x % ntm) * 256 * ldc + (blockIdx.x / ntm % ntn) * 128);
#else
        const unsigned long ca = (unsigned long)(C + (long)tm * 256 * ldc + tn * 128);
#endif
        const unsigned long ra = (unsigned long)(rowstat + (long)tm * 256);
        unsigned* o = tab + e * 12;
        o[0] = (unsigned)xa;
        o[1] = (unsigned)(xa >> 32) & 0xffffu;
        o[2] = real ? (unsigned)(mrem * lda * 2) : 0u;
        o[3] = (unsigned)wa;
        o[4] = (unsigned)(wa >> 32) & 0xffffu;
        o[5] = (unsigned)ca;
        o[6] = (unsigned)(ca >> 32) & 0xffffu;
        o[7] = real ? (unsigned)((mrem - 1) * ldc * 2 + 256) : 0u;
        o[8] = (unsigned)(tn * 128 * 4);
        o[9] = (unsigned)ra;
        o[10] = (unsigned)(ra >> 32) & 0xffffu;
        o[11] = (real && rowstat) ? (unsigned)(mrem * 8) : 0u;
    }
    const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) unsigned char*)smem;
    // LDS-DMA: a wave-instruction fills 8 rows of 128 bytes; wave w fills X rows 64 w .. 64 w + 63 (8 pieces) and W rows 32 w .. (4 pieces).
    // The 16-byte chunks of a row are XOR-swizzled by the row (gemm_tile.h lds_off); piece i = piece (i & 1) + (i >> 1) * 16 rows.
    unsigned xo[2], wo[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int xr = wave * 64 + i * 8 + (lane >> 3), wr = wave * 32 + i * 8 + (lane >> 3);
        xo[i] = (unsigned)(((long)xr * lda + (((lane & 7) ^ ((xr >> 1) & 7)) * 8)) * 2);
        wo[i] = (unsigned)(((long)wr * ldw + (((lane & 7) ^ ((wr >> 1) & 7)) * 8)) * 2);
    }
    const unsigned fb = (unsigned)(l31 * 128 + ((hi ^ ((l31 >> 1) & 7)) << 4));
    // stores: a 32 x 64 block of the wave leaves through its 16-row x 128-byte scratch in two halves (rows l31 < 16, then the rest).  A lane
    // writes its row's chunk pairs k = 0..3 (chunk 2 k + hi, XOR-swizzled by the row) and reads back chunk L & 7 of row L >> 3 (+ 8):
    // a store instruction then writes 8 rows x 128 contiguous bytes.
    const unsigned scr = lds0 + 3 * HP_STAGE + HP_TABLE_BYTES + (unsigned)wave * HP_SCRATCH;
    const unsigned wa0 = scr + (unsigned)((l31 & 15) * 128 + ((hi ^ (l31 & 7)) << 4));
    const unsigned rda = scr + (unsigned)((lane >> 3) * 128 + (((lane & 7) ^ ((lane >> 3) & 7)) << 4));
    const unsigned svo = (unsigned)((((long)(wm * 128 + (lane >> 3))) * ldc + wn * 64) * 2 + (lane & 7) * 16);
    const unsigned bo = (unsigned)((wn * 64 + 4 * hi) * 4), ro = (unsigned)((wm * 128 + l31) * 8);
    const unsigned tabv = lds0 + 3 * HP_STAGE;
    const unsigned wldsx = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)wave * 8192u);
    const unsigned wldsw = __builtin_amdgcn_readfirstlane(lds0 + 32768u + (unsigned)wave * 4096u);
    const unsigned wmo = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)wm * 16384u);
    const unsigned wno = __builtin_amdgcn_readfirstlane(lds0 + 32768u + (unsigned)wn * 8192u);
    const unsigned xs16 = __builtin_amdgcn_readfirstlane((unsigned)(lda * 32)), ws16 = __builtin_amdgcn_readfirstlane((unsigned)(ldw * 32));
    const unsigned cs8 = __builtin_amdgcn_readfirstlane((unsigned)(ldc * 16));
    const int ntl = __builtin_amdgcn_readfirstlane(nt_mine);
    // no bias / no colsum: a resource without records -- the loads return 0 and fetch nothing
    const i32x4 rb = rsrc_words(bias ? (const void*)bias : (const void*)A, bias ? (long)N * 4 : 0l);
    const i32x4 rcs = rsrc_words(colsum ? (const void*)colsum : (const void*)A, colsum ? (long)N * 4 : 0l);
    __syncthreads();
#if HP_TRACE == 1
    unsigned dbg0, dbg1, dbg2, dbg3;
#endif
    asm volatile(
#include HP_ASM_INC
#if HP_TRACE == 1
        : [dbg0] "=&v"(dbg0), [dbg1] "=&v"(dbg1), [dbg2] "=&v"(dbg2), [dbg3] "=&v"(dbg3)
#else
        :
#endif
        : [rb] "s"(rb), [rcs] "s"(rcs), [wldsx] "s"(wldsx), [wldsw] "s"(wldsw), [wmo] "s"(wmo), [wno] "s"(wno), [xs16] "s"(xs16), [ws16] "s"(ws16),
          [cs8] "s"(cs8), [ntl] "s"(ntl), [xo0] "v"(xo[0]), [xo1] "v"(xo[1]), [wo0] "v"(wo[0]), [wo1] "v"(wo[1]), [fb] "v"(fb), [wa0] "v"(wa0),
          [rda] "v"(rda), [svo] "v"(svo), [bo] "v"(bo), [ro] "v"(ro), [tab] "v"(tabv)
        : "memory", "vcc", "scc",
#include "gemm_hp_clobbers.inc"
    );
#if HP_TRACE == 1
    if (trace && tid == 0) {
        unsigned* o = trace + (long)blockIdx.x * 4;
        o[0] = dbg0, o[1] = dbg1, o[2] = dbg2, o[3] = dbg3;
    }
#endif
}
