// m324_gemm, schedule v14: two workgroups per CU out of phase ("ping-pong"), each a persistent 4-wave 256 x 128 tile kernel.
//
// Why (round 5, tools/ta_lab on the fc1 shape 10368 x 3072 x 768, one MI355X):
//   loads of the whole GEMM alone (LDS-DMA, 256 x 256 tiles) 19-20 us | MFMAs alone 28-29 us | both 32-34 us | the kernel (v10) 60 us
//   loads alone, 256 x 128 tiles (1.5 x the bytes) 30-31 us | with the MFMAs 34-35 us
//   W as fragment-packed global loads straight to VGPRs: the same 75-85 B/ns/CU as LDS-DMA -- the path is bound by bytes from L2,
//   not by the LDS-DMA form, so "weights off the LDS-DMA path" buys nothing.
// i.e. the operand traffic fits under the MFMAs even at 256 x 128; what v10 loses is everything that is NOT the main loop: at K = 768
// a tile is 12 K-stages between a prologue and an epilogue (GELU, LayerNorm fold, q|k|v RMSNorm: VALU work, 5-12 k cycles per tile)
// during which the matrix pipe of that CU idles, because its one workgroup is in lockstep.  Here a CU holds TWO workgroups (4 waves
// = one per SIMD each, <= 256 registers, 80 KiB of LDS each); the second starts late by about one epilogue, so that one's epilogue
// and prologue run beside the other's main loop -- which then has the matrix pipe to itself and runs its K-stages twice as fast.
// Phase is neutral (neither workgroup's tile time depends on where the other's epilogue falls inside its main loop), and a
// workgroup runs only 2-12 tiles per launch, so the offset set at the start is the offset at the end.  Placement (workgroups b and
// b + gridDim.x / 2 on the same CU) is the observed dispatch order, used for speed only: without it the kernel is a plain
// two-workgroups-per-CU tile kernel.
//
// LDS: a ring of FIVE 16-KiB chunks (a chunk = 128 rows x 64 k).  A K-stage is three chunks: Aa (tile rows 0-63 and 128-191: the
// accumulator rows i = 0, 1 of both rows of waves), W (the tile's 128 columns) and Ab (rows 64-127, 192-255: i = 2, 3).  A stage
// runs its MFMAs in two halves -- all four k-steps of i = 0, 1, then of i = 2, 3 (W fragments stay in registers for the second
// half) -- so Aa_s dies in the middle of stage s and the five chunks give every piece one whole stage of look-ahead:
//   chunk c = 3 s + {0: Aa, 1: W, 2: Ab} sits at ring position (c + 3) % 5
//   X_s (Aa_s, W_s landed; W_{s-1}, Ab_{s-1} dead): issue Aa_{s+1}, W_{s+1} during the first half
//   Y_s (Ab_s landed; Aa_s dead):                   issue Ab_{s+1} during the second half
// MFMA work is rotated one phase (8 MFMAs) against the fragment reads, as in v10 / v12: a phase's MFMAs run under the reads of the
// next phase, so both barriers sit where no fragment read is outstanding.  The epilogue scratch lives at ring positions 0-2 and the
// next tile's Aa_0, W_0 (positions 3, 4) are issued in front of the epilogue.
#include "gemm_tile.h"

namespace {

constexpr int CH14 = 128 * ROWB;                  // 16 KiB

template <typename TOUT, int ACT, int RES>
__global__ __launch_bounds__(256, 2) void gemm_pp_kernel(const bf16_t* __restrict__ A, long lda, const bf16_t* __restrict__ W, long ldw,
                                                         TOUT* C, long ldc, int M, int N, int K, Epilogue ep, int ntn, int ntiles,
                                                         int xcd_remap, int skew) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[5 * CH14];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, hi = lane >> 5;
    const int NS = K / 64;
    const int ntm = (M + 255) / 256;

    // LDS-DMA pieces of 8 rows x 128 B; wave w moves pieces 4w .. 4w+3 of every chunk.  Chunk row r of Aa is tile row
    // (r & 63) + 128 (r >> 6), of Ab 64 more; of W tile column r.
    unsigned gaa[4], gab[4], gw[4];
    __amdgpu_buffer_rsrc_t ra, rb;
    int m0 = 0, n0 = 0;
    auto tile_setup = [&](int t) {
        int tm, tn;
        tile_of(t, ntiles, ntm, ntn, xcd_remap, tm, tn);
        m0 = tm * 256;
        n0 = tn * 128;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = (wave * 4 + i) * 8 + (lane >> 3);
            const int c = ((lane & 7) ^ ((r >> 1) & 7)) * 8;
            const int ta = (r & 63) + 128 * (r >> 6);
            gaa[i] = (unsigned)(((long)min(ta, M - 1 - m0) * lda + c) * 2);
            gab[i] = (unsigned)(((long)min(ta + 64, M - 1 - m0) * lda + c) * 2);
            gw[i] = (unsigned)(((long)min(r, N - 1 - n0) * ldw + c) * 2);
        }
        ra = dma_rsrc(A + (long)m0 * lda);
        rb = dma_rsrc(W + (long)n0 * ldw);
    };
    // pieces i0, i0 + 1 of a chunk (which: 0 Aa, 1 W, 2 Ab) of K-stage st into ring position pos
    const __amdgpu_buffer_rsrc_t rnone = dma_rsrc_none(A);    // look-ahead pieces past the end of K: gemm_tile.h dma_rsrc_none
    auto issue2 = [&](int which, int i0, int st, int pos, bool live = true) {
        unsigned char* d = smem + pos * CH14 + wave * 4096 + i0 * 1024;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const unsigned g = which == 0 ? gaa[i0 + i] : (which == 1 ? gw[i0 + i] : gab[i0 + i]);
            dma_piece(live ? (which == 1 ? rb : ra) : rnone, d + i * 1024, g, (unsigned)(st * 128));
        }
    };

    f32x16 acc[4][2];
    const int aoff = lds_off(wm * 64 + l31, hi), boff = lds_off(wn * 64 + l31, hi);
    bf16x8 fa[2][2][2];                                      // [set][k-step of the phase][ii]
    bf16x8 fw[4][2];                                         // [k-step of the stage][j]: kept for the second half of the stage
    // fragment reads of phase ph (0, 1: half a from chunk pa; 2, 3: half b from chunk pb): A of k-steps 2 (ph & 1), + 1 into set
    // ph & 1; phases 0, 1 also read W of those k-steps
    auto load_frags = [&](int ph, int pa, int pw) {
        const unsigned char* ba = smem + pa * CH14;
        const unsigned char* bw = smem + pw * CH14;
        const int set = ph & 1;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int ks = 2 * (ph & 1) + q, x = ks << 5;
            if (ph < 2) {
#pragma unroll
                for (int j = 0; j < 2; ++j) fw[ks][j] = *reinterpret_cast<const bf16x8*>(bw + ((boff + j * 4096) ^ x));
            }
#pragma unroll
            for (int ii = 0; ii < 2; ++ii) fa[set][q][ii] = *reinterpret_cast<const bf16x8*>(ba + ((aoff + ii * 4096) ^ x));
        }
    };
    auto mma8 = [&](int ph) {                                // the 8 MFMAs of phase ph (fragments of set ph & 1)
        const int set = ph & 1, ib = ph < 2 ? 0 : 2;
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[ib + ii][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[2 * (ph & 1) + q][j], fa[set][q][ii], acc[ib + ii][j], 0, 0, 0);
    };
#define M324_SG(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
    auto sched_a = [&]() {                                   // 8 MFMAs, 8 fragment reads, 4 LDS-DMA pieces (first half of a stage)
        M324_SG(0x008, 1); M324_SG(0x100, 2); M324_SG(0x008, 1); M324_SG(0x100, 2);
        M324_SG(0x008, 1); M324_SG(0x100, 2); M324_SG(0x008, 1); M324_SG(0x100, 2);
        M324_SG(0x008, 1); M324_SG(0x020, 1); M324_SG(0x008, 1); M324_SG(0x020, 1);
        M324_SG(0x008, 1); M324_SG(0x020, 1); M324_SG(0x008, 1); M324_SG(0x020, 1);
    };
    auto sched_b = [&]() {                                   // 8 MFMAs, 4 fragment reads, 2 LDS-DMA pieces (second half)
        M324_SG(0x008, 1); M324_SG(0x100, 1); M324_SG(0x008, 1); M324_SG(0x100, 1);
        M324_SG(0x008, 1); M324_SG(0x100, 1); M324_SG(0x008, 1); M324_SG(0x100, 1);
        M324_SG(0x008, 1); M324_SG(0x020, 1); M324_SG(0x008, 1); M324_SG(0x020, 1);
        M324_SG(0x008, 2);
    };

    tile_setup(blockIdx.x);
    // first tile: Aa_0, W_0 (positions 3, 4), then Ab_0 (position 0)
    issue2(0, 0, 0, 3); issue2(0, 2, 0, 3);
    issue2(1, 0, 0, 4); issue2(1, 2, 0, 4);
    issue2(2, 0, 0, 0); issue2(2, 2, 0, 0);
    // the second workgroup of a CU starts late by about one epilogue (`skew` units of 1024 cycles): see the file header
    if (skew > 0 && (int)blockIdx.x >= (int)(gridDim.x >> 1)) {
        for (int i = 0; i < skew; ++i) __builtin_amdgcn_s_sleep(16);
    }
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int mt = m0, nt = n0;
        const bool more = t + (int)gridDim.x < ntiles;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int ii = 0; ii < 2; ++ii) fa[1][q][ii] = (bf16x8)(0);
#pragma unroll
        for (int ks = 2; ks < 4; ++ks)
#pragma unroll
            for (int j = 0; j < 2; ++j) fw[ks][j] = (bf16x8)(0);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");      // X_0: Aa_0, W_0 landed (Ab_0 may fly)
        M324_BARRIER();
        int pa = 3;                                           // ring position of Aa_s; W_s at pa + 1, Ab_s at pa + 2 (mod 5)
        for (int s = 0; s < NS; ++s) {
            int pw = pa + 1, pb = pa + 2, pan = pa + 3, pwn = pa + 4;      // ... of Aa_{s+1}, W_{s+1}; Ab_{s+1} takes Aa_s's place
            pw = pw >= 5 ? pw - 5 : pw;
            pb = pb >= 5 ? pb - 5 : pb;
            pan = pan >= 5 ? pan - 5 : pan;
            pwn = pwn >= 5 ? pwn - 5 : pwn;
            const int sn = s + 1 < NS ? s + 1 : NS - 1;       // past the end: counted and written, not fetched
            const bool live = s + 1 < NS;
            load_frags(0, pa, pw);
            issue2(0, 0, sn, pan, live); issue2(1, 0, sn, pwn, live);
            mma8(3);                                          // (s-1, second half, k-steps 2, 3); zeros in a tile's first stage
            sched_a();
            load_frags(1, pa, pw);
            issue2(0, 2, sn, pan, live); issue2(1, 2, sn, pwn, live);
            mma8(0);
            sched_a();
            asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");    // Y_s: Ab_s landed, Aa_s is in registers
            M324_BARRIER();
            load_frags(2, pb, pw);
            issue2(2, 0, sn, pa, live);
            mma8(1);
            sched_b();
            load_frags(3, pb, pw);
            issue2(2, 2, sn, pa, live);
            mma8(2);
            sched_b();
            asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");    // X_{s+1}: Aa_{s+1}, W_{s+1} landed; stage s is in registers
            M324_BARRIER();
            pa = pan;
        }
        mma8(3);                                              // (NS-1, second half, k-steps 2, 3)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // no LDS-DMA may outlive the main loop: the ring becomes scratch
        M324_BARRIER();
        if (more) {                                           // the next tile's first chunks land under this tile's epilogue
            tile_setup(t + gridDim.x);
            issue2(0, 0, 0, 3); issue2(0, 2, 0, 3);
            issue2(1, 0, 0, 4); issue2(1, 2, 0, 4);
        }
        store_tile_lds<TOUT, ACT, RES, 4>(acc, reinterpret_cast<float*>(smem) + wave * ep_wave_floats(ACT), C, ldc, M, N, mt + wm * 128,
                                          nt + wn * 64, lane, ep);
        if (more) {
            // everything this wave has in flight -- the prefetched chunks and the epilogue's stores, which share vmcnt and may
            // retire out of order with respect to each other -- must be done before the scratch becomes ring again (the builtin:
            // hipcc's own wait-count pass must see the drain, gemm_ring4.hip)
            __builtin_amdgcn_s_waitcnt(0x0F70);               // vmcnt(0)
            M324_BARRIER();
            issue2(2, 0, 0, 0); issue2(2, 2, 0, 0);           // Ab_0 into the scratch's first chunk
        }
    }
#undef M324_SG
}

}  // namespace

namespace m324 {

int launch_pp(const m324_gemm_args* a, hipStream_t s, const Epilogue& ep, int act_code, int res_code, int xcd_remap) {
    const int ntn = ceil_div(a->N, 128), ntiles = ntn * ceil_div(a->M, 256);
    static const int n_cu = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
            n = 256;
        return n - (n & 7);                                  // a multiple of 8: tile t and t + grid share an XCD
    }();
    const int slots = 2 * n_cu;
    const dim3 grid(ntiles < slots ? ntiles : slots);
    // start offset of a CU's second workgroup, in units of 1024 cycles: about one epilogue (M324_PP_SKEW overrides; -1 = none)
    int skew = tunable(TUN_PP_SKEW);
    if (skew == 0) skew = (act_code & 7) == 1 || (act_code & 7) == 2 || (act_code & 7) == 4 ? 7 : 4;
    if (skew < 0 || (int)grid.x < slots) skew = 0;
#define M324_PP(TOUT, ACT, RES)                                                                                         \
    hipLaunchKernelGGL((gemm_pp_kernel<TOUT, ACT, RES>), grid, dim3(256), 0, s, (const bf16_t*)a->A, a->lda,            \
                       (const bf16_t*)a->W, a->ldw, (TOUT*)a->C, a->ldc, a->M, a->N, a->K, ep, ntn, ntiles, xcd_remap, skew)
    // bf16 outputs without residual only: what the chooser sends here (wide projections and MLP hidden layers, forward and
    // training); the other epilogues stay on the round-4 schedules
    if (a->out_dtype != M324_BF16) M324_FAIL(M324_ERR_UNSUPPORTED, "m324_gemm: the ping-pong kernel writes bf16 outputs");
    const int key = act_code * 4 + res_code;
    switch (key) {
        case 0 * 4 + 0: M324_PP(bf16_t, 0, 0); break;
        case 1 * 4 + 0: M324_PP(bf16_t, 1, 0); break;
        case 2 * 4 + 0: M324_PP(bf16_t, 2, 0); break;      // training: GELU + stored pre-activation
        case 3 * 4 + 0: M324_PP(bf16_t, 3, 0); break;      // training: x gelu'(pre-activation)
        case 4 * 4 + 0: M324_PP(bf16_t, 4, 0); break;      // head-major q|k|v
        case 8 * 4 + 0: M324_PP(bf16_t, 8, 0); break;      // LayerNorm-fold consumers (ACT | 8)
        case 9 * 4 + 0: M324_PP(bf16_t, 9, 0); break;
        case 12 * 4 + 0: M324_PP(bf16_t, 12, 0); break;
        default:
            M324_FAIL(M324_ERR_UNSUPPORTED, "m324_gemm: no ping-pong kernel for epilogue act=%d res=%d", act_code, res_code);
    }
#undef M324_PP
    return M324_OK;
}

int pp_grid(const m324_gemm_args* a) {
    static const int n_cu = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
            n = 256;
        return n - (n & 7);
    }();
    const int ntiles = ceil_div(a->N, 128) * ceil_div(a->M, 256);
    return ntiles < 2 * n_cu ? ntiles : 2 * n_cu;
}

}  // namespace m324
