// m324_gemm, schedules v11 (4-wave persistent 256 x 256 chunk ring) and v12 (4-wave 256 x 128, three-stage ring).  Own
// translation unit because v11 must be compiled
// WITHOUT -amdgpu-mfma-vgpr-form: a wave owns a 128 x 128 block = 256 accumulator registers, which fill the AGPR half
// of the unified file while fragments, addresses and the epilogue use the VGPR half (400 registers, no spills); with
// the flag hipcc keeps the accumulators in VGPRs and shuffles everything else through v_accvgpr_mov.
#include "gemm_tile.h"

namespace {

// ------------------------------------------------------------------------------------------------
// v11: v10's chunk ring (see gemm.hip) with FOUR waves (one per SIMD), each a 128 x 128 block = 4 x 4 accumulators
// (256 registers of the unified 512-entry file), as a PERSISTENT kernel: one workgroup per CU walks over tiles
// blockIdx.x, blockIdx.x + gridDim.x, ...
//  * Per K-stage the workgroup reads 128 KiB of fragments out of LDS instead of 192 KiB and the texture path sees 4
//    instruction streams instead of 8 (tools/dma_lab: 29 instead of 33 cycles per KiB).  On this chip that matters
//    through POWER: tools/clk_lab shows the MFMA stream alone pulls the shader clock down to ~1.78 GHz and the
//    instrumented main loops run at 1.5-1.6 GHz -- every LDS / texture byte saved is clock regained.
//  * At K = 768 a tile is 12 K-stages (~19 us) between a ~3.3 us prologue (cold first loads) and a ~3.5 us epilogue.
//    The tile loop issues the first three chunks (A_0, W_0, A_1) of the NEXT tile right after the last fragment read
//    of this one, so they land under the epilogue (whose LDS scratch lives in chunks 3-4).  Loads and stores of one
//    wave share vmcnt and may retire out of order with respect to each other, so the top of the next tile waits for
//    vmcnt(0): everything this wave has in flight.
// Same stage / phase structure as v10; a phase is 16 MFMAs, 8 fragment reads and 4 LDS-DMA pieces.
template <typename TOUT, int ACT, int RES>
__global__ __launch_bounds__(256) void gemm_ring4_kernel(const bf16_t* __restrict__ A, long lda, const bf16_t* __restrict__ W,
                                                         long ldw, TOUT* C, long ldc, int M, int N, int K, Epilogue ep, int ntn,
                                                         int ntiles, int xcd_remap) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[5 * CHUNK10];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, hi = lane >> 5;
    const int NS = K / 64;

    // LDS-DMA: a wave-instruction fills 8 rows x 128 B; wave w moves row groups 8w .. 8w+7 of A and of W
    unsigned ga[8], gb[8];                                  // byte offsets inside the tile's row panels (buffer-form LDS-DMA)
    __amdgpu_buffer_rsrc_t ra, rb;
    int m0 = 0, n0 = 0;
    auto tile_setup = [&](int t) {
        int tm, tn;
        tile_of(t, ntiles, (M + BM5 - 1) / BM5, ntn, xcd_remap, tm, tn);
        m0 = tm * BM5;
        n0 = tn * BN5;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int r = (wave * 8 + i) * 8 + (lane >> 3);
            const int c = ((lane & 7) ^ ((r >> 1) & 7)) * 8;
            ga[i] = (unsigned)(((long)min(r, M - 1 - m0) * lda + c) * 2);
            gb[i] = (unsigned)(((long)min(r, N - 1 - n0) * ldw + c) * 2);
        }
        ra = dma_rsrc(A + (long)m0 * lda);
        rb = dma_rsrc(W + (long)n0 * ldw);
    };
    const __amdgpu_buffer_rsrc_t rnone = dma_rsrc_none(A);    // look-ahead pieces past the end of K: gemm_tile.h dma_rsrc_none
    const bool refetch = (xcd_remap & 8) != 0;
    auto issue4 = [&](const unsigned (&g)[8], int i0, int st, int pos, bool live = true) {
        unsigned char* d = smem + pos * CHUNK10 + wave * 8192 + i0 * 1024;
        const bool isa = &g[0] == &ga[0];
#pragma unroll
        for (int i = 0; i < 4; ++i) dma_piece(live ? (isa ? ra : rb) : rnone, d + i * 1024, g[i0 + i], (unsigned)(st * 128));
    };
    auto issue_prologue = [&]() {                           // A_0, W_0, A_1 -> chunks 0, 1, 2
        issue4(ga, 0, 0, 0); issue4(ga, 4, 0, 0);
        issue4(gb, 0, 0, 1); issue4(gb, 4, 0, 1);
        const int s1 = NS > 1 ? 1 : 0;
        issue4(ga, 0, s1, 2); issue4(ga, 4, s1, 2);
    };

    const int aoff = lds_off(wm * 128 + l31, hi), boff = lds_off(wn * 128 + l31, hi);
    bf16x8 fa[2][4], fb[2][4];
    f32x16 acc[2][4][2];                                    // [column half][i][j]: the epilogue works on 128 x 64 halves
    auto load_frags = [&](int set, int pa, int pw, int ks) {
        const unsigned char* ba = smem + pa * CHUNK10;
        const unsigned char* bw = smem + pw * CHUNK10;
        const int x = ks << 5;
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[set][j] = *reinterpret_cast<const bf16x8*>(bw + ((boff + j * 4096) ^ x));
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[set][i] = *reinterpret_cast<const bf16x8*>(ba + ((aoff + i * 4096) ^ x));
    };
    auto mma16 = [&](int set) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[j >> 1][i][j & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[set][j], fa[set][i], acc[j >> 1][i][j & 1], 0, 0, 0);
    };
#define M324_SG(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
    auto sched_phase = [&]() {
        M324_SG(0x008, 1); M324_SG(0x100, 1); M324_SG(0x008, 1); M324_SG(0x100, 1);
        M324_SG(0x008, 1); M324_SG(0x100, 1); M324_SG(0x008, 1); M324_SG(0x100, 1);
        M324_SG(0x008, 1); M324_SG(0x100, 1); M324_SG(0x008, 1); M324_SG(0x100, 1);
        M324_SG(0x008, 1); M324_SG(0x100, 1); M324_SG(0x008, 1); M324_SG(0x100, 1);
        M324_SG(0x008, 1); M324_SG(0x020, 1); M324_SG(0x008, 1); M324_SG(0x020, 1);
        M324_SG(0x008, 1); M324_SG(0x020, 1); M324_SG(0x008, 1); M324_SG(0x020, 1);
        M324_SG(0x008, 4);
    };

    tile_setup(blockIdx.x);
    issue_prologue();
    {                                                       // first tile: fill the whole ring (W_1, A_2 -> chunks 3, 4)
        const int s1 = NS > 1 ? 1 : 0, s2 = NS > 2 ? 2 : NS - 1;
        issue4(gb, 0, s1, 3); issue4(gb, 4, s1, 3);
        issue4(ga, 0, s2, 4); issue4(ga, 4, s2, 4);
    }
    int pa = 0, pw = 1;
    auto stage = [&](int s, auto issue_tag) {
        constexpr bool ISSUE = decltype(issue_tag)::value;
        int pwn = pa + 3, pan = pa + 4;
        pwn = pwn >= 5 ? pwn - 5 : pwn;
        pan = pan >= 5 ? pan - 5 : pan;
        const int sw = s + 1 < NS ? s + 1 : NS - 1, sa = s + 2 < NS ? s + 2 : NS - 1;
        const bool wl = s + 1 < NS || refetch, al = s + 2 < NS || refetch;
        load_frags(0, pa, pw, 0);
        if constexpr (ISSUE) issue4(gb, 0, sw, pwn, wl);
        mma16(1);                                           // (s-1, k-step 3); zeros in the first iteration
        sched_phase();
        load_frags(1, pa, pw, 1);
        if constexpr (ISSUE) issue4(gb, 4, sw, pwn, wl);
        mma16(0);
        sched_phase();
        load_frags(0, pa, pw, 2);
        if constexpr (ISSUE) issue4(ga, 0, sa, pan, al);
        mma16(1);
        sched_phase();
        load_frags(1, pa, pw, 3);
        if constexpr (ISSUE) issue4(ga, 4, sa, pan, al);
        mma16(0);
        sched_phase();
        asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
        M324_BARRIER();
        pa = pa + 2 >= 5 ? pa - 3 : pa + 2;
        pw = pw + 2 >= 5 ? pw - 3 : pw + 2;
    };
    asm volatile("s_waitcnt vmcnt(24)" ::: "memory");      // first tile: stage 0 landed (A_1, W_1, A_2 may fly)
    M324_BARRIER();
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int mt = m0, nt = n0;                          // this tile's origin (m0 / n0 move on before the epilogue)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[h][i][j][r] = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) { fa[1][i] = (bf16x8)(0); fb[1][i] = (bf16x8)(0); }
        pa = 0, pw = 1;
        // the first tile found the whole ring issued (stage 0 has nothing to add); later tiles had three chunks
        // prefetched under the previous epilogue, whose scratch occupied chunks 3-4
        if (t == (int)blockIdx.x) stage(0, std::false_type{});
        else stage(0, std::true_type{});
        for (int s = 1; s < NS; ++s) stage(s, std::true_type{});
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the re-fetched chunks too: the ring is about to be reused
        M324_BARRIER();
        if (t + (int)gridDim.x < ntiles) {                   // next tile's first chunks land under this tile's epilogue
            tile_setup(t + gridDim.x);
            issue_prologue();
        }
        mma16(1);                                           // (NS-1, k-step 3)
        float* scr = reinterpret_cast<float*>(smem + 3 * CHUNK10) + wave * ep_wave_floats(ACT);
        store_tile_lds<TOUT, ACT, RES, 4>(acc[0], scr, C, ldc, M, N, mt + wm * 128, nt + wn * 128, lane, ep);
        store_tile_lds<TOUT, ACT, RES, 4>(acc[1], scr, C, ldc, M, N, mt + wm * 128, nt + wn * 128 + 64, lane, ep);
        if (t + (int)gridDim.x < ntiles) {
            // Before the next tile starts, everything this wave has in flight must be done: the three prefetched chunks
            // and the epilogue's stores (loads and stores share vmcnt and may retire out of order with respect to each
            // other).  The builtin, not inline asm: hipcc's own wait-count pass must see the drain, or it protects the
            // epilogue's bias loads against the first fragment read of every K-stage with a vmcnt(0) of its own.
            __builtin_amdgcn_s_waitcnt(0x0F70);             // vmcnt(0)
            M324_BARRIER();
        }
    }
#undef M324_SG
}

// ------------------------------------------------------------------------------------------------
// v12: 256 x 128 tiles for the outputs that are only 768 columns wide (every attention / MLP output projection of the
// trunk and of DINO: 28 + 28 launches per clip).  256 x 256 tiles give those 123 workgroups for 256 CUs; the 128 x 128
// tiles of v2 fill the chip but move 64 KiB through the texture path per 4.2 MFLOP and are bound by it (tools/dma_lab:
// 29-33 cycles per KiB -> 0.87 us per K-stage against 0.28 us of MFMA).  A 256 x 128 tile moves 48 KiB for the same
// work, 41 x 6 = 246 tiles cover the chip in one round, and a K-stage is 48 KiB of LDS, so THREE whole stages fit
// (144 KiB): two stages of look-ahead instead of v10's 1.5.  Four waves (one per SIMD) as 2 (M) x 2 (N), each a
// 128 x 64 block = 4 x 2 accumulators; a phase (k-step of 16) is 8 MFMAs, 6 fragment reads and 3 LDS-DMA pieces.
// MFMA work is rotated by one k-step against the LDS stages exactly as in v10:
//   X_{s-1}: stage s landed, slot of stage s-1 free -> iteration s streams stage s+2 into it
//   phases: MFMA (s-1,3) | (s,0) | (s,1) | (s,2), each under the fragment reads of the next k-step
//   lgkmcnt(0) (stage s is in registers), vmcnt(12) (stage s+1 landed; the 12 pieces of s+2 may fly), X_s.
constexpr int STAGE12 = (256 + 128) * ROWB;       // 48 KiB

template <typename TOUT, int ACT, int RES>
__global__ __launch_bounds__(256) void gemm_ring3_kernel(const bf16_t* __restrict__ A, long lda, const bf16_t* __restrict__ W,
                                                         long ldw, TOUT* C, long ldc, int M, int N, int K, Epilogue ep, int ntn,
                                                         int xcd_remap) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[3 * STAGE12];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, hi = lane >> 5;
    int tm, tn;
    tile_of(blockIdx.x, gridDim.x, (M + 256 - 1) / 256, ntn, xcd_remap, tm, tn);
    const int m0 = tm * 256, n0 = tn * 128;

    // LDS-DMA pieces of 8 rows x 128 B: wave w moves row groups 8w .. 8w+7 of A (32 groups) and 4w .. 4w+3 of W (16 groups)
    unsigned g[12];                                         // byte offsets inside the tile's row panels (buffer-form LDS-DMA)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int r = (wave * 8 + i) * 8 + (lane >> 3);
        g[i] = (unsigned)(((long)min(r, M - 1 - m0) * lda + ((lane & 7) ^ ((r >> 1) & 7)) * 8) * 2);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (wave * 4 + i) * 8 + (lane >> 3);
        g[8 + i] = (unsigned)(((long)min(r, N - 1 - n0) * ldw + ((lane & 7) ^ ((r >> 1) & 7)) * 8) * 2);
    }
    const __amdgpu_buffer_rsrc_t ra = dma_rsrc(A + (long)m0 * lda), rb = dma_rsrc(W + (long)n0 * ldw);
    const __amdgpu_buffer_rsrc_t rnone = dma_rsrc_none(A);    // look-ahead pieces past the end of K: gemm_tile.h dma_rsrc_none
    const bool refetch = (xcd_remap & 8) != 0;
    auto issue3 = [&](int i0, int st, int slot, bool live = true) {           // pieces i0 .. i0+2 of stage st
        unsigned char* base = smem + slot * STAGE12;
#pragma unroll
        for (int i = i0; i < i0 + 3; ++i) {
            unsigned char* d = i < 8 ? base + wave * 8192 + i * 1024 : base + 32768 + wave * 4096 + (i - 8) * 1024;
            dma_piece(live ? (i < 8 ? ra : rb) : rnone, d, g[i], (unsigned)(st * 128));
        }
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int NS = K / 64;
    const int aoff = lds_off(wm * 128 + l31, hi), boff = 32768 + lds_off(wn * 64 + l31, hi);
    bf16x8 fa[2][4], fb[2][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[1][i] = (bf16x8)(0);
#pragma unroll
    for (int j = 0; j < 2; ++j) fb[1][j] = (bf16x8)(0);
    auto load_frags = [&](int set, int slot, int ks) {
        const unsigned char* base = smem + slot * STAGE12;
        const int x = ks << 5;
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[set][j] = *reinterpret_cast<const bf16x8*>(base + ((boff + j * 4096) ^ x));
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[set][i] = *reinterpret_cast<const bf16x8*>(base + ((aoff + i * 4096) ^ x));
    };
    auto mma8 = [&](int set) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[set][j], fa[set][i], acc[i][j], 0, 0, 0);
    };
#define M324_SG(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
    auto sched_phase = [&]() {                               // M r M r M r M r M r M r M G M G G
        M324_SG(0x008, 1); M324_SG(0x100, 1); M324_SG(0x008, 1); M324_SG(0x100, 1);
        M324_SG(0x008, 1); M324_SG(0x100, 1); M324_SG(0x008, 1); M324_SG(0x100, 1);
        M324_SG(0x008, 1); M324_SG(0x100, 1); M324_SG(0x008, 1); M324_SG(0x100, 1);
        M324_SG(0x008, 1); M324_SG(0x020, 1); M324_SG(0x008, 1); M324_SG(0x020, 2);
    };

    // fp32 residual stream (the trunk's / DINO's out-projection and fc2: plain rows, RES == 1): its values for the epilogue are
    // requested first of all (gemm_tile.h res_prefetch; 128 registers of the 284 this one-wave-per-SIMD kernel leaves unused).  They
    // are older than every ring piece, so the counted waits below retire them with stage 0.
    constexpr bool PRE_RES = RES == 1 && sizeof(TOUT) == 4;
    ResPre<4> pres;
    const bool use_pres = PRE_RES && ep.residual != nullptr && (xcd_remap & 16) == 0;       // bit 4: A/B (M324_XCD)
    if constexpr (PRE_RES) {
        if (use_pres) res_prefetch<4>(ep, M, N, m0 + wm * 128, n0 + wn * 64, lane, pres);
    }
    // prologue: the whole ring (stages 0, 1, 2); stage 0 then has nothing to issue and is peeled
    issue3(0, 0, 0); issue3(3, 0, 0); issue3(6, 0, 0); issue3(9, 0, 0);
    {
        const int s1 = NS > 1 ? 1 : 0, s2 = NS > 2 ? 2 : NS - 1;
        issue3(0, s1, 1); issue3(3, s1, 1); issue3(6, s1, 1); issue3(9, s1, 1);
        issue3(0, s2, 2); issue3(3, s2, 2); issue3(6, s2, 2); issue3(9, s2, 2);
    }
    asm volatile("s_waitcnt vmcnt(24)" ::: "memory");      // stage 0 landed (stages 1, 2 may fly)
    M324_BARRIER();
    int slot = 0;                                           // s % 3
    auto stage = [&](int s, auto issue_tag) {
        constexpr bool ISSUE = decltype(issue_tag)::value;
        const int fslot = slot == 0 ? 2 : slot - 1;         // (s + 2) % 3 == (s - 1) % 3
        const int sn = s + 2 < NS ? s + 2 : NS - 1;
        const bool live = s + 2 < NS || refetch;
        load_frags(0, slot, 0);
        if constexpr (ISSUE) issue3(0, sn, fslot, live);
        mma8(1);                                            // (s-1, k-step 3); zeros in the first iteration
        sched_phase();
        load_frags(1, slot, 1);
        if constexpr (ISSUE) issue3(3, sn, fslot, live);
        mma8(0);
        sched_phase();
        load_frags(0, slot, 2);
        if constexpr (ISSUE) issue3(6, sn, fslot, live);
        mma8(1);
        sched_phase();
        load_frags(1, slot, 3);
        if constexpr (ISSUE) issue3(9, sn, fslot, live);
        mma8(0);
        sched_phase();
        asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" ::: "memory");
        M324_BARRIER();
        slot = slot == 2 ? 0 : slot + 1;
    };
    stage(0, std::false_type{});
    for (int s = 1; s < NS; ++s) stage(s, std::true_type{});
    mma8(1);                                                // (NS-1, k-step 3)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // no LDS-DMA may outlive the main loop: the ring becomes scratch
#undef M324_SG
    M324_BARRIER();
    store_tile_lds<TOUT, ACT, RES, 4>(acc, reinterpret_cast<float*>(smem) + wave * ep_wave_floats(ACT), C, ldc, M, N, m0 + wm * 128,
                                      n0 + wn * 64, lane, ep, nullptr, PRE_RES ? &pres : nullptr, use_pres);
}


}  // namespace

namespace m324 {

int launch_ring4(const m324_gemm_args* a, hipStream_t s, const Epilogue& ep, int act_code, int res_code, int xcd_remap, int variant) {
    const int ntn = ceil_div(a->N, variant == 12 ? 128 : BN5), ntiles = ntn * ceil_div(a->M, BM5);
    static const int n_cu = [] {                            // v11: one persistent workgroup per CU (160 KiB of LDS each)
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
            n = 256;
        return n;
    }();
    const dim3 grid(variant == 12 ? ntiles : (ntiles < n_cu ? ntiles : n_cu));
#define M324_R4(TOUT, ACT, RES)                                                                                              \
    do {                                                                                                                     \
        if (variant == 12)                                                                                            \
            hipLaunchKernelGGL((gemm_ring3_kernel<TOUT, ACT, RES>), grid, dim3(256), 0, s, (const bf16_t*)a->A, a->lda,      \
                               (const bf16_t*)a->W, a->ldw, (TOUT*)a->C, a->ldc, a->M, a->N, a->K, ep, ntn, xcd_remap);      \
        else                                                                                                                 \
            hipLaunchKernelGGL((gemm_ring4_kernel<TOUT, ACT, RES>), grid, dim3(256), 0, s, (const bf16_t*)a->A, a->lda,      \
                               (const bf16_t*)a->W, a->ldw, (TOUT*)a->C, a->ldc, a->M, a->N, a->K, ep, ntn, ntiles,          \
                               xcd_remap);                                                                                   \
    } while (0)
#define M324_R4_OUT(ACT, RES)                                      \
    do {                                                           \
        if (a->out_dtype == M324_BF16) M324_R4(bf16_t, ACT, RES);  \
        else M324_R4(float, ACT, RES);                             \
    } while (0)
    const int key = act_code * 4 + res_code;
    switch (key) {
        case 0 * 4 + 0: M324_R4_OUT(0, 0); break;
        case 0 * 4 + 1: M324_R4_OUT(0, 1); break;
        case 0 * 4 + 2: M324_R4_OUT(0, 2); break;
        case 1 * 4 + 0: M324_R4_OUT(1, 0); break;
        case 1 * 4 + 2: M324_R4_OUT(1, 2); break;
        case 2 * 4 + 0: M324_R4_OUT(2, 0); break;
        case 3 * 4 + 0: M324_R4_OUT(3, 0); break;
        case 4 * 4 + 0: M324_R4(bf16_t, 4, 0); break;
        case 8 * 4 + 0: M324_R4_OUT(8, 0); break;          // the LayerNorm-fold instantiations: consumers (ACT | 8) ...
        case 9 * 4 + 0: M324_R4(bf16_t, 9, 0); break;
        case 12 * 4 + 0: M324_R4(bf16_t, 12, 0); break;
        case 16 * 4 + 1: M324_R4(bf16_t, 16, 1); break;    // ... producers of a bf16 stream (statistics) ...
        case 16 * 4 + 2: M324_R4(bf16_t, 16, 2); break;
        case 48 * 4 + 1: M324_R4(float, 48, 1); break;     // ... and of an fp32 stream (statistics + bf16 twin)
        case 48 * 4 + 2: M324_R4(float, 48, 2); break;
        default:                   // m324_gemm only builds the combinations above
            M324_FAIL(M324_ERR_UNSUPPORTED, "m324_gemm: no 4-wave ring kernel for epilogue act=%d res=%d", act_code, res_code);
    }
#undef M324_R4_OUT
#undef M324_R4
    return M324_OK;
}

}  // namespace m324
