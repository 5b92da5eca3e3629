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
    const bf16_t* ga[8];
    const bf16_t* gb[8];
    int m0 = 0, n0 = 0;
    auto tile_setup = [&](int t) {
        int lid = t;
        if (xcd_remap & 1) {
            const int q = ntiles >> 3, r = ntiles & 7, x = lid & 7, loc = lid >> 3;
            lid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + loc;
        }
        m0 = (lid / ntn) * BM5;
        n0 = (lid % ntn) * BN5;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int r = (wave * 8 + i) * 8 + (lane >> 3);
            const int c = ((lane & 7) ^ ((r >> 1) & 7)) * 8;
            ga[i] = A + (long)min(m0 + r, M - 1) * lda + c;
            gb[i] = W + (long)min(n0 + r, N - 1) * ldw + c;
        }
    };
    auto issue4 = [&](const bf16_t* const (&g)[8], int i0, int st, int pos) {
        unsigned char* d = smem + pos * CHUNK10 + wave * 8192 + i0 * 1024;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((glb_ptr_t*)(g[i0 + i] + (long)st * 64), (lds_ptr_t*)(d + i * 1024), 16, 0, 0);
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
        load_frags(0, pa, pw, 0);
        if constexpr (ISSUE) issue4(gb, 0, sw, pwn);
        mma16(1);                                           // (s-1, k-step 3); zeros in the first iteration
        sched_phase();
        load_frags(1, pa, pw, 1);
        if constexpr (ISSUE) issue4(gb, 4, sw, pwn);
        mma16(0);
        sched_phase();
        load_frags(0, pa, pw, 2);
        if constexpr (ISSUE) issue4(ga, 0, sa, pan);
        mma16(1);
        sched_phase();
        load_frags(1, pa, pw, 3);
        if constexpr (ISSUE) issue4(ga, 4, sa, pan);
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
        float* scr = reinterpret_cast<float*>(smem + 3 * CHUNK10) + wave * EP_WAVE_FLOATS;
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
    int lid = blockIdx.x;
    if (xcd_remap & 1) {
        const int nb = gridDim.x, q = nb >> 3, r = nb & 7, x = lid & 7, loc = lid >> 3;
        lid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + loc;
    }
    const int m0 = (lid / ntn) * 256, n0 = (lid % ntn) * 128;

    // LDS-DMA pieces of 8 rows x 128 B: wave w moves row groups 8w .. 8w+7 of A (32 groups) and 4w .. 4w+3 of W (16 groups)
    const bf16_t* g[12];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int r = (wave * 8 + i) * 8 + (lane >> 3);
        g[i] = A + (long)min(m0 + r, M - 1) * lda + ((lane & 7) ^ ((r >> 1) & 7)) * 8;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (wave * 4 + i) * 8 + (lane >> 3);
        g[8 + i] = W + (long)min(n0 + r, N - 1) * ldw + ((lane & 7) ^ ((r >> 1) & 7)) * 8;
    }
    auto issue3 = [&](int i0, int st, int slot) {           // pieces i0 .. i0+2 of stage st
        unsigned char* base = smem + slot * STAGE12;
#pragma unroll
        for (int i = i0; i < i0 + 3; ++i) {
            unsigned char* d = i < 8 ? base + wave * 8192 + i * 1024 : base + 32768 + wave * 4096 + (i - 8) * 1024;
            __builtin_amdgcn_global_load_lds((glb_ptr_t*)(g[i] + (long)st * 64), (lds_ptr_t*)d, 16, 0, 0);
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
        load_frags(0, slot, 0);
        if constexpr (ISSUE) issue3(0, sn, fslot);
        mma8(1);                                            // (s-1, k-step 3); zeros in the first iteration
        sched_phase();
        load_frags(1, slot, 1);
        if constexpr (ISSUE) issue3(3, sn, fslot);
        mma8(0);
        sched_phase();
        load_frags(0, slot, 2);
        if constexpr (ISSUE) issue3(6, sn, fslot);
        mma8(1);
        sched_phase();
        load_frags(1, slot, 3);
        if constexpr (ISSUE) issue3(9, sn, fslot);
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
    store_tile_lds<TOUT, ACT, RES, 4>(acc, reinterpret_cast<float*>(smem) + wave * EP_WAVE_FLOATS, C, ldc, M, N, m0 + wm * 128,
                                      n0 + wn * 64, lane, ep);
}


// ------------------------------------------------------------------------------------------------
// v14: persistent 256 x 128 tiles with a DEFERRED epilogue.
// tools/gemm_lab with the epilogue compiled out (round 2): 30-45 % of every K = 768 GEMM of the clip is its epilogue --
// bias / GELU / residual arithmetic, the LDS transpose and, above all, a burst of stores (and residual loads) that all
// CUs issue at the same moment, while the matrix pipes idle and HBM sat idle during the main loops before.  Here a
// workgroup keeps TWO accumulator sets (4 waves, one per SIMD, each a 128 x 64 block: 2 x 128 accumulators in the AGPR
// half of the register file) and walks its tiles as ONE continuous K-stage stream: while the MFMAs of tile t+1 run, the
// finished accumulators of tile t are drained in eight slots of 16 rows x 64 columns per wave, one per K-stage, in
// K-stages 1..8 of tile t+1, their instructions pinned between the MFMAs with sched_group_barrier.
//  * LDS: a ring of FIVE 24-KiB chunks (chunk 2g + h = rows [128 h, 128 h + 128) of A and [64 h, 64 h + 64) of W for the
//    64 k of stream stage g: 2.5 stages resident, v10's scheme) + a private 8.5-KiB transpose scratch per wave.
//  * The LDS-DMA stream runs 1.5-2 stages ahead of the MFMAs and simply continues into the next tile: no prologue per
//    tile, no re-fetched stages; past the last tile it re-reads its last stage into free chunks (never read).
//  * All LDS-DMA pieces and the residual / bias / gamma loads are INLINE-ASM buffer loads: hipcc's wait-count pass then
//    knows of no outstanding load, so it neither guards the scratch's ds_write / ds_read with a vmcnt(0) (it cannot
//    prove that an LDS-DMA does not alias them) nor drains the queue for the residual rows; the counted vmcnt at the end
//    of every K-stage is the only wait.  Per-lane offsets are loop-invariant VGPRs, everything tile- and stage-dependent
//    is SGPR arithmetic on the buffer descriptors / scalar offsets.
//  * Every slot is branch-free: stores are buffer stores whose descriptor is cut to the rows that exist (rows past M in
//    the last row tile, and everything while there is no previous tile yet, fall outside and are dropped by the
//    hardware's range check), so the slots stay inside the basic block whose MFMAs they hide under.
// Needs K >= 576 (nine K-stages carry the slots), N % 128 == 0, M % 8 == 0, every operand below 2 GiB.
constexpr int CH14 = (128 + 64) * ROWB;          // 24 KiB
constexpr int EP14 = EP_WAVE_FLOATS * 4;         // 8704 B of transpose scratch per wave
typedef int i32x4 __attribute__((ext_vector_type(4)));
#define M324_INL __attribute__((always_inline))

__device__ __forceinline__ i32x4 make_rsrc(const void* base, unsigned bytes) {
    const unsigned long long b = (unsigned long long)base;
    i32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
    r[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32));     // stride 0: raw buffer
    r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
    r[3] = 0x00020000;
    return r;
}
// one LDS-DMA piece: 64 lanes x 16 B from base + voff + soff into LDS at lds + lane * 16
template <int IMM>
__device__ __forceinline__ void dma16(i32x4 rsrc, unsigned voff, unsigned soff, unsigned lds) {
    asm volatile("s_add_u32 m0, %3, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voff), "s"(rsrc), "s"(soff), "s"(lds),
                 "n"(IMM));
}
__device__ __forceinline__ f32x4 bload16(i32x4 rsrc, unsigned voff, unsigned soff) {
    f32x4 v;
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(v) : "v"(voff), "s"(rsrc), "s"(soff));
    return v;
}

template <typename TOUT, int ACT, int RES>
__global__ __launch_bounds__(256) void gemm_dfe_kernel(const bf16_t* __restrict__ A, long lda, const bf16_t* __restrict__ W,
                                                       long ldw, TOUT* C, long ldc, int M, int N, int K, Epilogue ep, int ntn,
                                                       int ntiles, int xcd_remap) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[5 * CH14 + 4 * EP14];
    constexpr int ESZ = sizeof(TOUT);
    constexpr bool WIDE = ESZ == 2;                          // bf16 outputs: 8 columns per lane, else 4
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, hi = lane >> 5;
    const int NS = K / 64;
    const unsigned smem0 = (unsigned)(uintptr_t)(lds_ptr_t*)smem;
    float* const scr = reinterpret_cast<float*>(smem + 5 * CH14 + wave * EP14);

    auto tile_origin = [&](int t, int& m0, int& n0) M324_INL {
        int lid = t;
        if (xcd_remap & 1) {
            const int q = ntiles >> 3, r = ntiles & 7, x = lid & 7, loc = lid >> 3;
            lid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + loc;
        }
        m0 = (lid / ntn) * 256;
        n0 = (lid % ntn) * 128;
    };

    using Z = std::integral_constant<int, 0>;
    using T3 = std::integral_constant<int, 3>;
    // ---- LDS-DMA issue streams.  Half h of a stage = chunk 2g + h: wave w moves row groups 4w .. 4w+3 of the chunk's 128
    // A rows and 2w, 2w+1 of its 64 W rows (pieces of 8 rows x 128 B).  Stream H1 (half 1) runs one stage ahead of the
    // MFMAs, stream H0 (half 0) two stages; each walks tile after tile on its own.  Per-lane part of a piece's address:
    // row (lane >> 3) of the group, 16-byte column (lane & 7) ^ ((row >> 1) & 7) with row = 8 g + (lane >> 3), i.e.
    // (4 (g & 1) + (lane >> 4)) & 7 -- it depends on the PARITY of the group only.  The scalar offset of a piece = byte
    // offset of its row group + 128 B per K-stage.
    const i32x4 rsA = make_rsrc(A, 0x7FFFFFFFu), rsW = make_rsrc(W, 0x7FFFFFFFu);
    const unsigned vo_a[2] = {(unsigned)((lane >> 3) * lda * 2 + (((lane & 7) ^ ((lane >> 4) & 7)) << 4)),
                              (unsigned)((lane >> 3) * lda * 2 + (((lane & 7) ^ ((4 + (lane >> 4)) & 7)) << 4))};
    const unsigned vo_w[2] = {(unsigned)((lane >> 3) * ldw * 2 + (((lane & 7) ^ ((lane >> 4) & 7)) << 4)),
                              (unsigned)((lane >> 3) * ldw * 2 + (((lane & 7) ^ ((4 + (lane >> 4)) & 7)) << 4))};
    struct Stream {
        unsigned so[6];                                      // scalar byte offsets of the six pieces (4 of A, 2 of W)
        int t, st;                                           // tile, stage within the tile of the NEXT issue
    };
    Stream S0, S1;
    auto stream_setup = [&](Stream& s, int h) M324_INL {
        int m0, n0;
        tile_origin(s.t, m0, n0);
#pragma unroll
        for (int i = 0; i < 4; ++i)                          // M % 8 == 0: a row group is all inside or all outside
            s.so[i] = (unsigned)__builtin_amdgcn_readfirstlane(min(m0 + 128 * h + (wave * 4 + i) * 8, M - 8) * (int)lda * 2);
#pragma unroll
        for (int i = 0; i < 2; ++i)
            s.so[4 + i] = (unsigned)__builtin_amdgcn_readfirstlane((n0 + 64 * h + (wave * 2 + i) * 8) * (int)ldw * 2);
    };
    const unsigned dstA = smem0 + wave * 4096, dstW = smem0 + 16384 + wave * 2048;      // this wave's pieces inside a chunk
    auto issue3 = [&](const Stream& s, auto i0_tag, int pos) M324_INL {   // pieces i0 .. i0+2 of the stream's current (tile, stage)
        constexpr int I0 = decltype(i0_tag)::value;
        const unsigned cb = pos * CH14;
        if constexpr (I0 == 0) {
            dma16<0>(rsA, vo_a[0], s.so[0], dstA + cb);
            dma16<1024>(rsA, vo_a[1], s.so[1], dstA + cb);
            dma16<2048>(rsA, vo_a[0], s.so[2], dstA + cb);
        } else {
            dma16<3072>(rsA, vo_a[1], s.so[3], dstA + cb);
            dma16<0>(rsW, vo_w[0], s.so[4], dstW + cb);
            dma16<1024>(rsW, vo_w[1], s.so[5], dstW + cb);
        }
    };
    // the common step (next K-stage of the same tile) is branch-free SALU that floats between the MFMAs; the tile change
    // (a handful of divisions) is a branch, taken once per tile, at the end of the stage
    auto stream_step = [&](Stream& s) M324_INL {
        ++s.st;
#pragma unroll
        for (int i = 0; i < 6; ++i) s.so[i] += 128;
    };
    auto stream_wrap = [&](Stream& s, int h) M324_INL {
        if (s.st == NS) {
            if (s.t + (int)gridDim.x < ntiles) {
                s.t += gridDim.x;
                s.st = 0;
                stream_setup(s, h);
            } else {                                          // past the end: the last stage again (never read)
                s.st = NS - 1;
#pragma unroll
                for (int i = 0; i < 6; ++i) s.so[i] -= 128;
            }
        }
    };

    // ---- fragments / MFMAs
    const int aoff = lds_off(l31, hi), boff = 16384 + lds_off(l31, hi);
    bf16x8 fa[2][4], fb[2][2];
    f32x16 acc[2][4][2];
    auto load_frags = [&](int set, const unsigned char* ba, const unsigned char* bw, int ks) M324_INL {
        const int x = ks << 5;
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[set][j] = *reinterpret_cast<const bf16x8*>(bw + ((boff + j * 4096) ^ x));
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[set][i] = *reinterpret_cast<const bf16x8*>(ba + ((aoff + i * 4096) ^ x));
    };
#define M324_MMA8(FSET, ASET)                                                                                                    \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)                                  \
        acc[ASET][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[FSET][j], fa[FSET][i], acc[ASET][i][j], 0, 0, 0)
#define M324_SG(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
    // a phase = 8 MFMAs with the 6 fragment reads of the next k-step behind the first six.  Memory instructions keep
    // their PROGRAM order around the inline-asm loads (asm volatile is a fence for the scheduler's memory operations), so
    // the source below lists them in the order they are to issue; MFMAs and VALU float, and these groups pin them:
    // EW / ER: LDS writes / extra LDS reads of the riding epilogue slot per MFMA, EV: its VALU per MFMA, ES: stores behind
    // each of the last four MFMAs.
    auto sched_phase = [&](auto ew, auto er, auto ev, auto es) M324_INL {
        constexpr int EW = decltype(ew)::value, ER = decltype(er)::value, EV = decltype(ev)::value, ES = decltype(es)::value;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            M324_SG(0x008, 1);
            if (k < 6) M324_SG(0x100, 1);
            if (EW) M324_SG(0x200, EW);
            if (ER && k >= 6) M324_SG(0x100, ER);
            if (EV) M324_SG(0x002, EV);
            if (ES && k >= 4) M324_SG(0x040, ES);
        }
    };

    // ---- deferred epilogue of the PREVIOUS tile (origin pm, pn).  Slot e = rows [16 e, 16 e + 16) of the wave's 128 x 64:
    // block i = e >> 1 (the 32 x 64 unit of the LDS transpose, written to the scratch by its first slot), half = e & 1.
    int pm = 0, pn = 0;
    unsigned out_bytes = 0;                                  // 0 while there is no previous tile: every store falls outside
    // lane parts of the output / residual addresses (row within the slot, column within the wave's 64): loop invariant
    const int lrow = WIDE ? (lane >> 3) : (lane >> 4), lcol = WIDE ? (lane & 7) * 8 : (lane & 15) * 4;
    unsigned vo_out[4], vo_res[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int r = lrow + (WIDE ? 8 : 4) * p;
        vo_out[p] = (unsigned)((r * (int)ldc + lcol) * ESZ);
        vo_res[p] = (unsigned)((r * (int)ep.ldr + lcol) * 4);
    }
    const unsigned res_rows_eff = (RES == 2 && ep.res_rows > 0 && ep.res_rows < M) ? (unsigned)ep.res_rows : (unsigned)M;
    f32x4 res[4];                                            // fp32 residual rows of the next slot, fetched a stage ahead
    f32x4 bia[2], gam[2];                                    // bias / gamma of this lane's columns (previous tile's pn)
    auto epi_consts = [&]() M324_INL {                       // once per tile: bias / gamma of the previous tile's columns
        const int nc = pn + wn * 64 + lcol;
        const i32x4 rb = make_rsrc(ep.bias, ep.bias ? (unsigned)N * 4 : 0u);          // null -> zeros
        const i32x4 rg = make_rsrc(ep.gamma, ep.gamma ? (unsigned)N * 4 : 0u);
        bia[0] = bload16(rb, (unsigned)nc * 4, 0);
        gam[0] = bload16(rg, (unsigned)nc * 4, 0);
        if (WIDE) {
            bia[1] = bload16(rb, (unsigned)nc * 4 + 16, 0);
            gam[1] = bload16(rg, (unsigned)nc * 4 + 16, 0);
        }
    };
    // descriptors of the residual and of the output are fixed; a slot adds its uniform offset to the lane parts, so the
    // hardware's range check sees the whole offset: rows past the end read zeros / are not stored
    const i32x4 rsR = make_rsrc(ep.residual, RES != 0 ? res_rows_eff * (unsigned)ep.ldr * 4 : 0u);
    unsigned prow = 0;                                       // previous tile's first row modulo the residual's broadcast period
    auto epi_prefetch = [&](int e) M324_INL {
        if constexpr (RES != 0) {
            unsigned rb = prow + (unsigned)(wm * 128 + e * 16);          // rows [rb, rb + 16), period a multiple of 16 and >= 256
            if (RES == 2) rb = rb >= res_rows_eff ? rb - res_rows_eff : rb;
            const unsigned so = (rb * (unsigned)ep.ldr + (unsigned)(pn + wn * 64)) * 4;
#pragma unroll
            for (int p = 0; p < 4; ++p) res[p] = bload16(rsR, vo_res[p] + so, 0);
        }
    };
    auto slot_write = [&](const f32x16 (&a)[2]) M324_INL {  // accumulators of one 32 x 64 block -> scratch (lane = row)
        float* wr = scr + l31 * EP_LD + 4 * hi;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(wr + j * 32 + 8 * g) = make_float4(a[j][4 * g], a[j][4 * g + 1], a[j][4 * g + 2], a[j][4 * g + 3]);
    };
    float4 sx[4];                                            // the slot's rows between its LDS reads and its stores
    auto slot_read = [&](int e) M324_INL {
        const int half = e & 1;
        const float* rd = scr + (half * 16 + lrow) * EP_LD + lcol;
        if constexpr (WIDE) {
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                sx[2 * p] = *reinterpret_cast<const float4*>(rd + p * 8 * EP_LD);
                sx[2 * p + 1] = *reinterpret_cast<const float4*>(rd + p * 8 * EP_LD + 4);
            }
        } else {
#pragma unroll
            for (int p = 0; p < 4; ++p) sx[p] = *reinterpret_cast<const float4*>(rd + p * 4 * EP_LD);
        }
    };
    auto slot_store = [&](int e) M324_INL {
        const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)C, 0, (int)out_bytes, 0x00020000);
        const unsigned so = ((unsigned)(pm + wm * 128 + e * 16) * (unsigned)ldc + (unsigned)(pn + wn * 64)) * ESZ;
        if constexpr (WIDE) {
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                float4 x = sx[2 * p], y = sx[2 * p + 1];
                x.x += bia[0][0]; x.y += bia[0][1]; x.z += bia[0][2]; x.w += bia[0][3];
                y.x += bia[1][0]; y.y += bia[1][1]; y.z += bia[1][2]; y.w += bia[1][3];
                if (ACT == 1) { apply_gelu4<TOUT>(x); apply_gelu4<TOUT>(y); }
                x.x *= gam[0][0]; x.y *= gam[0][1]; x.z *= gam[0][2]; x.w *= gam[0][3];
                y.x *= gam[1][0]; y.y *= gam[1][1]; y.z *= gam[1][2]; y.w *= gam[1][3];
                const u32x4v o = {pack_bf16x2(x.x, x.y), pack_bf16x2(x.z, x.w), pack_bf16x2(y.x, y.y), pack_bf16x2(y.z, y.w)};
                __builtin_amdgcn_raw_buffer_store_b128(o, ro, (int)(vo_out[p] + so), 0, 0);
            }
        } else {
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                float4 x = sx[p];
                x.x += bia[0][0]; x.y += bia[0][1]; x.z += bia[0][2]; x.w += bia[0][3];
                if (ACT == 1) apply_gelu4<TOUT>(x);
                x.x *= gam[0][0]; x.y *= gam[0][1]; x.z *= gam[0][2]; x.w *= gam[0][3];
                if constexpr (RES != 0) { x.x += res[p][0]; x.y += res[p][1]; x.z += res[p][2]; x.w += res[p][3]; }
                const u32x4v o = {__float_as_uint(x.x), __float_as_uint(x.y), __float_as_uint(x.z), __float_as_uint(x.w)};
                __builtin_amdgcn_raw_buffer_store_b128(o, ro, (int)(vo_out[p] + so), 0, 0);
            }
        }
    };

    // ---- one K-stage.  FIRST: the tile's first stage, whose phase 0 still multiplies the previous tile's last k-step;
    // EPI = e: fetch the residual rows of slot e (e <= 7; e = 0 also bias / gamma) and run slot e - 1 (e >= 1) of the
    // previous tile; -1: neither.
    int pa = 0;                                              // ring position of chunk 2g (half 0 of the stage being read)
    auto stage = [&](auto set_tag, auto first_tag, auto epi_tag) M324_INL {
        constexpr int SET = decltype(set_tag)::value;
        constexpr bool FIRST = decltype(first_tag)::value;
        constexpr int EPI = decltype(epi_tag)::value;
        constexpr bool SLOT = EPI >= 1;
        constexpr int EVn = !SLOT ? 0 : (ACT == 1 ? (WIDE ? 16 : 8) : 4);   // VALU of the slot per MFMA of phase 3
        const int pb = pa + 1 >= 5 ? pa - 4 : pa + 1;
        int p3 = pa + 3, p4 = pa + 4;                        // positions of chunks 2g+3 (stream H1), 2g+4 (stream H0)
        p3 = p3 >= 5 ? p3 - 5 : p3;
        p4 = p4 >= 5 ? p4 - 5 : p4;
        const unsigned char* ba = smem + (wm ? pb : pa) * CH14;
        const unsigned char* bw = smem + (wn ? pb : pa) * CH14;
        constexpr bool SW = SLOT && ((EPI - 1) & 1) == 0;    // this slot opens a 32 x 64 block: accumulators -> scratch
        // phase 0: MFMAs of k-step 3 of the previous stage | fragments of k-step 0, the slot's LDS writes, pieces 0-2 of H1
        load_frags(0, ba, bw, 0);
        if constexpr (SW) slot_write(acc[SET ^ 1][(EPI - 1) >> 1]);
        issue3(S1, Z{}, p3);
        if constexpr (FIRST) { M324_MMA8(1, SET ^ 1); } else { M324_MMA8(1, SET); }
        sched_phase(std::integral_constant<int, SW ? 1 : 0>{}, Z{}, Z{}, Z{});
        // phase 1: k-step 0 | fragments of k-step 1, the next slot's residual rows (+ bias / gamma once per tile), pieces 3-5 of H1
        load_frags(1, ba, bw, 1);
        if constexpr (EPI == 0) epi_consts();
        if constexpr (EPI >= 0 && EPI <= 7) epi_prefetch(EPI);
        issue3(S1, T3{}, p3);
        stream_step(S1);
        M324_MMA8(0, SET);
        sched_phase(Z{}, Z{}, Z{}, Z{});
        // phase 2: k-step 1 | fragments of k-step 2, the slot's LDS reads, pieces 0-2 of H0
        load_frags(0, ba, bw, 2);
        if constexpr (SLOT) slot_read(EPI - 1);
        issue3(S0, Z{}, p4);
        M324_MMA8(1, SET);
        sched_phase(Z{}, std::integral_constant<int, SLOT ? 2 : 0>{}, Z{}, Z{});
        // phase 3: k-step 2 | fragments of k-step 3, the slot's arithmetic and stores, pieces 3-5 of H0
        load_frags(1, ba, bw, 3);
        if constexpr (SLOT) slot_store(EPI - 1);
        issue3(S0, T3{}, p4);
        stream_step(S0);
        M324_MMA8(0, SET);
        sched_phase(Z{}, Z{}, std::integral_constant<int, EVn>{}, std::integral_constant<int, SLOT ? 1 : 0>{});
        stream_wrap(S1, 1);
        stream_wrap(S0, 0);
        // chunk 2g+3 landed (the six pieces of 2g+4 may fly); the residual / bias rows fetched at the top of this stage are
        // older than both -- the "+v" ties keep their consumers behind this wait
        asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)"
                     : "+v"(res[0]), "+v"(res[1]), "+v"(res[2]), "+v"(res[3]), "+v"(bia[0]), "+v"(bia[1]), "+v"(gam[0]), "+v"(gam[1])::"memory");
        M324_BARRIER();
        pa = pa + 2 >= 5 ? pa - 3 : pa + 2;
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    auto run_tile = [&](auto set_tag) M324_INL {
        constexpr int SET = decltype(set_tag)::value;
        using ST = std::integral_constant<int, SET>;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[SET][i][j][r] = 0.f;
        stage(ST{}, std::true_type{}, std::integral_constant<int, 0>{});
        if (!ep.gamma) { gam[0] = (f32x4)(1.f); gam[1] = (f32x4)(1.f); }
        stage(ST{}, std::false_type{}, std::integral_constant<int, 1>{});
        stage(ST{}, std::false_type{}, std::integral_constant<int, 2>{});
        stage(ST{}, std::false_type{}, std::integral_constant<int, 3>{});
        stage(ST{}, std::false_type{}, std::integral_constant<int, 4>{});
        stage(ST{}, std::false_type{}, std::integral_constant<int, 5>{});
        stage(ST{}, std::false_type{}, std::integral_constant<int, 6>{});
        stage(ST{}, std::false_type{}, std::integral_constant<int, 7>{});
        stage(ST{}, std::false_type{}, std::integral_constant<int, 8>{});
        for (int s = 9; s < NS; ++s) stage(ST{}, std::false_type{}, std::integral_constant<int, -1>{});
    };
    auto drain = [&](auto set_tag) M324_INL {               // the very last tile: nothing left to hide its epilogue under
        constexpr int SET = decltype(set_tag)::value;
        M324_MMA8(1, SET);                                   // its last k-step
        epi_consts();
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(bia[0]), "+v"(bia[1]), "+v"(gam[0]), "+v"(gam[1])::"memory");
        if (!ep.gamma) { gam[0] = (f32x4)(1.f); gam[1] = (f32x4)(1.f); }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            epi_prefetch(e);
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(res[0]), "+v"(res[1]), "+v"(res[2]), "+v"(res[3])::"memory");
            if ((e & 1) == 0) slot_write(acc[SET][e >> 1]);
            slot_read(e);
            slot_store(e);
        }
    };

    // ---- prologue: chunks 0, 1 (stage 0) and 2 (half 0 of stage 1); afterwards H1 is one stage ahead, H0 two
    S0.t = S1.t = blockIdx.x;
    S0.st = S1.st = 0;
    stream_setup(S0, 0);
    stream_setup(S1, 1);
    issue3(S0, Z{}, 0); issue3(S0, T3{}, 0);
    stream_step(S0); stream_wrap(S0, 0);
    issue3(S1, Z{}, 1); issue3(S1, T3{}, 1);
    stream_step(S1); stream_wrap(S1, 1);
    issue3(S0, Z{}, 2); issue3(S0, T3{}, 2);
    stream_step(S0); stream_wrap(S0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) { fa[1][i] = (bf16x8)(0); }
#pragma unroll
    for (int j = 0; j < 2; ++j) { fb[1][j] = (bf16x8)(0); }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[1][i][j][r] = 0.f;
#pragma unroll
    for (int p = 0; p < 4; ++p) res[p] = (f32x4)(0.f);
    bia[0] = bia[1] = (f32x4)(0.f);
    gam[0] = gam[1] = (f32x4)(1.f);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");        // stage 0 landed (chunk 2 may fly)
    M324_BARRIER();

    int t = blockIdx.x;
    while (true) {
        int m0, n0;
        tile_origin(t, m0, n0);
        run_tile(I0{});                                      // accumulates set 0, drains set 1 (the previous tile)
        pm = m0; pn = n0; prow = (unsigned)m0 % res_rows_eff; out_bytes = (unsigned)M * (unsigned)ldc * ESZ;
        t += gridDim.x;
        if (t >= ntiles) { drain(I0{}); break; }
        tile_origin(t, m0, n0);
        run_tile(I1{});
        pm = m0; pn = n0; prow = (unsigned)m0 % res_rows_eff;
        t += gridDim.x;
        if (t >= ntiles) { drain(I1{}); break; }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the re-fetched chunks: no LDS-DMA may outlive the workgroup
#undef M324_SG
#undef M324_MMA8
}

}  // namespace

namespace m324 {

int launch_ring4(const m324_gemm_args* a, hipStream_t s, const Epilogue& ep, int act_code, int res_code, int xcd_remap, int variant) {
#ifdef M324_DFE_LAB      // compile-time lab: only two v14 instantiations (fast rebuilds while reading the ISA)
    {
        const int ntn_ = ceil_div(a->N, 128), ntiles_ = ntn_ * ceil_div(a->M, BM5);
        const dim3 grid_(ntiles_ < 256 ? ntiles_ : 256);
        if (a->out_dtype == M324_F32)
            hipLaunchKernelGGL((gemm_dfe_kernel<float, 0, 1>), grid_, dim3(256), 0, s, (const bf16_t*)a->A, a->lda, (const bf16_t*)a->W,
                               a->ldw, (float*)a->C, a->ldc, a->M, a->N, a->K, ep, ntn_, ntiles_, xcd_remap);
        else
            hipLaunchKernelGGL((gemm_dfe_kernel<bf16_t, 1, 0>), grid_, dim3(256), 0, s, (const bf16_t*)a->A, a->lda, (const bf16_t*)a->W,
                               a->ldw, (bf16_t*)a->C, a->ldc, a->M, a->N, a->K, ep, ntn_, ntiles_, xcd_remap);
        return M324_OK;
    }
#else
    const int ntn = ceil_div(a->N, (variant == 12 || variant == 14) ? 128 : BN5), ntiles = ntn * ceil_div(a->M, BM5);
    static const int n_cu = [] {                            // v11: one persistent workgroup per CU (160 KiB of LDS each)
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
            n = 256;
        return n;
    }();
    const dim3 grid(variant == 12 ? ntiles : (ntiles < n_cu ? ntiles : n_cu));
#define M324_R4(TOUT, ACT, RES)                                                                                              \
    do {                                                                                                                     \
        if (variant == 14) {                                                                                                 \
            if constexpr ((ACT) == 0 || (ACT) == 1)                                                            \
                hipLaunchKernelGGL((gemm_dfe_kernel<TOUT, ACT, RES>), grid, dim3(256), 0, s, (const bf16_t*)a->A, a->lda,    \
                                   (const bf16_t*)a->W, a->ldw, (TOUT*)a->C, a->ldc, a->M, a->N, a->K, ep, ntn, ntiles,      \
                                   xcd_remap);                                                                               \
        } else if (variant == 12)                                                                                            \
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
        default:                   // m324_gemm only builds the combinations above
            M324_FAIL(M324_ERR_UNSUPPORTED, "m324_gemm: no 4-wave ring kernel for epilogue act=%d res=%d", act_code, res_code);
    }
#undef M324_R4_OUT
#undef M324_R4
    return M324_OK;
#endif
}

}  // namespace m324
