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

#include <stdio.h>
#include "gemm_tile.h"

namespace {

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
        mma_tile<TIN>(sa, sb, arow0, brow0, hi, acc);
        if (more) store_tile((kt + 1) & 1);
        __syncthreads();
    }

    store_tile_out<TOUT, -1, -1>(acc, C, ldc, M, N, m0 + wm * 64, n0 + wn * 64, l31, hi, ep);
}


// ------------------------------------------------------------------------------------------------
// v2: same tile / LDS image / MFMA schedule, but the K-tiles are staged by LDS-DMA
// (buffer_load_dwordx4 ... lds: HBM -> LDS without a VGPR round trip and without the ds_write pass that
// made v1 LDS-bound: 32 KiB of ds_write_b128 per K-tile at ~79 B/clk/CU is ~415 cycles against 512
// cycles of MFMA).  A wave-instruction writes 1 KiB = 8 tile rows linearly (LDS address = wave-uniform
// base + lane * 16), so the XOR swizzle is applied to the SOURCE address: lane l fills row
// r = 8*g + (l >> 3), slot p = l & 7, which must hold logical chunk c = p ^ ((r >> 1) & 7).
// Rows past M (or N) are clamped to the last valid row instead of zero-filled: an output row depends
// only on its own A row / W row, and rows >= M, columns >= N are never stored.
// Double-buffered; the barrier at the top of iteration kt both publishes tile kt (hipcc drains the
// LDS-DMA queue, vmcnt(0), before s_barrier) and retires every wave's reads of the buffer that tile
// kt+1 is about to overwrite.

template <typename TIN, typename TOUT, int ACT, int RES>
__global__ __launch_bounds__(256, 2) void gemm_glds_kernel(const TIN* __restrict__ A, long lda, const TIN* __restrict__ W,
                                                           long ldw, TOUT* C, long ldc, int M, int N, int K, Epilogue ep,
                                                           int ntn, int xcd_remap) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[4 * TILE_BYTES];   // A0 B0 A1 B1
    constexpr int EPC = Elem<TIN>::PER16;
    constexpr int BK = ROWB / sizeof(TIN);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, hi = lane >> 5;
    // 1-D grid.  Workgroup b runs on XCD b % 8 (observed dispatch order; used for speed only): give every
    // XCD a contiguous range of logical tiles = whole A row-panels with all their column tiles, so a panel
    // is pulled into ONE XCD's L2 and re-used by its ntn column tiles instead of being fetched by all eight.
    int tm, tn;
    tile_of(blockIdx.x, gridDim.x, (M + BM - 1) / BM, ntn, xcd_remap, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;
    A += (long)blockIdx.y * ep.strideA;      // batched launch (split-K weight gradients): blockIdx.y = problem index
    W += (long)blockIdx.y * ep.strideW;
    C += (long)blockIdx.y * ep.strideC;

    // this wave stages row groups g = wave*4 + i (8 rows each) of both operands
    unsigned va[4], vb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (wave * 4 + i) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        va[i] = (unsigned)(((long)min(r, M - 1 - m0) * lda + c * EPC) * sizeof(TIN));
        vb[i] = (unsigned)(((long)min(r, N - 1 - n0) * ldw + c * EPC) * sizeof(TIN));
    }
    const __amdgpu_buffer_rsrc_t ra = dma_rsrc(A + (long)m0 * lda), rb = dma_rsrc(W + (long)n0 * ldw);
    auto issue_tile = [&](int kt, int buf) {
        unsigned char* sa = smem + buf * 2 * TILE_BYTES + wave * 4096;
        unsigned char* sb = sa + TILE_BYTES;
        const unsigned so = (unsigned)(kt * BK * sizeof(TIN));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            dma_piece(ra, sa + i * 1024, va[i], so);
            dma_piece(rb, sb + i * 1024, vb[i], so);
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
    const int arow0 = wm * 64 + l31, brow0 = wn * 64 + l31;
    LnPreT<2> ln_pre;
    ln_prefetch<ACT, 2>(ep, M, N, m0 + wm * 64, n0 + wn * 64, lane, ln_pre);
    issue_tile(0, 0);
    ln_finish<ACT, 2>(ep, lane, ln_pre);
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // tile kt landed (never left to __syncthreads()'s fence)
        __syncthreads();
        if (kt + 1 < nk) issue_tile(kt + 1, (kt + 1) & 1);
        const unsigned char* sa = smem + (kt & 1) * 2 * TILE_BYTES;
        mma_tile<TIN, true>(sa, sa + TILE_BYTES, arow0, brow0, hi, acc);
    }
    __syncthreads();                            // all waves are done reading the stages: reuse them as epilogue scratch
    store_tile_lds<TOUT, ACT, RES, 2>(acc, reinterpret_cast<float*>(smem) + wave * ep_wave_floats(ACT), C, ldc, M, N, m0 + wm * 64,
                                      n0 + wn * 64, lane, ep, &ln_pre);
}


// ------------------------------------------------------------------------------------------------
// v5: 256 x 256 tile, 8 waves as 2 (M) x 4 (N), each wave a 128 x 64 block = 4 x 2 accumulators
// (128 VGPR).  Per K-tile a wave issues 32 MFMAs (1024 matrix-pipe cycles) against 24 fragment reads,
// and the workgroup moves 64 KiB by LDS-DMA per 8.4 MFLOP -- half the LDS traffic per FLOP of the
// 128-wide tiles and twice the work per barrier.  Two LDS stages of 64 KiB (one workgroup per CU).
// Used where 256-wide column tiles quantise well (N % 256 == 0) and the grid still fills the chip.
constexpr int STAGE5 = (BM5 + BN5) * ROWB;   // 64 KiB

template <typename TIN, typename TOUT, int ACT, int RES>
__global__ __launch_bounds__(512, 2) void gemm_glds5_kernel(const TIN* __restrict__ A, long lda, const TIN* __restrict__ W,
                                                            long ldw, TOUT* C, long ldc, int M, int N, int K, Epilogue ep,
                                                            int ntn, int xcd_remap) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * STAGE5];
    constexpr int EPC = Elem<TIN>::PER16;
    constexpr int BK = ROWB / sizeof(TIN);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int l31 = lane & 31, hi = lane >> 5;
    int tm, tn;
    tile_of(blockIdx.x, gridDim.x, (M + BM5 - 1) / BM5, ntn, xcd_remap, tm, tn);
    const int m0 = tm * BM5, n0 = tn * BN5;

    // staging: 32 row groups of 8 rows per operand; wave w takes groups w*4 .. w*4+3 of A and of W
    unsigned va[4], vb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (wave * 4 + i) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        va[i] = (unsigned)(((long)min(r, M - 1 - m0) * lda + c * EPC) * sizeof(TIN));
        vb[i] = (unsigned)(((long)min(r, N - 1 - n0) * ldw + c * EPC) * sizeof(TIN));
    }
    const __amdgpu_buffer_rsrc_t ra = dma_rsrc(A + (long)m0 * lda), rb = dma_rsrc(W + (long)n0 * ldw);
    auto issue_tile = [&](int kt, int buf) {
        unsigned char* sa = smem + buf * STAGE5 + wave * 4096;
        unsigned char* sb = sa + BM5 * ROWB;
        const unsigned so = (unsigned)(kt * BK * sizeof(TIN));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            dma_piece(ra, sa + i * 1024, va[i], so);
            dma_piece(rb, sb + i * 1024, vb[i], so);
        }
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = K / BK;
    const int arow0 = wm * 128 + l31, brow0 = wn * 64 + l31;
    const LnPreT<4>* const ln_pre_ptr = nullptr;
    issue_tile(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // tile kt landed (never left to __syncthreads()'s fence)
        __syncthreads();
        if (kt + 1 < nk) issue_tile(kt + 1, (kt + 1) & 1);
        const unsigned char* sa = smem + (kt & 1) * STAGE5;
        const unsigned char* sb = sa + BM5 * ROWB;
        if constexpr (sizeof(TIN) == 2) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                bf16x8 af[4], bfr[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) bfr[j] = *reinterpret_cast<const bf16x8*>(sb + lds_off(brow0 + j * 32, ks * 2 + hi));
#pragma unroll
                for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const bf16x8*>(sa + lds_off(arow0 + i * 32, ks * 2 + hi));
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                f32x2 af[4], bfr[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) bfr[j] = *reinterpret_cast<const f32x2*>(sb + lds_off(brow0 + j * 32, c) + hi * 8);
#pragma unroll
                for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const f32x2*>(sa + lds_off(arow0 + i * 32, c) + hi * 8);
#pragma unroll
                for (int e = 0; e < 2; ++e)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bfr[j][e], af[i][e], acc[i][j], 0, 0, 0);
            }
        }
    }
    __syncthreads();
    store_tile_lds<TOUT, ACT, RES, 4>(acc, reinterpret_cast<float*>(smem) + wave * ep_wave_floats(ACT), C, ldc, M, N, m0 + wm * 128,
                                      n0 + wn * 64, lane, ep, ln_pre_ptr);
}


// ------------------------------------------------------------------------------------------------
// v10: the chunk ring.  A predecessor (v7, retired) streamed K in half-tiles of 32; tools/dma_lab: an LDS-DMA piece that
// covers 16 rows x 64 B (such half-tile rows) costs the texture path 41 cycles per KiB, one that covers 8 rows x 128 B 29-33 -- with 64-byte rows the loads
// alone cap a 256 x 256-tile GEMM at ~1700 TF/s, next to an MFMA-only ceiling of 1825.  Here an LDS row holds K = 64
// (128 B, v2/v5's swizzle) and the 160 KiB of LDS are a ring of FIVE 32-KiB chunks, a chunk being one operand's
// 256 rows x 64 k of one K-stage: chunk 2s = A of stage s, 2s+1 = W of stage s, chunk q at ring position q % 5, so
// stages s, s+1 and the A half of s+2 are resident while stage s is multiplied (2.5 stages of look-ahead).
// The MFMA work is rotated by one k-step against the LDS stages: an iteration runs the MFMAs of (s-1, k-step 3) --
// fragments already in registers -- then (s, 0..2), so barrier X_s sits where stage s is no longer read from LDS and
// the fragments of (s+1, 0) are fetched under MFMAs that do not need them:
//   X_{s-1}: stage s landed, stage s-1 free
//   phase 0: MFMA (s-1,3) | fragments (s,0) | 2 pieces of W_{s+1}      phase 2: MFMA (s,1) | fragments (s,2) | 2 pieces of A_{s+2}
//   phase 1: MFMA (s,0)   | fragments (s,1) | 2 pieces of W_{s+1}      phase 3: MFMA (s,2) | fragments (s,3) | 2 pieces of A_{s+2}
//   lgkmcnt(0) (stage s is in registers), vmcnt(4) (W_{s+1} landed; the 4 pieces of A_{s+2} may fly), X_s.
// Past the end of K the last stage is fetched again into free chunks (never read), so the loop is branch-free.


// Every ring kernel's body sees its LDS through THREE __restrict__ pointers -- LDS-DMA destinations, fragment reads, epilogue scratch
// (round 5).  hipcc's wait-count pass assumes that an LDS access may touch the bytes of any LDS-DMA it knows to be in flight and puts
// `s_waitcnt vmcnt(0)` in front of it: in the first K-stage of every tile (the ring prologue / the next tile's prefetched chunks had just
// been issued) and in front of the epilogue's first scratch access (the chunks prefetched under the epilogue had to LAND before the
// epilogue could start) -- a full drain of the pipeline at each.  The kernels' own counted waits and barriers order every chunk's fill
// against its reads; the alias scopes tell the pass so.  (The TN weight-gradient kernel lost 17-21 % of its time to the same drains.)
// INVARIANT the three views rest on (they DO name the same bytes, so by the letter of `restrict` this is outside the language):
// every pair of conflicting accesses -- a chunk's LDS-DMA fill and its fragment reads, a chunk's last read and its refill, a scratch
// write and the ring bytes under it -- is separated by a counted `s_waitcnt vmcnt` AND a workgroup barrier that are written as
// `asm volatile(... ::: "memory")`, which no LLVM pass moves memory operations across; hipcc only turns `restrict` into scoped-noalias
// metadata after inlining, i.e. it may reorder between two such fences, where no conflicting pair lives.  Guards: tools/audit_barriers.py
// (every barrier of an LDS-DMA kernel has its vmcnt wait in the ISA; tests/test_static.py), the compiler version the ISA was read on is
// pinned there too (a new hipcc must be re-validated: grep the loops for `vmcnt(0)`), and the race tests stay in the default GPU
// suite (tools/ln_stress.py form: test_kernels_gpu.py `*_beside_chunk_ring_*`, the v15 race test).
template <typename TOUT, int ACT, int RES>
__device__ __forceinline__ void gemm_ring_body(unsigned char* __restrict__ ring_w, const unsigned char* __restrict__ smem,
                                               float* __restrict__ scratch, const bf16_t* __restrict__ A, long lda,
                                               const bf16_t* __restrict__ W, long ldw, TOUT* C, long ldc, int M, int N, int K,
                                               const Epilogue& ep, int ntn, int xcd_remap) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int l31 = lane & 31, hi = lane >> 5;
    // Persistent: one workgroup per CU walks the tiles blockIdx.x, blockIdx.x + gridDim.x, ... (gridDim.x is a multiple of 8
    // or the tile count itself, so a workgroup's tiles stay on its XCD's share of the XCD-aware order).  Stamps of a K = 768
    // tile (tools/gemm_trace.py): 4300 cycles of prologue -- texture-path time of the 160 pieces that fill the ring --
    // before the first MFMA, and an epilogue of 5-14 k cycles during which the texture path idles.  The next tile's first two
    // chunks (A_0, W_0) are therefore issued in front of the epilogue, whose scratch lives in chunks 2-4.
    const int ntm = (M + BM5 - 1) / BM5, ntiles = ntm * ntn;
    int m0 = 0, n0 = 0;
    // LDS-DMA: a wave-instruction fills 8 rows x 128 B; wave w moves row groups 4w .. 4w+3 of A and of W.
    // As BUFFER loads (resource = the tile's first row, 32-bit per-lane byte offset, the K-stage as the scalar offset):
    // tools/coissue_lab measured what a piece costs a SIMD that is streaming MFMAs -- global_load_lds (per-lane 64-bit
    // addresses) ~65 cycles, buffer_load ... lds ~5.  The stage of this kernel carried 16 pieces per SIMD: 750 of its 2800 cycles.
    unsigned va[4], vb[4];                                  // byte offsets of the lane's 16 bytes inside the tile's row panel
    __amdgpu_buffer_rsrc_t ra, rb;
    auto tile_setup = [&](int t) {
        int tm, tn;
        tile_of(t, ntiles, ntm, ntn, xcd_remap, tm, tn);
        m0 = tm * BM5, n0 = tn * BN5;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = (wave * 4 + i) * 8 + (lane >> 3);
            const int c = ((lane & 7) ^ ((r >> 1) & 7)) * 8;
            va[i] = (unsigned)(((long)min(r, M - 1 - m0) * lda + c) * 2);
            vb[i] = (unsigned)(((long)min(r, N - 1 - n0) * ldw + c) * 2);
        }
        ra = dma_rsrc(A + (long)m0 * lda);
        rb = dma_rsrc(W + (long)n0 * ldw);
    };
    tile_setup(blockIdx.x);
    // (live = false: a look-ahead piece past the end of K -- counted, written, not fetched: gemm_tile.h dma_rsrc_none)
    const __amdgpu_buffer_rsrc_t rnone = dma_rsrc_none(A);
    const bool refetch = (xcd_remap & 8) != 0;              // M324_XCD bit 3 (A/B): fetch the last stage again instead, as rounds 1-4 did
    auto issue2a = [&](int i0, int st, int pos, bool live = true) {
        unsigned char* d = ring_w + pos * CHUNK10 + wave * 4096 + i0 * 1024;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            dma_piece(live ? ra : rnone, d + i * 1024, va[i0 + i], (unsigned)(st * 128));
    };
    auto issue2b = [&](int i0, int st, int pos, bool live = true) {
        unsigned char* d = ring_w + pos * CHUNK10 + wave * 4096 + i0 * 1024;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            dma_piece(live ? rb : rnone, d + i * 1024, vb[i0 + i], (unsigned)(st * 128));
    };

    f32x16 acc[4][2];

    const int NS = K / 64;
    // fragment of k-step ks: chunk 2 ks + hi of the lane's row; the swizzle only touches chunk bits, so k-step ks is the
    // k-step-0 offset ^ (ks << 5)
    const int aoff = lds_off(wm * 128 + l31, hi), boff = lds_off(wn * 64 + l31, hi);
    bf16x8 fa[2][4], fb[2][2];
    auto load_frags = [&](int set, int pa, int pw, int ks) {
        const unsigned char* ba = smem + pa * CHUNK10;
        const unsigned char* bw = smem + pw * CHUNK10;
        const int x = ks << 5;
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[set][j] = *reinterpret_cast<const bf16x8*>(bw + ((boff + j * 4096) ^ x));
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[set][i] = *reinterpret_cast<const bf16x8*>(ba + ((aoff + i * 4096) ^ x));
    };
    auto mma8 = [&](int set) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[set][j], fa[set][i], acc[i][j], 0, 0, 0);
    };
#define M324_SG(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
    auto sched_kstep = [&]() {
        M324_SG(0x008, 1); M324_SG(0x100, 1); M324_SG(0x008, 1); M324_SG(0x100, 1);
        M324_SG(0x008, 1); M324_SG(0x100, 1); M324_SG(0x008, 1); M324_SG(0x100, 1);
        M324_SG(0x008, 1); M324_SG(0x100, 1); M324_SG(0x008, 1); M324_SG(0x100, 1);
        M324_SG(0x008, 1); M324_SG(0x020, 1); M324_SG(0x008, 1); M324_SG(0x020, 1);
    };

    // LayerNorm fold, consumer (gemm_tile.h ln_prefetch / ln_finish): a wave fetches -- unmerged table: and merges -- 32 of its row of
    // waves' 128 rows where registers are free: the FIRST tile's around the ring prologue (entries requested in front of the pieces,
    // merged behind them), the NEXT tile's behind this tile's epilogue, in front of the drain that ends it anyway.  Only (rstd,
    // -rstd mean) of one row and one colsum travel through the main loop (the accumulators leave no room for the entries: held
    // across it they spilled, and a scratch reload inside the loop waits out every LDS-DMA piece older than itself).
    const int ln_table = EP_WAVE_FLOATS - wn * ep_wave_floats(ACT);      // floats from this wave's scratch to its row of waves' table
    float2 ln_rs;
    float ln_cs;
    {
        LnPreT<4> first;
        ln_prefetch<ACT, 4, false>(ep, M, N, m0 + wm * 128, n0 + wn * 64, lane, first, wn, ln_table);
        // prologue of the FIRST tile: A_0, W_0 (chunks 0, 1); A_1 goes out in front of stage 0 like every later tile's, W_1 and A_2
        // inside stage 0.  (Rounds 1-4 issued the whole ring here and ran a peeled stage 0 without issues; hipcc's wait-count pass
        // answered every fragment read of such straight-line code behind an LDS-DMA issue with `s_waitcnt vmcnt(0)` -- a full drain at
        // the start of EVERY tile.  Inside the K loop it trusts the kernel's counted waits, so all K-stages now run there.)
        issue2a(0, 0, 0); issue2a(2, 0, 0);
        issue2b(0, 0, 1); issue2b(2, 0, 1);
        ln_finish<ACT, 4>(ep, lane, first);
        ln_rs = first.rs, ln_cs = first.cs;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // stage 0 landed
    M324_BARRIER();
    int pa = 0, pw = 1;                                     // ring positions of A_s, W_s
    auto stage = [&](int s, auto issue_tag) {
        constexpr bool ISSUE = decltype(issue_tag)::value;
        int pwn = pa + 3, pan = pa + 4;                     // positions of W_{s+1} (chunk 2s+3) and A_{s+2} (chunk 2s+4)
        pwn = pwn >= 5 ? pwn - 5 : pwn;
        pan = pan >= 5 ? pan - 5 : pan;
        const int sw = s + 1 < NS ? s + 1 : NS - 1, sa = s + 2 < NS ? s + 2 : NS - 1;
        const bool wl = s + 1 < NS || refetch, al = s + 2 < NS || refetch;
        load_frags(0, pa, pw, 0);
        if constexpr (ISSUE) issue2b(0, sw, pwn, wl);
        mma8(1);                                            // (s-1, k-step 3); zeros in the first iteration
        sched_kstep();
        load_frags(1, pa, pw, 1);
        if constexpr (ISSUE) issue2b(2, sw, pwn, wl);
        mma8(0);
        sched_kstep();
        load_frags(0, pa, pw, 2);
        if constexpr (ISSUE) issue2a(0, sa, pan, al);
        mma8(1);
        sched_kstep();
        load_frags(1, pa, pw, 3);
        if constexpr (ISSUE) issue2a(2, sa, pan, al);
        mma8(0);
        sched_kstep();
        asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        M324_BARRIER();
        pa = pa + 2 >= 5 ? pa - 3 : pa + 2;
        pw = pw + 2 >= 5 ? pw - 3 : pw + 2;
    };
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int mt = m0, nt = n0;                         // this tile's origin (m0 / n0 move on in front of the epilogue)
        const bool more = t + (int)gridDim.x < ntiles;
        LnPreT<4> ln_pre;
        ln_pre.rs = ln_rs, ln_pre.cs = ln_cs, ln_pre.table_off = ln_table, ln_pre.slot = 32 * wn;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[1][i] = (bf16x8)(0);
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[1][j] = (bf16x8)(0);
        pa = 0, pw = 1;
        {
            // A_0, W_0 are in place (first tile: the prologue; later tiles: they arrived under the previous epilogue); A_1 goes out
            // first, W_1 and A_2 with stage 0
            const int s1 = NS > 1 ? 1 : 0;
            issue2a(0, s1, 2); issue2a(2, s1, 2);
        }
        for (int s = 0; s < NS; ++s) stage(s, std::true_type{});
        mma8(1);                                            // (NS-1, k-step 3)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // no LDS-DMA may outlive the main loop: the ring becomes scratch
        M324_BARRIER();
        if (more) {                                         // the next tile's first chunks land under this tile's epilogue
            tile_setup(t + gridDim.x);
            issue2a(0, 0, 0); issue2a(2, 0, 0);
            issue2b(0, 0, 1); issue2b(2, 0, 1);
        }
        store_tile_lds<TOUT, ACT, RES, 4>(acc, scratch + wave * ep_wave_floats(ACT), C, ldc, M, N,
                                          mt + wm * 128, nt + wn * 64, lane, ep, &ln_pre);
        if (more) {
            LnPreT<4> next;                                 // m0 / n0 are the next tile's already
            ln_prefetch<ACT, 4>(ep, M, N, m0 + wm * 128, n0 + wn * 64, lane, next, wn, ln_table);
            ln_finish<ACT, 4>(ep, lane, next);
            ln_rs = next.rs, ln_cs = next.cs;
            // everything this wave has in flight -- the two prefetched chunks and the epilogue's stores (loads and stores share
            // vmcnt and may retire out of order with respect to each other) -- must be done before the next tile starts.  The
            // builtin, not inline asm: hipcc's own wait-count pass must see the drain (gemm_ring4.hip has the story).
            __builtin_amdgcn_s_waitcnt(0x0F70);             // vmcnt(0)
            M324_BARRIER();
        }
    }
#undef M324_SG
}

template <typename TOUT, int ACT, int RES>
__global__ __launch_bounds__(512, 2) void gemm_ring_kernel(const bf16_t* __restrict__ A, long lda, const bf16_t* __restrict__ W,
                                                           long ldw, TOUT* C, long ldc, int M, int N, int K, Epilogue ep, int ntn,
                                                           int xcd_remap) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[5 * CHUNK10];
    gemm_ring_body<TOUT, ACT, RES>(smem, smem, reinterpret_cast<float*>(smem + 2 * CHUNK10), A, lda, W, ldw, C, ldc, M, N, K, ep, ntn, xcd_remap);
}

// ------------------------------------------------------------------------------------------------
// v13: v2's 128 x 128 tile (two workgroups per CU) on v10's pipeline.  v2 double-buffers whole K-tiles behind one
// vmcnt(0) + barrier per tile: the texture path, which bounds these tiles (64 KiB of LDS-DMA per 4.2 MFLOP), idles while
// the workgroup drains and starts cold again after every barrier.  Here the 80 KiB a workgroup may use are a ring of
// FIVE 16-KiB chunks (one operand's 128 rows x 64 k of one K-stage; chunk 2s = A of stage s, 2s+1 = W of stage s), so
// 2.5 stages are resident, pieces are issued between the MFMAs of every k-step, the wait is a counted vmcnt(4), the
// MFMA work is rotated one k-step against the LDS stages and the prologue fills the whole ring -- exactly v10 with a
// 2 x 2 accumulator block per wave (a phase is 4 MFMAs, 4 fragment reads and 2 LDS-DMA pieces).
constexpr int CHUNK13 = 128 * ROWB;               // 16 KiB

// (the body of one workgroup = one tile; gemm_ring2_kernel runs it on its own grid, gemm_ring2_pair_kernel on one of two problems)
template <typename TOUT, int ACT, int RES>
__device__ __forceinline__ void ring2_tile(const bf16_t* __restrict__ A, long lda, const bf16_t* __restrict__ W, long ldw, TOUT* C, long ldc,
                                           int M, int N, int K, const Epilogue& ep, int ntn, int xcd_remap, int tile, int ntiles) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[5 * CHUNK13];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, hi = lane >> 5;
    int tm, tn;
    tile_of(tile, ntiles, (M + BM - 1) / BM, ntn, xcd_remap, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;

    // LDS-DMA: a wave-instruction fills 8 rows x 128 B; wave w moves row groups 4w .. 4w+3 of A and of W
    unsigned ga[4], gb[4];                                  // byte offsets inside the tile's row panels
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (wave * 4 + i) * 8 + (lane >> 3);
        const int c = ((lane & 7) ^ ((r >> 1) & 7)) * 8;
        ga[i] = (unsigned)(((long)min(r, M - 1 - m0) * lda + c) * 2);
        gb[i] = (unsigned)(((long)min(r, N - 1 - n0) * ldw + c) * 2);
    }
    const __amdgpu_buffer_rsrc_t ra = dma_rsrc(A + (long)m0 * lda), rb = dma_rsrc(W + (long)n0 * ldw);
    const __amdgpu_buffer_rsrc_t rnone = dma_rsrc_none(A);    // look-ahead pieces past the end of K: gemm_tile.h dma_rsrc_none
    const bool refetch = (xcd_remap & 8) != 0;
    auto issue2 = [&](const unsigned (&g)[4], int i0, int st, int pos, bool live = true) {
        unsigned char* d = smem + pos * CHUNK13 + wave * 4096 + i0 * 1024;
        const bool isa = &g[0] == &ga[0];
#pragma unroll
        for (int i = 0; i < 2; ++i) dma_piece(live ? (isa ? ra : rb) : rnone, d + i * 1024, g[i0 + i], (unsigned)(st * 128));
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int NS = K / 64;
    const int aoff = lds_off(wm * 64 + l31, hi), boff = lds_off(wn * 64 + l31, hi);
    bf16x8 fa[2][2], fb[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) { fa[1][i] = (bf16x8)(0); fb[1][i] = (bf16x8)(0); }
    auto load_frags = [&](int set, int pa, int pw, int ks) {
        const unsigned char* ba = smem + pa * CHUNK13;
        const unsigned char* bw = smem + pw * CHUNK13;
        const int x = ks << 5;
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[set][j] = *reinterpret_cast<const bf16x8*>(bw + ((boff + j * 4096) ^ x));
#pragma unroll
        for (int i = 0; i < 2; ++i) fa[set][i] = *reinterpret_cast<const bf16x8*>(ba + ((aoff + i * 4096) ^ x));
    };
    auto mma4 = [&](int set) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[set][j], fa[set][i], acc[i][j], 0, 0, 0);
    };
#define M324_SG(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
    auto sched_phase = [&]() {                               // M r M r M r r M G G
        M324_SG(0x008, 1); M324_SG(0x100, 1); M324_SG(0x008, 1); M324_SG(0x100, 1);
        M324_SG(0x008, 1); M324_SG(0x100, 2); M324_SG(0x008, 1); M324_SG(0x020, 2);
    };

    // fp32 residual stream with plain rows (the out-projections of the trunk and of DINO): the epilogue's residual values are requested
    // in front of the ring's first pieces (gemm_tile.h res_prefetch; 64 registers, 226 of the 256 two workgroups per CU leave a wave)
    constexpr bool PRE_RES = RES == 1 && sizeof(TOUT) == 4;
    ResPre<2> pres;
    const bool use_pres = PRE_RES && ep.residual != nullptr && (xcd_remap & 16) == 0;       // M324_XCD bit 4: A/B
    if constexpr (PRE_RES) {
        if (use_pres) res_prefetch<2>(ep, M, N, m0 + wm * 64, n0 + wn * 64, lane, pres);
    }
    LnPreT<2> ln_pre;
    ln_prefetch<ACT, 2>(ep, M, N, m0 + wm * 64, n0 + wn * 64, lane, ln_pre);
    {
        const int s1 = NS > 1 ? 1 : 0, s2 = NS > 2 ? 2 : NS - 1;
        issue2(ga, 0, 0, 0); issue2(ga, 2, 0, 0);
        issue2(gb, 0, 0, 1); issue2(gb, 2, 0, 1);
        issue2(ga, 0, s1, 2); issue2(ga, 2, s1, 2);
        issue2(gb, 0, s1, 3); issue2(gb, 2, s1, 3);
        issue2(ga, 0, s2, 4); issue2(ga, 2, s2, 4);
    }
    ln_finish<ACT, 2>(ep, lane, ln_pre);                    // the table entries are older than the ring's 20 pieces
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");      // stage 0 landed (A_1, W_1, A_2 may fly)
    M324_BARRIER();
    int pa = 0, pw = 1;
    auto stage = [&](int s, auto issue_tag) {
        constexpr bool ISSUE = decltype(issue_tag)::value;
        int pwn = pa + 3, pan = pa + 4;
        pwn = pwn >= 5 ? pwn - 5 : pwn;
        pan = pan >= 5 ? pan - 5 : pan;
        const int sw = s + 1 < NS ? s + 1 : NS - 1, sa = s + 2 < NS ? s + 2 : NS - 1;
        const bool wl = s + 1 < NS || refetch, al = s + 2 < NS || refetch;
        load_frags(0, pa, pw, 0);
        if constexpr (ISSUE) issue2(gb, 0, sw, pwn, wl);
        mma4(1);                                            // (s-1, k-step 3); zeros in the first iteration
        sched_phase();
        load_frags(1, pa, pw, 1);
        if constexpr (ISSUE) issue2(gb, 2, sw, pwn, wl);
        mma4(0);
        sched_phase();
        load_frags(0, pa, pw, 2);
        if constexpr (ISSUE) issue2(ga, 0, sa, pan, al);
        mma4(1);
        sched_phase();
        load_frags(1, pa, pw, 3);
        if constexpr (ISSUE) issue2(ga, 2, sa, pan, al);
        mma4(0);
        sched_phase();
        asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        M324_BARRIER();
        pa = pa + 2 >= 5 ? pa - 3 : pa + 2;
        pw = pw + 2 >= 5 ? pw - 3 : pw + 2;
    };
    stage(0, std::false_type{});
    for (int s = 1; s < NS; ++s) stage(s, std::true_type{});
    mma4(1);                                                // (NS-1, k-step 3)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // no LDS-DMA may outlive the main loop: the ring becomes scratch
#undef M324_SG
    M324_BARRIER();
    store_tile_lds<TOUT, ACT, RES, 2>(acc, reinterpret_cast<float*>(smem) + wave * ep_wave_floats(ACT), C, ldc, M, N, m0 + wm * 64,
                                      n0 + wn * 64, lane, ep, &ln_pre, PRE_RES ? &pres : nullptr, use_pres);
}

template <typename TOUT, int ACT, int RES>
__global__ __launch_bounds__(256, 2) void gemm_ring2_kernel(const bf16_t* __restrict__ A, long lda, const bf16_t* __restrict__ W,
                                                            long ldw, TOUT* C, long ldc, int M, int N, int K, Epilogue ep, int ntn,
                                                            int xcd_remap) {
    ring2_tile<TOUT, ACT, RES>(A, lda, W, ldw, C, ldc, M, N, K, ep, ntn, xcd_remap, (int)blockIdx.x, (int)gridDim.x);
}

// Two independent problems in one launch (m324_gemm_pair: the decoder's q and k|v projections -- 96 and 192 tiles, each a 12-stage
// latency chain at 2048 rows; two launches run them one after the other at the clock the MLP before them left behind).
struct Ring2Problem {
    const bf16_t* A; long lda;
    const bf16_t* W; long ldw;
    void* C; long ldc;
    int M, N, K;
    Epilogue ep;
    int ntn, xcd_remap, tiles;
};
template <typename TOUT, int ACT, int RES>
__global__ __launch_bounds__(256, 2) void gemm_ring2_pair_kernel(Ring2Problem p0, Ring2Problem p1) {
    if ((int)blockIdx.x < p0.tiles)
        ring2_tile<TOUT, ACT, RES>(p0.A, p0.lda, p0.W, p0.ldw, (TOUT*)p0.C, p0.ldc, p0.M, p0.N, p0.K, p0.ep, p0.ntn, p0.xcd_remap, (int)blockIdx.x, p0.tiles);
    else
        ring2_tile<TOUT, ACT, RES>(p1.A, p1.lda, p1.W, p1.ldw, (TOUT*)p1.C, p1.ldc, p1.M, p1.N, p1.K, p1.ep, p1.ntn, p1.xcd_remap,
                                   (int)blockIdx.x - p0.tiles, p1.tiles);
}

// ------------------------------------------------------------------------------------------------
// v9: skinny GEMM for M <= 64 (the 64 latent tokens of the shape encoder: every projection of the 4 point-transformer
// blocks and of the encoder cross-attention at B = 1).  A 128 x 128 tile kernel runs these on N / 128 = 6..24 CUs with the
// whole K loop serial (17-48 us for 0.1-0.3 GFLOP).  Here a workgroup owns 32 output columns, its 8 waves split K in
// interleaved 64-element chunks, operands go global -> registers directly in MFMA fragment order (no LDS: every
// element is used once per workgroup), and the eight partial 64 x 32 blocks are summed through LDS in a fixed order.
// k-assignment inside a chunk: lane half `hi` owns k = 32 hi .. 32 hi + 31 (four 16-byte fragments = four MFMA steps);
// A and W use the same assignment, so every k is contracted exactly once (only the summation order differs from the
// tile kernels).
constexpr int SK_LD = 36;      // floats per partial row: 32 + 4 pad (conflict-free b128 writes: 8 lanes = 8 rows)
constexpr int SK_WAVES = 8;

template <typename TOUT, int ACT>
__global__ __launch_bounds__(512) void gemm_skinny_kernel(const bf16_t* __restrict__ A, long lda, const bf16_t* __restrict__ W,
                                                          long ldw, TOUT* C, long ldc, int M, int N, int K, Epilogue ep) {
    __shared__ __attribute__((aligned(16))) float part[SK_WAVES * 64 * SK_LD];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int n0 = blockIdx.x * 32;
    const bf16_t* a0 = A + (long)min(l31, M - 1) * lda + hi * 32;
    const bf16_t* a1 = A + (long)min(32 + l31, M - 1) * lda + hi * 32;
    const bf16_t* w0 = W + (long)min(n0 + l31, N - 1) * ldw + hi * 32;
    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const int nchunk = K / 64;
    bf16x8 fa[3][2][4], fw[3][4];
    auto load = [&](int set, int c) {
        const long k = (long)c * 64;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            fw[set][s] = *reinterpret_cast<const bf16x8*>(w0 + k + s * 8);
            fa[set][0][s] = *reinterpret_cast<const bf16x8*>(a0 + k + s * 8);
            fa[set][1][s] = *reinterpret_cast<const bf16x8*>(a1 + k + s * 8);
        }
    };
    auto mma = [&](int set) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < 2; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[set][s], fa[set][i][s], acc[i], 0, 0, 0);
    };
    // three chunks of this wave in flight (the kernel is latency-bound: 24-96 workgroups stream all of W);
    // the trip is unrolled by 3 so that the register sets rotate without dynamic indexing
    constexpr int ST = SK_WAVES;
    int c = wave;
    if (c < nchunk) load(0, c);
    if (c + ST < nchunk) load(1, c + ST);
    for (; c < nchunk; c += 3 * ST) {
        if (c + 2 * ST < nchunk) load(2, c + 2 * ST);
        mma(0);
        if (c + ST < nchunk) {
            if (c + 3 * ST < nchunk) load(0, c + 3 * ST);
            mma(1);
        }
        if (c + 2 * ST < nchunk) {
            if (c + 4 * ST < nchunk) load(1, c + 4 * ST);
            mma(2);
        }
    }
    // partial blocks -> LDS ([wave][row][col], swapped accumulator layout: lane = row, registers = 4-column runs)
    float* mine = part + wave * 64 * SK_LD;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<float4*>(mine + (i * 32 + l31) * SK_LD + 8 * g + 4 * hi) =
                make_float4(acc[i][4 * g], acc[i][4 * g + 1], acc[i][4 * g + 2], acc[i][4 * g + 3]);
    __syncthreads();
    // thread t: row t / 8, columns (t % 8) * 4 .. + 3; fixed summation order wave 0..7
    const int row = tid >> 3, cc = (tid & 7) * 4;
    float4 x = *reinterpret_cast<const float4*>(part + row * SK_LD + cc);
#pragma unroll
    for (int w = 1; w < SK_WAVES; ++w) {
        const float4 p = *reinterpret_cast<const float4*>(part + (w * 64 + row) * SK_LD + cc);
        x.x += p.x; x.y += p.y; x.z += p.z; x.w += p.w;
    }
    const int n = n0 + cc;
    if (row >= M || n >= N) return;            // N % 4 == 0
    long orow = row;
    if (ep.row_gin > 0) orow = (long)(row / ep.row_gin) * ep.row_gout + (row % ep.row_gin) + ep.row_off;
    const int rrow = (ep.res_rows > 0 && ep.res_rows < M) ? row % ep.res_rows : row;
    if (ep.bias) {
        const float4 b = *reinterpret_cast<const float4*>(ep.bias + n);
        x.x += b.x; x.y += b.y; x.z += b.z; x.w += b.w;
    }
    if (ACT == 1) apply_gelu4<TOUT>(x);
    if (ep.gamma) {
        const float4 g = *reinterpret_cast<const float4*>(ep.gamma + n);
        x.x *= g.x; x.y *= g.y; x.z *= g.z; x.w *= g.w;
    }
    if (ep.residual) {
        const float4 r = ep.res_out ? load4_out<TOUT>(reinterpret_cast<const TOUT*>(ep.residual) + (long)rrow * ep.ldr + n)
                                    : *reinterpret_cast<const float4*>(ep.residual + (long)rrow * ep.ldr + n);
        x.x += r.x; x.y += r.y; x.z += r.z; x.w += r.w;
    }
    store4_out<TOUT>(C + orow * ldc + n, x.x, x.y, x.z, x.w);
}

// ------------------------------------------------------------------------------------------------
// TN kernel for weight gradients: C[n, j] = sum_m X[m, n] * Y[m, j]  (X = dY [M, N], Y = A [M, Kc], both token-major, as
// the forward / dgrad GEMMs leave them).  The contraction index m is the ROW index of both operands, so the MFMA
// fragments (8 consecutive m for one n) are columns of the LDS tile: they are fetched with gfx950's transposing LDS read
// (ds_read_b64_tr_b16: within a 16-lane group lane i supplies 8 bytes of row i >> 2, and lane c receives column c of
// the resulting 4 x 16 block -- measured with tools/tr_lab), two reads per fragment.  This removes the two transpose
// passes per Linear that the NN formulation needed (10 % of the training step).
//   tile 128 (n) x 128 (j), 64 tokens per stage; LDS stage = X[64][128] + Y[64][128] bf16 (256-byte rows), two stages;
//   an LDS-DMA piece = 4 rows x 256 B; slot of (row r, 16-byte chunk c) = c ^ 4 (r & 3): the four rows of a transposed
//   read and the two column blocks of a 32-lane access land in 8 disjoint bank groups;
//   grid.y = split-K slices over the tokens (partials summed by m324_colsum, deterministic); rows past the slice end are
//   zeroed in LDS (X only) before they are contracted.
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;
typedef __attribute__((address_space(3))) s16x4_t lds_s16x4_t;

__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(const bf16_t* __restrict__ X, long ldx, const bf16_t* __restrict__ Y,
                                                         long ldy, float* __restrict__ C, long ldc, int M, int N, int Kc, int ks,
                                                         long strideC, int ntj, int ntiles, int nslices, int xcd) {
    constexpr int TROW = 256;                    // bytes per LDS row (128 bf16)
    constexpr int TOP = 64 * TROW;               // one operand of a stage: 16 KiB
    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * 2 * TOP];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    int item = blockIdx.x;                       // flat grid of (slice, tile) items, a contiguous slice-major range per XCD: see tn_pipe_body
    if (xcd) {
        const int nb = gridDim.x, q = nb >> 3, r = nb & 7, x = item & 7, loc = item >> 3;
        item = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + loc;
    }
    const int slice = item / ntiles, tile = item - slice * ntiles;
    const int n0 = (tile / ntj) * 128, j0 = (tile % ntj) * 128;
    const int mbeg = slice * ks;
    const int mend = slice == nslices - 1 ? M : mbeg + ks;
    C += (long)slice * strideC;

    // staging: piece p = rows 4p .. 4p+3; wave w moves pieces 4w .. 4w+3 of X and of Y
    const int sr = lane >> 4, sc = (lane & 15) ^ (4 * sr);       // row within the piece, source chunk of this lane's slot
    const unsigned cx = (unsigned)(min(n0 + sc * 8, N - 8) * 2), cy = (unsigned)(min(j0 + sc * 8, Kc - 8) * 2);
    const __amdgpu_buffer_rsrc_t rx = dma_rsrc(X + (long)mbeg * ldx), ry = dma_rsrc(Y + (long)mbeg * ldy);
    auto issue = [&](int t, int stage) {
        unsigned char* st = smem + stage * 2 * TOP + wave * 4096;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const long row = min(t * 64 + (wave * 4 + p) * 4 + sr, mend - 1 - mbeg);      // row inside this slice of the tokens
            dma_piece(rx, st + p * 1024, (unsigned)(row * ldx * 2) + cx, 0);
            dma_piece(ry, st + TOP + p * 1024, (unsigned)(row * ldy * 2) + cy, 0);
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // transposed fragment reads: lane (group g = lane >> 4, i = lane & 15) supplies row (i >> 2) [+ 8 (g >> 1) + 4 rd + 16 ks]
    // and columns base + 16 (g & 1) + 4 (i & 3) .. + 3 of the wave's 32-column block
    const int g = lane >> 4, li = lane & 15;
    const int frow = (g >> 1) * 8 + (li >> 2);
    auto foff = [&](int colbase) {               // byte offset inside an operand tile of this lane's piece (rd = ks = 0)
        const int col = colbase + 16 * (g & 1) + 4 * (li & 3);
        return frow * TROW + (((col >> 3) ^ (4 * (li >> 2))) << 4) + ((col & 7) << 1);
    };
    int xo[2], yo[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) xo[b] = foff(wm * 64 + b * 32), yo[b] = TOP + foff(wn * 64 + b * 32);
    auto frag = [&](const unsigned char* st, int off, int ks16) {
        const unsigned char* p = st + off + ks16 * (16 * TROW);
        const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)p);
        const s16x4_t hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(p + 4 * TROW));
        const s16x8_t v = __builtin_shufflevector(lo, hi4, 0, 1, 2, 3, 4, 5, 6, 7);
        return __builtin_bit_cast(bf16x8, v);
    };

    const int nt = (mend - mbeg + 63) / 64;
    issue(0, 0);
    for (int t = 0; t < nt; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // tile t landed
        __syncthreads();
        if (t + 1 < nt) issue(t + 1, (t + 1) & 1);
        unsigned char* st = smem + (t & 1) * 2 * TOP;
        const int valid = mend - mbeg - t * 64;                 // rows of this tile inside the slice
        if (valid < 64) {                                       // last tile of the slice: zero the X rows past its end
            for (int e = tid; e < (64 - valid) * 16; e += 256)
                *reinterpret_cast<uint4*>(st + (valid + (e >> 4)) * TROW + ((e & 15) << 4)) = make_uint4(0, 0, 0, 0);
            __syncthreads();
        }
#pragma unroll
        for (int k16 = 0; k16 < 4; ++k16) {
            bf16x8 xf[2], yf[2];
#pragma unroll
            for (int b = 0; b < 2; ++b) xf[b] = frag(st, xo[b], k16), yf[b] = frag(st, yo[b], k16);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(yf[j], xf[i], acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();
    Epilogue ep{};
    store_tile_lds<float, 0, 0, 2>(acc, reinterpret_cast<float*>(smem) + wave * EP_WAVE_FLOATS, C, ldc, N, Kc, n0 + wm * 64,
                                   j0 + wn * 64, lane, ep);
}

// TN kernel, 256 x 256 tile on the v7 pipeline (used when the slices are whole 32-token half-tiles and the output is at
// least one 256 x 256 tile): half the LDS-DMA pieces per FLOP of the 128 x 128 kernel above, which is texture-path-bound.
// Ring slot = X[32 m][256 n] + Y[32 m][256 j] (512-byte rows, 16 KiB each); an LDS-DMA piece = 2 rows x 512 B; slot of
// (row r, 16-byte chunk c) = c ^ 4 (r & 3).  Per k-step a wave reads 6 fragments = 12 transposing reads for 8 MFMAs.
// The body sees the ring through THREE __restrict__ pointers (LDS-DMA destinations, fragment reads, epilogue scratch): the
// transposing read is an intrinsic with a memory operand, and hipcc's wait-count pass puts `s_waitcnt vmcnt(0)` in front of every LDS
// access that may touch the bytes of an LDS-DMA in flight -- twice per half-tile here, which drained the ring every iteration
// (rounds 1-4: 2320 cycles per half-tile for 1024 of MFMA, waves parked 74 % of their cycles, and a deeper ring changed nothing).
// With alias scopes it trusts the kernel's own counted waits and barriers, which order every slot's fill against its reads.
constexpr int TN_TROW = 512, TN_TOP = 32 * TN_TROW, TN_RING = 4;      // bytes per LDS row; one operand of a slot (16 KiB); slots
__device__ __forceinline__ void tn_pipe_body(unsigned char* __restrict__ ring_w, const unsigned char* __restrict__ ring_r,
                                             float* __restrict__ scratch, const bf16_t* __restrict__ X, long ldx,
                                             const bf16_t* __restrict__ Y, long ldy, float* __restrict__ C, long ldc, int M, int N, int Kc,
                                             int ks, long strideC, int ntj, int ntiles, int nslices, int xcd) {
    constexpr int TROW = TN_TROW, TOP = TN_TOP;
    constexpr int RING = TN_RING, PPW = 4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    // Flat grid of (slice, tile) items.  Workgroup i runs on XCD i % 8: every XCD takes a CONTIGUOUS range of the items (slice-major), so the
    // workgroups that walk one slice's token rows -- each 256-column panel of a slice is read by N / 256 or Kc / 256 of them -- meet in ONE
    // L2 instead of eight (round 6: with the (tile, slice) grid in dispatch order a slice's tiles sat on eight XCDs and every workgroup pulled
    // its panels from HBM / MALL: 290 MB instead of 95 for a 768 x 768 gradient of 31104 tokens; M324_XCD bit 0 = 0: the old order).
    int item = blockIdx.x;
    if (xcd) {
        const int nb = gridDim.x, q = nb >> 3, r = nb & 7, x = item & 7, loc = item >> 3;
        item = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + loc;
    }
    const int slice = item / ntiles, tile = item - slice * ntiles;
    const int n0 = (tile / ntj) * 256, j0 = (tile % ntj) * 256;
    const int mbeg = slice * ks;
    const int mend = slice == nslices - 1 ? M : mbeg + ks;      // (mend - mbeg) % 32 == 0 (host-checked)
    C += (long)slice * strideC;

    // staging: piece p = rows 2p, 2p+1 of the half-tile; wave w moves pieces 2w, 2w+1 (rows 4w .. 4w+3) of X and of Y
    const int sr = lane >> 5;                    // row within the piece
    unsigned gx[2], gy[2];                         // byte offsets inside this slice of the tokens
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int r = (wave * 2 + p) * 2 + sr;   // row within the half-tile
        const int c = (lane & 31) ^ (4 * (r & 3));
        gx[p] = (unsigned)(((long)r * ldx + min(n0 + c * 8, N - 8)) * 2);
        gy[p] = (unsigned)(((long)r * ldy + min(j0 + c * 8, Kc - 8)) * 2);
    }
    const __amdgpu_buffer_rsrc_t rx = dma_rsrc(X + (long)mbeg * ldx), ry = dma_rsrc(Y + (long)mbeg * ldy);
    auto issue_x = [&](int h, int slot) {
        unsigned char* st = ring_w + slot * 2 * TOP + wave * 2048;
#pragma unroll
        for (int p = 0; p < 2; ++p) dma_piece(rx, st + p * 1024, gx[p], (unsigned)((long)h * 32 * ldx * 2));
    };
    auto issue_y = [&](int h, int slot) {
        unsigned char* st = ring_w + slot * 2 * TOP + TOP + wave * 2048;
#pragma unroll
        for (int p = 0; p < 2; ++p) dma_piece(ry, st + p * 1024, gy[p], (unsigned)((long)h * 32 * ldy * 2));
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int g = lane >> 4, li = lane & 15;
    const int frow = (g >> 1) * 8 + (li >> 2);
    auto foff = [&](int colbase) {
        const int col = colbase + 16 * (g & 1) + 4 * (li & 3);
        return frow * TROW + (((col >> 3) ^ (4 * (li >> 2))) << 4) + ((col & 7) << 1);
    };
    int xo[4], yo[2];
#pragma unroll
    for (int b = 0; b < 4; ++b) xo[b] = foff(wm * 128 + b * 32);
#pragma unroll
    for (int b = 0; b < 2; ++b) yo[b] = TOP + foff(wn * 64 + b * 32);
    bf16x8 fx[2][4], fy[2][2];
    auto frag = [&](const unsigned char* p) {
        const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)p);
        const s16x4_t hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(p + 4 * TROW));
        const s16x8_t v = __builtin_shufflevector(lo, hi4, 0, 1, 2, 3, 4, 5, 6, 7);
        return __builtin_bit_cast(bf16x8, v);
    };
    auto load_frags = [&](int set, int slot, int k16) {
        const unsigned char* base = ring_r + slot * 2 * TOP + k16 * (16 * TROW);
#pragma unroll
        for (int b = 0; b < 2; ++b) fy[set][b] = frag(base + yo[b]);
#pragma unroll
        for (int b = 0; b < 4; ++b) fx[set][b] = frag(base + xo[b]);
    };
    auto mma8 = [&](int set) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fy[set][j], fx[set][i], acc[i][j], 0, 0, 0);
    };
#define M324_SG(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
    auto sched_kstep = [&]() {                   // 12 transposing reads on the first 6 MFMAs, 2 LDS-DMA pieces on the last two
        M324_SG(0x008, 1); M324_SG(0x100, 2); M324_SG(0x008, 1); M324_SG(0x100, 2);
        M324_SG(0x008, 1); M324_SG(0x100, 2); M324_SG(0x008, 1); M324_SG(0x100, 2);
        M324_SG(0x008, 1); M324_SG(0x100, 2); M324_SG(0x008, 1); M324_SG(0x100, 2);
        M324_SG(0x008, 1); M324_SG(0x020, 1); M324_SG(0x008, 1); M324_SG(0x020, 1);
    };
#define M324_WAIT_PIECES(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
    const int NH = (mend - mbeg) / 32;
    for (int h = 0; h < RING - 1; ++h) {
        issue_x(h < NH ? h : NH - 1, h);
        issue_y(h < NH ? h : NH - 1, h);
    }
    M324_WAIT_PIECES((RING - 2) * PPW);                 // half-tile 0 landed
    M324_BARRIER();
    load_frags(0, 0, 0);
    int slot = 0;
    for (int h = 0; h < NH; ++h) {
        const int nslot = slot + 1 == RING ? 0 : slot + 1, fslot = slot == 0 ? RING - 1 : slot - 1;     // fslot: half-tile h - 1's
        const int hn = h + RING - 1 < NH ? h + RING - 1 : NH - 1;
        M324_WAIT_PIECES((RING - 3) * PPW);             // half-tile h + 1 landed; RING - 3 later ones may fly
        M324_BARRIER();
        load_frags(1, slot, 1);
        issue_x(hn, fslot);
        mma8(0);
        sched_kstep();
        load_frags(0, nslot, 0);
        issue_y(hn, fslot);
        mma8(1);
        sched_kstep();
        slot = nslot;
    }
    M324_WAIT_PIECES(0);
#undef M324_WAIT_PIECES
#undef M324_SG
    M324_BARRIER();
    Epilogue ep{};
    store_tile_lds<float, 0, 0, 4>(acc, scratch + wave * EP_WAVE_FLOATS, C, ldc, N, Kc, n0 + wm * 128,
                                   j0 + wn * 64, lane, ep);
}

__global__ __launch_bounds__(512, 2) void gemm_tn_pipe_kernel(const bf16_t* __restrict__ X, long ldx, const bf16_t* __restrict__ Y,
                                                              long ldy, float* __restrict__ C, long ldc, int M, int N, int Kc,
                                                              int ks, long strideC, int ntj, int ntiles, int nslices, int xcd) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[TN_RING * 2 * TN_TOP];
    tn_pipe_body(smem, smem, reinterpret_cast<float*>(smem), X, ldx, Y, ldy, C, ldc, M, N, Kc, ks, strideC, ntj, ntiles, nslices, xcd);
}

// the vectorised epilogue of the LDS-DMA kernel needs 4-column runs to be addressable as float4 / uint2
static bool vec_ok(const m324_gemm_args* a) {
    const int osz = a->out_dtype == M324_BF16 ? 2 : 4;
    auto al = [](const void* p, int b) { return ((uintptr_t)p % b) == 0; };
    return a->N % 4 == 0 && a->N >= 4 && (a->ldc * osz) % (4 * osz) == 0 && al(a->C, 4 * osz) &&
           (!a->residual || (a->ldr % 4 == 0 && al(a->residual, 16))) && (!a->bias || al(a->bias, 16)) &&
           (!a->gamma || al(a->gamma, 16));
}

// Tile-order mode of tile_of(): M324_XCD bit 0 = contiguous range per XCD, bit 1 = the 4 x 2 group order for weights that
// do not fit an XCD's L2 beside the streaming A panels AND are wider than deep (q|k|v, fc1: measured HBM fetch -28 % / -13 %
// at equal time; fc2's K = 3072 A panels dominate its traffic and the group order re-fetches them: +35 %), bit 2 = force
// it (tests, lab), bit 3 = the ring kernels' look-ahead past the end of K fetches the last stage again instead of nothing (A/B).  Default 3.
static int xcd_mode(const m324_gemm_args* a) {
    const int t = m324::tunable(m324::TUN_XCD);
    const int old_refetch = t & (8 | 16);        // A/B bits handed through to the kernels: 3 = look-ahead pieces past the end of K fetch the
                                                 // last stage again (rounds 1-4); 4 = v12 without the residual prefetch (round 6)
    if (!(t & 1)) return old_refetch;
    const long wbytes = (long)a->N * a->K * (a->in_dtype == M324_BF16 ? 2 : 4);
    if ((t & 4) || ((t & 2) && wbytes > (5l << 19) && a->N >= 2 * a->K)) return 3 | old_refetch;
    return 1 | old_refetch;
}

// Kernel choice.  M324_GEMM=v1|v2|v5|v9|v10|v11|v12|v13|v15 forces a variant (A/B measurements, tests).  (v7, the half-tile ring
// the chunk-ring kernels replaced, was retired in round 3: no shape reaches it -- K is a multiple of 64 for bf16 -- and its
// A/B tables are kept in profiles/r01_ab_gemm_schedules.md.  v14, round 5's two persistent 256 x 128 workgroups per CU, was retired in
// round 6: v15 took every shape it was chosen for (41.4 against 44.2 us at 10368 x 2304 x 768); tables: profiles/r05_gemm_labs.md section 2.)
static int forced_variant() { return m324::tunable(m324::TUN_GEMM); }   // M324_GEMM at load / m324_set_tunable

// v10 is persistent: one 8-wave workgroup per CU (160 KiB of LDS each); M324_GEMM_PERSIST=0: one workgroup per tile (A/B)
static int ring_grid(long ntiles) {
    static const int n_cu = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
            n = 256;
        return n - (n & 7);                                 // a multiple of 8: tile t and t + grid share an XCD
    }();
    if (m324::tunable(m324::TUN_GEMM_PERSIST) == 0 || n_cu <= 0) return (int)ntiles;
    return (int)(ntiles < n_cu ? ntiles : n_cu);
}

static Epilogue make_epilogue(const m324_gemm_args* a);
// v15's epilogue code as gemm_hp.hip counts it: GELU bit 0, LayerNorm-fold consumer bit 3; -1: an epilogue it does not build
static int hp_act(const m324_gemm_args* a) {
    if (a->aux_mode != M324_AUX_NONE || (a->act != M324_ACT_NONE && a->act != M324_ACT_GELU)) return -1;
    return (a->act == M324_ACT_GELU ? 1 : 0) | (a->ln_rowstat ? 8 : 0);
}
static int hp_res(const m324_gemm_args* a) { return (!a->residual && a->row_gin <= 0) ? 0 : 1; }
static bool nbatch_one(const m324_gemm_args* a) { return a->batch <= 1; }

static int pick_variant(const m324_gemm_args* a) {
    if (!vec_ok(a)) return 1;
    int f = forced_variant();
    if (f == 7) f = 0;                           // retired schedule: the chooser decides
    if ((f == 1 || f == 5 || f == 9) && (a->ln_rowstat || a->ln_stats_out || a->ln_copy_out))
        f = 0;                                   // the LayerNorm fold is built into v2 / v10 / v11 / v12 / v13 only
    const bool bf16 = a->in_dtype == M324_BF16;
    const bool ring_ok = bf16 && a->K % 64 == 0 && a->K >= 128;      // v10 / v11: K-stages of 64, at least two
    if (f == 1 || f == 2 || f == 5) return f;
    if (f >= 10 && f <= 13) return ring_ok ? f : (f == 13 ? 2 : 5);
    if (f == 14) f = 0;                          // retired schedule: the chooser decides
    // v15 (gemm_hp.hip): the hand-placed K = 768 stream with the deferred epilogue; hp_ok lists what it takes
    if (f == 15) {
        if (ring_ok && nbatch_one(a) && m324::hp_ok(a, make_epilogue(a), hp_act(a), hp_res(a))) return 15;
        f = 0;
    }
    if (a->M <= 64 && bf16 && !a->aux_mode && (f == 0 || f == 9)) return 9;
    // 256 x 256 tiles halve the LDS-DMA traffic per FLOP: fastest whenever the column count quantises (N % 256 == 0)
    // and the tiles fill most of the 256 CUs in whole rounds; otherwise the 128 x 128 tiles of v2 (two workgroups per
    // CU) balance better.  Measured on the c2 shapes (tools/gemm_lab): the chunk-ring kernels win at >= 0.75 fill, v2
    // below; of the two, the 4-wave persistent v11 wins when the output is fp32 (residual epilogues: little VALU work,
    // 1.5x the LDS fragment traffic saved), the 8-wave v10 when it is bf16 (GELU / q|k|v epilogues want two waves per
    // SIMD).  K = 64 (a single K-stage) runs on the two-stage 256 x 256 kernel v5.
    // v15 (round 5, gemm_hp.hip): one wave per SIMD, hand-placed stream, the previous tile's epilogue between the MFMAs of the current
    // tile.  Measured on random data, same box, interleaved (profiles/r05_gemm_hp.md): plain 10368 x 2304 x 768 41.4 us against v13's
    // 48.8, v10's 50.2 (and 44.2 for round 5's retired two-workgroups-per-CU schedule) -- the training step 93.3 -> 92.1-92.7 ms (M324_HP bit 1, default); fc1 + GELU 10368 x 3072
    // 55.9 against v10's 57.8, 65536 x 3072 305.7 against 316.2: the chip is power-limited, a denser stream clocks lower, and with
    // the m324_rowstats_finish launch the stream needs in front of a folded consumer the clip does not move (bit 0, off).  With one
    // tile per workgroup nothing overlaps (the epilogue is the exposed tail): at least two tiles per CU.
    // M324_HP bit 2 (default): the GELU epilogues from 4096 tiles on -- the decoder's fc1 (65536 rows, nontemporal stores): its block
    // 0.770 -> 0.752 ms with the m324_rowstats_finish launch in front included (tools/block_lab.py, alternated processes)
    const long thp = (long)(a->N / 128) * ceil_div(a->M, BM5);
    const int hpm = m324::tunable(m324::TUN_HP);
    if (f == 0 && ring_ok && nbatch_one(a) && hpm != 0 && a->K == 768 && a->N % 128 == 0 && thp >= 512 && hp_act(a) >= 0 &&
        ((hp_act(a) & 1) ? ((hpm & 1) != 0 || ((hpm & 4) != 0 && thp >= 4096)) : (hpm & 2) != 0) &&
        m324::hp_ok(a, make_epilogue(a), hp_act(a), hp_res(a)))
        return 15;
    const long t5 = (long)ceil_div(a->N, BN5) * ceil_div(a->M, BM5);
    const double e5 = (double)t5 / (double)(((t5 + 255) / 256) * 256);
    if (a->N % BN5 == 0 && t5 >= 200 && e5 >= 0.75) {
        if (!bf16) return 5;
        if (ring_ok) return a->out_dtype == M324_F32 ? 11 : 10;
        return 5;
    }
    // narrow outputs with a long contraction (the MLP's fc2 at N = 768, K = 3072): 256 x 128 tiles fill the chip in one
    // round where 256 x 256 tiles cannot, with 3/4 of the 128 x 128 tiles' texture-path traffic (v12: 58 -> 51 us at
    // M = 10368, 55 -> 45 us at M = 8224); at K = 768 the two tie and v2 stays.
    const long t12 = (long)ceil_div(a->N, 128) * ceil_div(a->M, 256);
    const double e12 = (double)t12 / (double)(((t12 + 255) / 256) * 256);
    if (ring_ok && a->K >= 1024 && t12 >= 180 && e12 >= 0.70) return 12;
    // 128 x 128 tiles: the chunk-ring pipeline (v13) for the fp32 residual outputs at K = 768 (22.3 -> 21.4 us,
    // 20.2 -> 19.1 us; 10.35 -> 10.31 ms per clip)
    if (ring_ok && a->out_dtype == M324_F32 && !a->aux_mode) return 13;
    // ... and for small plain bf16 problems (the decoder's k|v and q projections, 2048 rows: <= one round of tiles, so a
    // tile's 12 K-stages are pure latency -- the ring's 2.5 stages of look-ahead: 11.4 -> 9.2 us, 10.5 -> 8.3 us)
    const bool heads = a->aux_mode == M324_AUX_QKV_HEADS || a->aux_mode == M324_AUX_QKV_HEADS_VT;   // cross-attention q / k|v
    if (ring_ok && (!a->aux_mode || heads) && (long)ceil_div(a->N, BN) * ceil_div(a->M, BM) <= 512) return 13;
    // bit 0 of M324_QKV_RING (default on): v13 for every fused q|k|v epilogue.  Alone v13 always ran the trunk's fused q|k|v GEMM
    // faster than v2 (46.3 us against 51.9, DINO 39.0 / 42.8); inside the clip it lost in round 3 (9.77 ms against 9.63) and wins
    // since the LayerNorm passes around it are folded (round 4, interleaved: 8.683 -> 8.618 ms; 256 frames 183.96 -> 183.77).
    if (ring_ok && heads && (m324::tunable(m324::TUN_QKV_RING) & 1) != 0) return 13;
    // bit 1 (A/B): v13 for every plain bf16 output as well (the training step's projections and dgrad GEMMs; microbench round 4:
    // M = 10368, N = 2304, K = 768 plain 46.9 us against v2's 50.9)
    if (ring_ok && !a->aux_mode && a->out_dtype == M324_BF16 && (m324::tunable(m324::TUN_QKV_RING) & 2) != 0) return 13;
    return 2;
}

template <typename TOUT, int ACT, int RES>
static int launch_pipe(const m324_gemm_args* a, hipStream_t s, const Epilogue& ep, int variant) {
    if (variant == 11 || variant == 12) return m324::launch_ring4(a, s, ep, ACT, RES, xcd_mode(a), variant);
    if (variant == 13) {
        hipLaunchKernelGGL((gemm_ring2_kernel<TOUT, ACT, RES>), dim3(ceil_div(a->N, BN) * ceil_div(a->M, BM)), dim3(256), 0, s,
                           (const bf16_t*)a->A, a->lda, (const bf16_t*)a->W, a->ldw, (TOUT*)a->C, a->ldc, a->M, a->N, a->K, ep,
                           ceil_div(a->N, BN), xcd_mode(a));
        return M324_OK;
    }
    if (variant == 10) {
        hipLaunchKernelGGL((gemm_ring_kernel<TOUT, ACT, RES>), dim3(ring_grid(ceil_div(a->N, BN5) * ceil_div(a->M, BM5))), dim3(512), 0, s,
                           (const bf16_t*)a->A, a->lda, (const bf16_t*)a->W, a->ldw, (TOUT*)a->C, a->ldc, a->M, a->N, a->K, ep,
                           ceil_div(a->N, BN5), xcd_mode(a));
        return M324_OK;
    }
    M324_FAIL(M324_ERR_UNSUPPORTED, "m324_gemm: no pipelined kernel for schedule %d", variant);
}

static Epilogue make_epilogue(const m324_gemm_args* a) {
    return Epilogue{a->bias, a->gamma, a->residual, a->ldr, a->res_rows, a->act, a->row_gin, a->row_gout, a->row_off,
                    a->strideA, a->strideW, a->strideC, a->aux, a->ldaux, a->aux_mode,
                    {(bf16_t*)a->qkv_q, (bf16_t*)a->qkv_k, (bf16_t*)a->qkv_v}, {a->qkv_qw, a->qkv_kw}, a->qkv_eps, a->qkv_qscale,
                    a->qkv_L, a->qkv_H, a->aux_mode == M324_AUX_QKV_HEADS_VT ? 1 : 0,
                    (a->out_dtype == M324_BF16 && (long)a->M * a->N * 2 > ((long)m324::tunable(m324::TUN_NT_MB) << 20)) ? 1 : 0,
                    (a->residual && (const void*)a->residual == (const void*)a->C && a->out_dtype == M324_BF16) ? 1 : 0,
                    reinterpret_cast<const float2*>(a->ln_rowstat), a->ln_colsum, reinterpret_cast<float2*>(a->ln_stats_out),
                    static_cast<bf16_t*>(a->ln_copy_out), a->ln_ldcopy, a->ln_rowstat ? a->ln_ncb : 0, a->ln_eps};
}

template <typename TIN, typename TOUT>
int launch(const m324_gemm_args* a, hipStream_t s) {
    const Epilogue ep = make_epilogue(a);
    dim3 grid(ceil_div(a->N, BN), ceil_div(a->M, BM));
    const int nbatch = a->batch > 1 ? a->batch : 1;
    const bool lnf = a->ln_rowstat || a->ln_stats_out || a->ln_copy_out;     // LayerNorm fold: the ACT | 8 instantiations
    if (a->aux_mode == M324_AUX_N3) {              // one schedule only: the 256 x 256 chunk ring (host-checked shape)
        if constexpr (sizeof(TOUT) == 2 && sizeof(TIN) == 2) {
            if (lnf)
                hipLaunchKernelGGL((gemm_ring_kernel<bf16_t, 13, 0>), dim3(ring_grid(ceil_div(a->N, BN5) * ceil_div(a->M, BM5))), dim3(512), 0, s,
                                   (const bf16_t*)a->A, a->lda, (const bf16_t*)a->W, a->ldw, (bf16_t*)a->C, a->ldc, a->M, a->N, a->K, ep,
                                   ceil_div(a->N, BN5), xcd_mode(a));
            else
                hipLaunchKernelGGL((gemm_ring_kernel<bf16_t, 5, 0>), dim3(ring_grid(ceil_div(a->N, BN5) * ceil_div(a->M, BM5))), dim3(512), 0, s,
                                   (const bf16_t*)a->A, a->lda, (const bf16_t*)a->W, a->ldw, (bf16_t*)a->C, a->ldc, a->M, a->N, a->K, ep,
                                   ceil_div(a->N, BN5), xcd_mode(a));
            M324_CHECK_LAUNCH("m324_gemm");
            return M324_OK;
        } else {
            M324_FAIL(M324_ERR_UNSUPPORTED, "m324_gemm: M324_AUX_N3 is a bf16 mode");
        }
    }
    const int variant = nbatch > 1 ? 2 : pick_variant(a);
    if (variant == 15) {
        const int rc_ = m324::launch_hp(a, s, ep, hp_act(a), hp_res(a), xcd_mode(a));
        if (rc_ != M324_OK) return rc_;
    } else if (variant == 1) {
        hipLaunchKernelGGL((gemm_kernel<TIN, TOUT>), grid, dim3(256), 0, s, (const TIN*)a->A, a->lda, (const TIN*)a->W,
                           a->ldw, (TOUT*)a->C, a->ldc, a->M, a->N, a->K, ep);
    } else if (variant == 9) {
        if (a->act == M324_ACT_GELU)
            hipLaunchKernelGGL((gemm_skinny_kernel<TOUT, 1>), dim3(ceil_div(a->N, 32)), dim3(512), 0, s, (const bf16_t*)a->A,
                               a->lda, (const bf16_t*)a->W, a->ldw, (TOUT*)a->C, a->ldc, a->M, a->N, a->K, ep);
        else
            hipLaunchKernelGGL((gemm_skinny_kernel<TOUT, 0>), dim3(ceil_div(a->N, 32)), dim3(512), 0, s, (const bf16_t*)a->A,
                               a->lda, (const bf16_t*)a->W, a->ldw, (TOUT*)a->C, a->ldc, a->M, a->N, a->K, ep);
    } else {
        const int res = !a->residual && a->row_gin <= 0 ? 0
                        : (a->residual && a->row_gin <= 0 && (a->res_rows <= 0 || a->res_rows >= a->M)) ? 1 : 2;
#define M324_GLDS(ACT, RES)                                                                                              \
    do {                                                                                                                 \
        if (variant >= 10) {                                                                             \
            const int rc_ = launch_pipe<TOUT, ACT, RES>(a, s, ep, variant);                                              \
            if (rc_ != M324_OK) return rc_;                                                                              \
        }                                                                                                                \
        else if (variant == 5)                                                                                           \
            hipLaunchKernelGGL((gemm_glds5_kernel<TIN, TOUT, ACT, RES>),                                                 \
                               dim3(ceil_div(a->N, BN5) * ceil_div(a->M, BM5)), dim3(512), 0, s, (const TIN*)a->A,       \
                               a->lda, (const TIN*)a->W, a->ldw, (TOUT*)a->C, a->ldc, a->M, a->N, a->K, ep,              \
                               ceil_div(a->N, BN5), xcd_mode(a));                                                        \
        else                                                                                                             \
            hipLaunchKernelGGL((gemm_glds_kernel<TIN, TOUT, ACT, RES>), dim3(grid.x * grid.y, nbatch), dim3(256), 0, s,  \
                               (const TIN*)a->A, a->lda, (const TIN*)a->W, a->ldw, (TOUT*)a->C, a->ldc, a->M, a->N,      \
                               a->K, ep, (int)grid.x, xcd_mode(a));                                                      \
    } while (0)
        // LayerNorm fold: bf16 operands on the chunk-ring / 128 x 128 LDS-DMA kernels only (m324_gemm checked the shape)
#define M324_LNF(ACT, RES)                                                                                               \
    do {                                                                                                                 \
        if (variant == 11 || variant == 12) {                                                                            \
            const int rc_ = m324::launch_ring4(a, s, ep, ACT, RES, xcd_mode(a), variant);                                \
            if (rc_ != M324_OK) return rc_;                                                                              \
        } else if (variant == 13)                                                                                        \
            hipLaunchKernelGGL((gemm_ring2_kernel<TOUT, ACT, RES>), dim3(grid.x * grid.y), dim3(256), 0, s,              \
                               (const bf16_t*)a->A, a->lda, (const bf16_t*)a->W, a->ldw, (TOUT*)a->C, a->ldc, a->M,      \
                               a->N, a->K, ep, (int)grid.x, xcd_mode(a));                                                \
        else if (variant == 10)                                                                                          \
            hipLaunchKernelGGL((gemm_ring_kernel<TOUT, ACT, RES>), dim3(ring_grid(ceil_div(a->N, BN5) * ceil_div(a->M, BM5))),      \
                               dim3(512), 0, s, (const bf16_t*)a->A, a->lda, (const bf16_t*)a->W, a->ldw, (TOUT*)a->C,   \
                               a->ldc, a->M, a->N, a->K, ep, ceil_div(a->N, BN5), xcd_mode(a));                          \
        else                                                                                                             \
            hipLaunchKernelGGL((gemm_glds_kernel<bf16_t, TOUT, ACT, RES>), dim3(grid.x * grid.y, 1), dim3(256), 0, s,    \
                               (const bf16_t*)a->A, a->lda, (const bf16_t*)a->W, a->ldw, (TOUT*)a->C, a->ldc, a->M,      \
                               a->N, a->K, ep, (int)grid.x, xcd_mode(a));                                                \
    } while (0)
        if (lnf) {                             // m324_gemm validated the combination: consumer XOR producer
            if constexpr (sizeof(TIN) == 2) {
                if (a->ln_rowstat) {               // consumer: ACT | 8, no residual
                    if (a->aux_mode == M324_AUX_QKV_HEADS || a->aux_mode == M324_AUX_QKV_HEADS_VT) {
                        if constexpr (sizeof(TOUT) == 2) M324_LNF(12, 0);
                    } else if (a->act == M324_ACT_GELU) {
                        if constexpr (sizeof(TOUT) == 2) M324_LNF(9, 0);
                    } else {
                        M324_LNF(8, 0);
                    }
                } else if constexpr (sizeof(TOUT) == 2) {      // producer, bf16 stream: statistics only
                    if (res == 1) M324_LNF(16, 1); else M324_LNF(16, 2);
                } else {                                        // producer, fp32 stream: statistics + bf16 twin
                    if (res == 1) M324_LNF(48, 1); else M324_LNF(48, 2);
                }
            }
        } else if (a->aux_mode == M324_AUX_QKV_HEADS || a->aux_mode == M324_AUX_QKV_HEADS_VT) {
            if constexpr (sizeof(TOUT) == 2) M324_GLDS(4, 0);
        } else if (a->aux_mode == M324_AUX_STORE_PREACT || a->aux_mode == M324_AUX_STORE_GELU_GRAD) {
            M324_GLDS(2, 0);
        } else if (a->aux_mode == M324_AUX_MUL_GELU_GRAD || a->aux_mode == M324_AUX_MUL) {
            M324_GLDS(3, 0);
        } else if (a->act == M324_ACT_GELU) {
            if (res == 0) M324_GLDS(1, 0); else M324_GLDS(1, 2);
        } else {
            if (res == 0) M324_GLDS(0, 0); else if (res == 1) M324_GLDS(0, 1); else M324_GLDS(0, 2);
        }
#undef M324_GLDS
#undef M324_LNF
    }
    M324_CHECK_LAUNCH("m324_gemm");
    return M324_OK;
}

}  // namespace


// Name (as rocprofv3's kernel trace prints the template, element types abbreviated) and grid in threads of the kernel
// m324_gemm would launch for `a`: lets bench.py label its HIP-event rows with the same symbols the committed rocprof
// summaries use.  Host-only; mirrors launch() above.
extern "C" int m324_gemm_plan(const m324_gemm_args* a, char* buf, int n) {
    M324_REQUIRE(a && buf && n > 0, "m324_gemm_plan: bad arguments");
    const int nbatch = a->batch > 1 ? a->batch : 1;
    const bool lnf = a->ln_rowstat || a->ln_stats_out || a->ln_copy_out;
    if (a->aux_mode == M324_AUX_N3) {
        snprintf(buf, (size_t)n, "gemm_ring_kernel<unsigned short, %d, 0> grid=%ldx1x1", lnf ? 13 : 5,
                 (long)ring_grid((long)ceil_div(a->N, BN5) * ceil_div(a->M, BM5)) * 512);
        return 10;
    }
    const int variant = nbatch > 1 ? 2 : pick_variant(a);
    const char* tout = a->out_dtype == M324_BF16 ? "unsigned short" : "float";
    const char* tin = a->in_dtype == M324_BF16 ? "unsigned short" : "float";
    const int res = !a->residual && a->row_gin <= 0 ? 0
                    : (a->residual && a->row_gin <= 0 && (a->res_rows <= 0 || a->res_rows >= a->M)) ? 1 : 2;
    int act = a->act == M324_ACT_GELU ? 1 : 0, rs = act ? (res ? 2 : 0) : res;
    if (a->aux_mode == M324_AUX_QKV_HEADS || a->aux_mode == M324_AUX_QKV_HEADS_VT) act = 4, rs = 0;
    else if (a->aux_mode == M324_AUX_STORE_PREACT || a->aux_mode == M324_AUX_STORE_GELU_GRAD) act = 2, rs = 0;
    else if (a->aux_mode == M324_AUX_MUL_GELU_GRAD || a->aux_mode == M324_AUX_MUL) act = 3, rs = 0;
    if (lnf) {
        if (a->ln_rowstat) act |= 8, rs = 0;
        else act = a->out_dtype == M324_F32 ? 48 : 16, rs = res;
    }
    long wg = 0, threads = 256;
    const char* name = "gemm_kernel";
    switch (variant) {
        case 1: wg = (long)ceil_div(a->N, BN) * ceil_div(a->M, BM); break;
        case 2: name = "gemm_glds_kernel"; wg = (long)ceil_div(a->N, BN) * ceil_div(a->M, BM); break;
        case 5: name = "gemm_glds5_kernel"; wg = (long)ceil_div(a->N, BN5) * ceil_div(a->M, BM5); threads = 512; break;
        case 9: name = "gemm_skinny_kernel"; wg = ceil_div(a->N, 32); threads = 512; break;
        case 10: name = "gemm_ring_kernel"; wg = ring_grid((long)ceil_div(a->N, BN5) * ceil_div(a->M, BM5)); threads = 512; break;
        case 11: name = "gemm_ring4_kernel"; wg = (long)ceil_div(a->N, BN5) * ceil_div(a->M, BM5); if (wg > 256) wg = 256; break;
        case 12: name = "gemm_ring3_kernel"; wg = (long)ceil_div(a->N, 128) * ceil_div(a->M, BM5); break;
        case 13: name = "gemm_ring2_kernel"; wg = (long)ceil_div(a->N, BN) * ceil_div(a->M, BM); break;
        case 15: name = "gemm_hp_kernel"; wg = m324::hp_grid(a); break;
        default: break;
    }
    // grid in threads, as rocprofv3's kernel trace prints it (x, y, z)
    if (variant == 1)
        snprintf(buf, (size_t)n, "%s<%s, %s> grid=%dx%dx1", name, tin, tout, ceil_div(a->N, BN) * 256, ceil_div(a->M, BM));
    else if (variant == 9) snprintf(buf, (size_t)n, "%s<%s, %d> grid=%ldx1x1", name, tout, act ? 1 : 0, wg * threads);
    else if (variant == 2) snprintf(buf, (size_t)n, "%s<%s, %s, %d, %d> grid=%ldx%dx1", name, tin, tout, act, rs, wg * threads, nbatch);
    else if (variant == 5) snprintf(buf, (size_t)n, "%s<%s, %s, %d, %d> grid=%ldx1x1", name, tin, tout, act, rs, wg * threads);
    else snprintf(buf, (size_t)n, "%s<%s, %d, %d> grid=%ldx1x1", name, tout, act, rs, wg * threads);
    return variant;
}

extern "C" int m324_gemm_tn(const void* X, long ldx, const void* Y, long ldy, float* C, long ldc, int M, int N, int Kc,
                            int slices, long strideC, void* stream) {
    M324_REQUIRE(X && Y && C, "m324_gemm_tn: null pointer");
    M324_REQUIRE(M > 0 && N >= 8 && Kc >= 8 && N % 8 == 0 && Kc % 8 == 0, "m324_gemm_tn: bad sizes M=%d N=%d Kc=%d", M, N, Kc);
    M324_REQUIRE(ldx >= N && ldy >= Kc && ldx % 8 == 0 && ldy % 8 == 0 && ldc >= Kc && ldc % 4 == 0 &&
                     ((uintptr_t)X % 16) == 0 && ((uintptr_t)Y % 16) == 0 && ((uintptr_t)C % 16) == 0,
                 "m324_gemm_tn: operands must be 16-byte aligned with leading dimensions that are multiples of 8");
    M324_REQUIRE(slices >= 1 && slices <= 65535 && (slices == 1 || strideC >= (long)N * ldc), "m324_gemm_tn: bad slicing");
    const int ks = ((M + slices - 1) / slices + 63) / 64 * 64;          // tokens per slice, whole 64-row tiles
    M324_REQUIRE((long)ks * (slices - 1) < M, "m324_gemm_tn: %d slices leave an empty slice for M=%d", slices, M);
    // the LDS-DMA pieces address a slice's rows with 32-bit byte offsets from its first row (buffer loads, 2 GiB window)
    M324_REQUIRE(((long)ks + 64) * (ldx > ldy ? ldx : ldy) * 2 < 0x7FFFFFFFl,
                 "m324_gemm_tn: a slice of %d tokens x leading dimension %ld exceeds the 2 GiB the staging offsets cover; use more slices",
                 ks, ldx > ldy ? ldx : ldy);
    // 256 x 256 tiles on the pipelined kernel when every slice is whole 32-token half-tiles and the output fills the tiles
    // (M324_GEMM_TN=128 forces the small kernel)
    const bool big = m324::tunable(m324::TUN_GEMM_TN) != 128 && M % 32 == 0 && N % 256 == 0 && Kc % 256 == 0 && M / slices >= 256;
    if (big) {
        const int ntj = Kc / 256;
        const int ntiles = (N / 256) * ntj;
        hipLaunchKernelGGL(gemm_tn_pipe_kernel, dim3(ntiles * slices), dim3(512), 0, (hipStream_t)stream, (const bf16_t*)X, ldx,
                           (const bf16_t*)Y, ldy, C, ldc, M, N, Kc, ks, strideC, ntj, ntiles, slices, m324::tunable(m324::TUN_XCD) & 1);
    } else {
        const int ntj = ceil_div(Kc, 128), ntiles = ceil_div(N, 128) * ntj;
        hipLaunchKernelGGL(gemm_tn_kernel, dim3(ntiles * slices), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)X, ldx,
                           (const bf16_t*)Y, ldy, C, ldc, M, N, Kc, ks, strideC, ntj, ntiles, slices, m324::tunable(m324::TUN_XCD) & 1);
    }
    M324_CHECK_LAUNCH("m324_gemm_tn");
    return M324_OK;
}

static int gemm_validate(const m324_gemm_args* a) {
    const bool qkv_mode = a && (a->aux_mode == M324_AUX_QKV_HEADS || a->aux_mode == M324_AUX_QKV_HEADS_VT);
    const bool n3_mode = a && a->aux_mode == M324_AUX_N3;
    M324_REQUIRE(a && a->A && a->W && (a->C || qkv_mode || n3_mode), "m324_gemm: null pointer");
    M324_REQUIRE(a->M > 0 && a->N > 0 && a->K > 0, "m324_gemm: empty problem M=%d N=%d K=%d", a->M, a->N, a->K);
    const int bk = a->in_dtype == M324_BF16 ? 64 : 32;
    M324_REQUIRE(a->K % bk == 0, "m324_gemm: K=%d must be a multiple of %d", a->K, bk);
    const int esz = a->in_dtype == M324_BF16 ? 2 : 4;
    M324_REQUIRE((a->lda * esz) % 16 == 0 && (a->ldw * esz) % 16 == 0 && ((uintptr_t)a->A % 16) == 0 &&
                     ((uintptr_t)a->W % 16) == 0,
                 "m324_gemm: A/W rows must be 16-byte aligned");
    M324_REQUIRE(a->lda >= a->K && a->ldw >= a->K && a->ldc >= a->N, "m324_gemm: leading dimension too small");
    // LDS-DMA pieces address a tile's rows with 32-bit byte offsets from the tile's first row (buffer loads, 2 GiB window)
    M324_REQUIRE(257l * (a->lda > a->ldw ? a->lda : a->ldw) * esz + (long)a->K * esz < 0x7FFFFFFFl,
                 "m324_gemm: leading dimension %ld too large for the 256-row staging window", a->lda > a->ldw ? a->lda : a->ldw);
    M324_REQUIRE(!a->residual || a->ldr >= a->N, "m324_gemm: ldr too small");
    M324_REQUIRE(a->batch <= 1 || (vec_ok(a) && !a->residual && a->batch <= 65535),
                 "m324_gemm: a batched launch needs a vectorisable, residual-free problem");
    M324_REQUIRE(a->aux_mode >= 0 && a->aux_mode <= M324_AUX_MUL, "m324_gemm: aux_mode %d", a->aux_mode);
    if (n3_mode) {
        M324_REQUIRE(a->aux && a->qkv_qw && a->act == M324_ACT_GELU && !a->residual && !a->gamma && a->row_gin <= 0 && a->batch <= 1,
                     "m324_gemm: M324_AUX_N3 = Linear + GELU + [3, N] contraction: aux (partial sums) and qkv_qw (the [3, N] weight) "
                     "must be given, no residual / gamma / row map / batch");
        M324_REQUIRE(a->in_dtype == M324_BF16 && a->out_dtype == M324_BF16 && a->N % 256 == 0 && a->K % 64 == 0 && a->K >= 128,
                     "m324_gemm: M324_AUX_N3 needs bf16, N %% 256 == 0, K %% 64 == 0, K >= 128 (N=%d K=%d)", a->N, a->K);
        M324_REQUIRE(((uintptr_t)a->qkv_qw % 16) == 0 && (!a->bias || ((uintptr_t)a->bias % 16) == 0), "m324_gemm: M324_AUX_N3 operands misaligned");
    }
    if (qkv_mode) {
        // Vt has no padding and a 32-token block of the epilogue must stay inside one batch entry: L % 64 == 0
        M324_REQUIRE(a->aux_mode != M324_AUX_QKV_HEADS_VT || a->qkv_L % 64 == 0,
                     "m324_gemm: M324_AUX_QKV_HEADS_VT needs qkv_L %% 64 == 0 (L=%d)", a->qkv_L);
        // q|k|v, k|v (qkv_q NULL) or q alone (qkv_k and qkv_v NULL): N = number of parts x H x 64
        const int parts = (a->qkv_q ? 1 : 0) + (a->qkv_k ? 1 : 0) + (a->qkv_v ? 1 : 0);
        const bool shape_ok = (a->qkv_q && a->qkv_k && a->qkv_v) || (!a->qkv_q && a->qkv_k && a->qkv_v) || (a->qkv_q && !a->qkv_k && !a->qkv_v);
        M324_REQUIRE(shape_ok && a->qkv_H > 0 && a->N == parts * a->qkv_H * 64,
                     "m324_gemm: M324_AUX_QKV_HEADS outputs must be q|k|v, k|v or q with N = parts * H * 64 (N=%d H=%d parts=%d)", a->N,
                     a->qkv_H, parts);
        M324_REQUIRE(a->in_dtype == M324_BF16 && a->out_dtype == M324_BF16 &&
                         a->qkv_L > 0 && a->M % a->qkv_L == 0 && !a->residual && !a->gamma &&
                         a->act == M324_ACT_NONE && a->row_gin <= 0 && a->batch <= 1 && a->M > 64 && vec_ok(a),
                     "m324_gemm: M324_AUX_QKV_HEADS needs a plain bf16 [B*L, 3*H*64] projection with M > 64");
        M324_REQUIRE(a->aux_mode != M324_AUX_QKV_HEADS_VT || a->qkv_v, "m324_gemm: M324_AUX_QKV_HEADS_VT without a V output");
        M324_REQUIRE(((uintptr_t)a->qkv_q % 16) == 0 && ((uintptr_t)a->qkv_k % 16) == 0 && ((uintptr_t)a->qkv_v % 16) == 0 &&
                         (!a->qkv_qw || ((uintptr_t)a->qkv_qw % 16) == 0) && (!a->qkv_kw || ((uintptr_t)a->qkv_kw % 16) == 0),
                     "m324_gemm: misaligned qkv outputs / norm weights");
    } else if (a->aux_mode && !n3_mode) {
        const int osz = a->out_dtype == M324_BF16 ? 2 : 4;
        M324_REQUIRE(a->aux && a->ldaux >= a->N && a->ldaux % 4 == 0 && ((uintptr_t)a->aux % (4 * osz)) == 0 && vec_ok(a) &&
                         a->row_gin <= 0 && a->batch <= 1,
                     "m324_gemm: aux operand needs a vectorisable, un-remapped, un-batched problem with ldaux >= N");
        M324_REQUIRE((a->aux_mode != M324_AUX_STORE_PREACT && a->aux_mode != M324_AUX_STORE_GELU_GRAD) || a->act == M324_ACT_GELU,
                     "m324_gemm: M324_AUX_STORE_PREACT / M324_AUX_STORE_GELU_GRAD only make sense with an activation");
        M324_REQUIRE((a->aux_mode != M324_AUX_MUL_GELU_GRAD && a->aux_mode != M324_AUX_MUL) || a->act == M324_ACT_NONE,
                     "m324_gemm: M324_AUX_MUL_GELU_GRAD / M324_AUX_MUL exclude an activation");
        M324_REQUIRE(!a->residual, "m324_gemm: the aux operand excludes a residual");
    }
    if (a->ln_rowstat || a->ln_stats_out || a->ln_copy_out) {
        // LayerNorm fold: the tile kernels' LDS-transposed epilogue only (not the skinny or the scalar-store kernel)
        M324_REQUIRE(a->in_dtype == M324_BF16 && a->N % 64 == 0 && a->M > 64 && vec_ok(a) && a->batch <= 1 && a->K >= 128,
                     "m324_gemm: the LayerNorm fold needs a bf16, vectorisable, un-batched problem with M > 64, N %% 64 == 0 and "
                     "K >= 128 (M=%d N=%d K=%d)", a->M, a->N, a->K);
        M324_REQUIRE(a->aux_mode != M324_AUX_STORE_PREACT && a->aux_mode != M324_AUX_MUL_GELU_GRAD && a->aux_mode != M324_AUX_STORE_GELU_GRAD &&
                         a->aux_mode != M324_AUX_MUL,
                     "m324_gemm: the LayerNorm fold is an inference feature (no training aux modes)");
        // the combinations that are built (gemm_tile.h store_tile_lds, ACTX bits): a consumer has no residual / gamma / row
        // map (and a bf16 output behind GELU); a producer is x = residual + A W^T (+ bias, gamma) with its statistics, plus
        // the bf16 twin exactly when x is fp32
        if (a->ln_rowstat) {
            M324_REQUIRE(!a->ln_stats_out && !a->ln_copy_out, "m324_gemm: one GEMM either consumes or produces LayerNorm statistics");
            M324_REQUIRE(!a->residual && !a->gamma && a->row_gin <= 0, "m324_gemm: a folded LayerNorm consumer takes no residual / gamma / row map");
            M324_REQUIRE(a->act == M324_ACT_NONE || a->out_dtype == M324_BF16, "m324_gemm: LayerNorm fold + GELU needs a bf16 output");
        } else {
            M324_REQUIRE(a->ln_stats_out && a->residual && a->act == M324_ACT_NONE,
                         "m324_gemm: a LayerNorm statistics producer is a residual update without activation");
            M324_REQUIRE((a->out_dtype == M324_F32) == (a->ln_copy_out != nullptr),
                         "m324_gemm: ln_copy_out goes with an fp32 output (and only with it)");
        }
        M324_REQUIRE(!a->ln_rowstat || (a->ln_colsum && ((uintptr_t)a->ln_rowstat % 8) == 0 && ((uintptr_t)a->ln_colsum % 16) == 0),
                     "m324_gemm: ln_rowstat needs ln_colsum (16-byte aligned) and an 8-byte aligned row table");
        M324_REQUIRE(!a->ln_rowstat || a->ln_ncb <= 0 || (a->ln_ncb <= 16 && a->ln_ncb % 2 == 0 && a->ln_ncb * 64 == a->K && a->ln_eps >= 0.f &&
                                                           (long)a->ln_ncb * a->M * 8 < (1l << 32)),
                     "m324_gemm: an unmerged row table has K / 64 blocks of 64 columns, an even count <= 16, and less than 4 GiB "
                     "(ln_ncb=%d K=%d M=%d)", a->ln_ncb, a->K, a->M);
        M324_REQUIRE((!a->ln_stats_out && !a->ln_copy_out) || (a->row_gin <= 0 && a->aux_mode == M324_AUX_NONE),
                     "m324_gemm: ln_stats_out / ln_copy_out describe the rows of C: no row remap, no aux mode");
        M324_REQUIRE(!a->ln_stats_out || ((uintptr_t)a->ln_stats_out % 8) == 0, "m324_gemm: ln_stats_out misaligned");
        M324_REQUIRE(!a->ln_copy_out || (a->out_dtype == M324_F32 && a->ln_ldcopy >= a->N && a->ln_ldcopy % 4 == 0 &&
                                         ((uintptr_t)a->ln_copy_out % 8) == 0),
                     "m324_gemm: ln_copy_out is the bf16 twin of an fp32 output (ldcopy >= N, multiple of 4, 8-byte aligned)");
    }
    return M324_OK;
}

extern "C" int m324_gemm(const m324_gemm_args* a, void* stream) {
    const int rc = gemm_validate(a);
    if (rc != M324_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    if (a->in_dtype == M324_BF16 && a->out_dtype == M324_BF16) return launch<bf16_t, bf16_t>(a, s);
    if (a->in_dtype == M324_BF16 && a->out_dtype == M324_F32) return launch<bf16_t, float>(a, s);
    if (a->in_dtype == M324_F32 && a->out_dtype == M324_F32) return launch<float, float>(a, s);
    M324_FAIL(M324_ERR_UNSUPPORTED, "m324_gemm: unsupported dtype pair in=%d out=%d", a->in_dtype, a->out_dtype);
}

// Two GEMMs in ONE launch (horizontal fusion).  Built for the pair that needs it: two bf16 projections with the head-major q|k|v
// epilogue that the chooser sends to the 128 x 128 chunk ring (the decoder's q and k|v projections).  Any other pair:
// M324_ERR_UNSUPPORTED and nothing launched -- the caller issues two m324_gemm calls.
extern "C" int m324_gemm_pair(const m324_gemm_args* a, const m324_gemm_args* b, void* stream) {
    int rc = gemm_validate(a);
    if (rc != M324_OK) return rc;
    rc = gemm_validate(b);
    if (rc != M324_OK) return rc;
    auto heads = [](const m324_gemm_args* g) {
        return (g->aux_mode == M324_AUX_QKV_HEADS || g->aux_mode == M324_AUX_QKV_HEADS_VT) && g->in_dtype == M324_BF16 && g->out_dtype == M324_BF16 &&
               !g->residual && !g->ln_rowstat && !g->ln_stats_out && !g->ln_copy_out && g->batch <= 1 && g->act == M324_ACT_NONE;
    };
    if (!heads(a) || !heads(b) || pick_variant(a) != 13 || pick_variant(b) != 13)
        M324_FAIL(M324_ERR_UNSUPPORTED, "m324_gemm_pair: built for two bf16 head-major projections on the 128 x 128 chunk ring");
    auto problem = [](const m324_gemm_args* g) {
        return Ring2Problem{(const bf16_t*)g->A, g->lda, (const bf16_t*)g->W, g->ldw, g->C, g->ldc, g->M, g->N, g->K, make_epilogue(g),
                            ceil_div(g->N, BN), xcd_mode(g), ceil_div(g->N, BN) * ceil_div(g->M, BM)};
    };
    const Ring2Problem p0 = problem(a), p1 = problem(b);
    hipLaunchKernelGGL((gemm_ring2_pair_kernel<bf16_t, 4, 0>), dim3(p0.tiles + p1.tiles), dim3(256), 0, (hipStream_t)stream, p0, p1);
    M324_CHECK_LAUNCH("m324_gemm_pair");
    return M324_OK;
}

