// m324_attention: flash-style softmax(Q K^T * scale) V for head_dim 64 on gfx950 matrix cores.
//
// Operands arrive head-major from m324_qkv_split: Q[Bq,H,Lq,64], K[B,H,Lk,64], Vt[B,H,64,Lkp]
// (V transposed, zero padded to a multiple of 64 keys).  A workgroup = 4 waves = 128 query rows of
// one (batch, head); each wave owns 32 query rows and walks the keys in tiles of 64.
//
// "Swapped" formulation -- no cross-lane shuffle and no P round trip through LDS:
//   S^T[kv, q] = K[kv, :] . Q[q, :]        MFMA A = K rows (from LDS), B = Q rows (registers)
//       C layout: lane holds column q = lane & 31, rows kv = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
//       -> a query's scores live in ONE lane pair (l, l ^ 32): row max / sum are in-register
//          reductions plus a single cross-half exchange.
//   O^T[d, q]  = Vt[d, kv] . P^T[kv, q]    MFMA A = Vt rows (from LDS), B = P^T = the lane's own S^T
//       registers (the contraction index kv may be visited in any order as long as A and B agree, so
//       Vt fragments are simply gathered in the C-layout's kv order: two 8-byte LDS reads per MFMA).
// Vt key order: within every aligned group of 16 keys the columns are stored as 0-3, 8-11, 4-7, 12-15
// (m324_qkv_split writes them so): that is the order in which the swapped MFMA contracts keys, so a lane's
// 8-key A fragment is ONE contiguous 16-byte chunk.
// LDS per stage: K tile [64 kv][64 d] and Vt tile [64 d][64 kv], rows of 128 bytes (bf16), 16-byte chunks
// XOR-swizzled (chunk ^ ((row >> 1) & 7)) so fragment reads are conflict-free ds_read_b128.  Tiles are
// staged by LDS-DMA (buffer_load_dwordx4 ... lds, swizzle applied to the source offset) into a 3-stage ring:
// two tiles in flight, counted vmcnt + raw s_barrier, no ds_write pass and no staging registers.
// The fp32 parity variant uses v_mfma_f32_32x32x2_f32 on padded fp32 tiles (single buffered).
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>
#include "common.h"

namespace {

constexpr int QW = 32;          // query rows per wave
constexpr int NW = 4;           // waves per workgroup
constexpr int QB = QW * NW;     // 128 query rows per workgroup
constexpr int KV = 64;          // keys per tile
constexpr float LOG2E = 1.4426950408889634f;

// ------------------------------------------------------------------------------------------- bf16
__device__ __forceinline__ int k_off(int row, int c16) { return row * 128 + ((c16 ^ ((row >> 1) & 7)) << 4); }
typedef __attribute__((address_space(3))) void lds_ptr_t;
typedef __attribute__((address_space(1))) const void glb_ptr_t;
constexpr int ASTAGE = 16384;   // K tile (8 KiB) + Vt tile (8 KiB)

// LDS-DMA pieces are BUFFER loads: resource = the (batch, head)'s operand, per-lane 32-bit byte offset, the tile as the scalar offset.
// tools/coissue_lab: beside a wave's MFMA stream a global_load_lds piece (per-lane 64-bit addresses) costs its SIMD ~65 cycles, a
// buffer_load ... lds piece ~5.  num_records = the operand's valid bytes: rows past the end read as zeros (the kernels mask them
// anyway), so no per-lane row clamp is needed.
__device__ __forceinline__ void dma_piece(__amdgpu_buffer_rsrc_t rs, unsigned char* lds, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t*)lds, 16, voff, soff, 0, 0);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t dma_rsrc(const void* base, long bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)(bytes > 0x7FFFFFFFl ? 0x7FFFFFFFl : bytes), 0x00020000);
}


// A row of an O^T-style accumulator pair (lanes l and l + 32 hold alternating 4-value groups of row l & 31, 16 values per
// 32-column block db): v_permlane32_swap trades the odd groups of the lower lanes for the even groups of the upper ones, so
// each lane stores two whole 16-byte chunks per block instead of four 8-byte pieces.  chunk index = db * 4 + 2 gp + hi.
template <typename F>
__device__ __forceinline__ void store_row_chunks(const f32x16 (&acc)[2], float mul, int hi, bool ok, F&& put) {
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int gp = 0; gp < 2; ++gp) {
            uint32_t a[2], c[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                a[k] = pack_bf16x2(acc[db][(2 * gp) * 4 + 2 * k] * mul, acc[db][(2 * gp) * 4 + 2 * k + 1] * mul);
                c[k] = pack_bf16x2(acc[db][(2 * gp + 1) * 4 + 2 * k] * mul, acc[db][(2 * gp + 1) * 4 + 2 * k + 1] * mul);
                const auto sw = __builtin_amdgcn_permlane32_swap(a[k], c[k], false, false);
                a[k] = sw[0];
                c[k] = sw[1];
            }
            if (ok) put(db * 4 + 2 * gp + hi, make_uint4(a[0], a[1], c[0], c[1]));
        }
}

// PRESCALED: Q already carries scale * log2(e) (m324_qkv_split's q_scale), so scores are log2-domain.
// NQ: 32-row query blocks per wave (1 or 2).  With NQ = 2 every K / Vt fragment read from LDS feeds two
// MFMAs, a wave issues 32 MFMAs per barrier instead of 16, and the two blocks' softmax chains are
// independent, so the scheduler can run one block's exp2 / pack work under the other block's MFMAs.
// NWV: waves per workgroup (4 or 8).  Eight waves = 256 queries share every K / Vt tile, which halves the LDS-DMA
// pieces per FLOP (the texture path is ~90 % busy at four waves) and, for the 10 368-token global attention, turns
// 972 workgroups on 768 slots (two rounds, the second a quarter full) into 492 on 256 slots (1.92 rounds).
// VROW: the V operand is row-major V[B,H,Lk,64] (what a fused QKV projection epilogue writes) instead of the transposed,
// key-permuted Vt: the V tile is staged like the K tile and the A fragments of O^T = V^T P^T (8 keys of one d) are
// fetched with the transposing LDS read (ds_read_b64_tr_b16: a 16-lane group turns a [4 keys][16 d] block into one
// column per lane; semantics measured with tools/tr_lab), picking the keys in the order the swapped MFMA contracts them.
typedef __attribute__((ext_vector_type(4))) short a_s16x4_t;
typedef __attribute__((ext_vector_type(8))) short a_s16x8_t;
typedef __attribute__((address_space(3))) a_s16x4_t a_lds_s16x4_t;

// NST: stages of the K / V ring.  NST = 1 is the one-tile form (Lk <= 64: the decoder's 64 latent tokens under 2048 shared
// queries x 32 frames = 6144 workgroups that each live for one tile): 16 KiB of LDS and a 128-register budget, so that
// four of these latency-bound workgroups share a CU instead of two.
// NST = 2 (round 6, per-frame blocks): one tile of look-ahead instead of two, 32 KiB of LDS and a 128-register budget, so that FOUR
// of these short-lived workgroups (a few tiles each: 257 / 324 keys) share a CU instead of three -- their start-up latency (query
// fragments, first tiles) is what the launch spends its time on, and more co-resident workgroups hide more of it.
template <bool PRESCALED, int NQ, int NWV, bool VROW = false, int NST = 3>
__global__ __launch_bounds__(NWV * 64, NWV == 8 ? (VROW ? 2 : 4) : ((NST == 1 || NST == 2) ? 4 : 1)) void attn_bf16_kernel(const bf16_t* __restrict__ Q, long q_bstride,
                                                        const bf16_t* __restrict__ K, const bf16_t* __restrict__ Vt,
                                                        bf16_t* __restrict__ O, long ldo, int H, int Lq, int Lk,
                                                        int Lkp, float scale_log2e, float* __restrict__ lse, int nqt, int xflags) {
    // [stage][K | Vt]; the one-tile form appends a wave-private 4-KiB block per wave for the whole-row output stores (its only
    // stage is still being read by the slower waves when the first one is done; 32 KiB keeps four workgroups per CU)
    __shared__ __attribute__((aligned(1024))) unsigned char smem[NST * ASTAGE + (NST == 1 ? NWV * 4096 : 0)];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    int b = blockIdx.z, h = blockIdx.y, qt = blockIdx.x;
    if (nqt > 0) {
        // flat grid (long sequences): workgroup i runs on XCD i % 8; give every XCD a contiguous range of the
        // (batch, head, query tile) list, i.e. whole heads, so the ~20 workgroups that walk one head's K / V in step
        // share its tiles in ONE L2 instead of five workgroups in each of the eight
        const int nb = gridDim.x, q = nb >> 3, r = nb & 7, x = qt & 7, loc = qt >> 3;
        const int lid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + loc;
        qt = lid % nqt;
        h = (lid / nqt) % H;
        b = lid / (nqt * H);
    }
    const int q0 = (qt * NWV + wave) * (QW * NQ);

    const bf16_t* Qh = Q + (long)b * q_bstride + (long)h * Lq * 64;
    const bf16_t* Kh = K + ((long)b * H + h) * (long)Lk * 64;
    const bf16_t* Vh = Vt + ((long)b * H + h) * (VROW ? (long)Lk * 64 : 64 * (long)Lkp);

    // Q fragments (B operand): lane (q = l31, hi) holds Q[q][ks*16 + hi*8 .. +7] for ks = 0..3
    bf16x8 qf[NQ][4];
#pragma unroll
    for (int n = 0; n < NQ; ++n) {
        const int q = q0 + n * QW + l31;
        const bool ok = q < Lq;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            uint4 v = ok ? *reinterpret_cast<const uint4*>(Qh + (long)q * 64 + ks * 16 + hi * 8) : make_uint4(0, 0, 0, 0);
            qf[n][ks] = *reinterpret_cast<bf16x8*>(&v);
        }
    }

    // LDS-DMA staging.  A wave-instruction fills 8 tile rows (1 KiB); the 8 row groups of the K tile and of the Vt
    // tile are dealt to the waves (NWV = 4: groups 2w, 2w+1; NWV = 8: group w).  Lane l fills row r = 8g + (l >> 3),
    // slot l & 7, which holds chunk (l & 7) ^ ((r >> 1) & 7).  K rows past Lk read as zeros (their scores are masked);
    // Vt is zero padded.
    constexpr int GPW = 8 / NWV;                 // row groups per wave per operand
    constexpr int PPT = 2 * GPW;                 // LDS-DMA pieces per wave per tile
    unsigned vk[GPW], vv[GPW];                   // the lane's byte offsets inside a K / V tile of the (batch, head)
#pragma unroll
    for (int i = 0; i < GPW; ++i) {
        const int srow = (wave * GPW + i) * 8 + (lane >> 3);
        const int scol = ((lane & 7) ^ ((srow >> 1) & 7)) * 8;
        vk[i] = (unsigned)((srow * 64 + scol) * 2);
        // VROW: V rows are keys; slot swizzle c ^ 4 ((row >> 1) & 1) keeps the 4 rows x 64 B of a transposing read apart
        vv[i] = VROW ? (unsigned)((srow * 64 + ((lane & 7) ^ (4 * ((srow >> 1) & 1))) * 8) * 2)
                     : (unsigned)(((long)srow * Lkp + scol) * 2);
    }
    const __amdgpu_buffer_rsrc_t rk = dma_rsrc(Kh, (long)Lk * 128);
    const __amdgpu_buffer_rsrc_t rv = dma_rsrc(Vh, VROW ? (long)Lk * 128 : 64l * Lkp * 2);
    auto issue_tile = [&](int t) {
        const int kv0 = t * KV;
        unsigned char* sk = smem + (t % NST) * ASTAGE + wave * (GPW * 1024);
        unsigned char* sv = sk + 8192;
#pragma unroll
        for (int i = 0; i < GPW; ++i) dma_piece(rk, sk + i * 1024, vk[i], (unsigned)(kv0 * 128));
#pragma unroll
        for (int i = 0; i < GPW; ++i) dma_piece(rv, sv + i * 1024, vv[i], (unsigned)(VROW ? kv0 * 128 : kv0 * 2));
    };

    f32x16 o[NQ][2];
#pragma unroll
    for (int n = 0; n < NQ; ++n)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[n][i][r] = 0.f;
    // Online softmax with a LAZY reference maximum (log2 domain).  m_ref is the value the scores are
    // measured against; it is folded into the MFMA accumulator's initial value (s' = K.Q - m_ref comes
    // straight out of the matrix core) and is only moved when some score of the tile exceeds it by more
    // than THR -- so the common tile costs one exp2, one add and half a max3 / cvt per score and never
    // touches the O accumulators.  P <= 2^THR keeps everything far inside fp32 / bf16 range.
    constexpr float THR = 8.0f;
    float m_ref[NQ], l_run[NQ];     // l_run: this lane's partial row sum (its 32 of the tile's 64 keys)
#pragma unroll
    for (int n = 0; n < NQ; ++n) m_ref[n] = 0.f, l_run[n] = 0.f;
    bool first = true;

    const int nt = (Lk + KV - 1) / KV;
    issue_tile(0);
    if (NST != 2 && nt > 1) issue_tile(1);
    // Static priority for the second-dispatched half of an 8-wave workgroup (experiment switch M324_ATTN_EXP bit 0): the
    // younger wave of a SIMD loses every VALU arbitration against its older partner (microarch guide, "two waves per SIMD",
    // item 4); one s_setprio for the whole loop, no per-segment flips.
    if (NWV == 8 && (xflags & 1) && wave >= 4) __builtin_amdgcn_s_setprio(1);
    // lane part of a K / Vt fragment address: row l31 of a 32-row block, chunk hi swizzled by the row.  k-step ks toggles
    // chunk bits 1-2, i.e. XORs the byte offset with ks << 5, and the stage base (a multiple of 16 KiB) can be added
    // before that XOR: one v_add per tile + one v_xor per read replace the full swizzle arithmetic per read (35 of the
    // ~150 VALU instructions of a tile, in a kernel whose VALU time equals its MFMA time).
    const int ko0 = k_off(l31, hi);

    // Partly filled last query tile (per-frame blocks: L = 257 -> its third 128-row tile holds ONE row, L = 324 -> 68): the waves
    // whose 32 rows lie past Lq only stage tiles and meet the barriers.  The guard lives in a second copy of the loop that only
    // those workgroups run (round 2 put the branch into the one loop every workgroup runs: +4 % on the whole kernel).
    const bool idle = q0 >= Lq;
    auto tiles = [&](auto guard_tag) {
        constexpr bool GUARD = decltype(guard_tag)::value;
        for (int t = 0; t < nt; ++t) {
            // tile t landed (this wave's 4 pieces; tile t+1's may still fly), then the barrier publishes every
            // wave's pieces and retires all reads of the stage that tile t+2 is about to overwrite
            if (NST != 2 && t + 1 < nt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPT) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (NST == 2) {                      // two stages: the barrier retired every read of the stage tile t + 1 goes into
                if (t + 1 < nt) issue_tile(t + 1);
            } else if (t + 2 < nt) issue_tile(t + 2);
            if (GUARD && idle) continue;      // a wave without query rows stages its pieces and meets the barriers, nothing else
            const unsigned char* sk = smem + (t % NST) * ASTAGE;
            const unsigned char* sv = sk + 8192;
            const int kos = ko0 + (t % NST) * ASTAGE;

            // ---- S'^T = K Q^T - m_ref : two 32-key blocks per query block
            f32x16 s[NQ][2];
    #pragma unroll
            for (int n = 0; n < NQ; ++n) {
                const float init = PRESCALED ? -m_ref[n] : -m_ref[n] / scale_log2e;
    #pragma unroll
                for (int kb = 0; kb < 2; ++kb)
    #pragma unroll
                    for (int r = 0; r < 16; ++r) s[n][kb][r] = init;
            }
            if constexpr (NQ == 1 && !VROW && NWV == 8) {
                // all eight K fragments in flight before the first MFMA (the compiler's order -- two reads, wait, two MFMAs -- waits
                // out an LDS round trip four times per tile; in-kernel stamps, tools/attn_trace.py)
                bf16x8 kfa[8];          // in MFMA order: (kb, ks) = (i & 1, i >> 1)
                auto rd = [&](int i) { kfa[i] = *reinterpret_cast<const bf16x8*>(smem + (i & 1) * 4096 + (kos ^ ((i >> 1) << 5))); };
                rd(0); rd(1); rd(2); rd(3);
                __builtin_amdgcn_sched_barrier(0);
    #pragma unroll
                for (int i = 0; i < 8; ++i) {
                    s[0][i & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfa[i], qf[0][i >> 1], s[0][i & 1], 0, 0, 0);
                    if (i + 4 < 8) rd(i + 4);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (i + 4 < 8) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
            } else {
    #pragma unroll
            for (int kb = 0; kb < 2; ++kb)
    #pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    bf16x8 kf = *reinterpret_cast<const bf16x8*>(smem + kb * 4096 + (kos ^ (ks << 5)));
    #pragma unroll
                    for (int n = 0; n < NQ; ++n)
                        s[n][kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[n][ks], s[n][kb], 0, 0, 0);
                }
            }
            if (!PRESCALED) {
    #pragma unroll
                for (int n = 0; n < NQ; ++n)
    #pragma unroll
                    for (int kb = 0; kb < 2; ++kb)
    #pragma unroll
                        for (int r = 0; r < 16; ++r) s[n][kb][r] *= scale_log2e;
            }
            const int kv0 = t * KV;
            if (kv0 + KV > Lk) {   // ragged last tile: mask keys >= Lk (wave-uniform branch)
    #pragma unroll
                for (int n = 0; n < NQ; ++n)
    #pragma unroll
                    for (int kb = 0; kb < 2; ++kb)
    #pragma unroll
                        for (int r = 0; r < 16; ++r)
                            if (kv0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi >= Lk) s[n][kb][r] = -INFINITY;
            }
            // each lane checks its own half of the row against THR; the wave-wide vote combines the halves, so the
            // cross-half exchange (an LDS round trip and a wait in the middle of the tile) is only paid when the reference moves
            float mx[NQ];
            bool calm = true;
    #pragma unroll
            for (int n = 0; n < NQ; ++n) {
                float v = -INFINITY;
    #pragma unroll
                for (int kb = 0; kb < 2; ++kb)
    #pragma unroll
                    for (int r = 0; r < 16; ++r) v = fmaxf(v, s[n][kb][r]);
                mx[n] = v;
                calm = calm && (v <= THR);
            }
            // move the reference only when needed (wave-uniform decision)
            if (first || !__all(calm)) {
    #pragma unroll
                for (int n = 0; n < NQ; ++n) {
                    mx[n] = fmaxf(mx[n], __shfl_xor(mx[n], 32, 64));
                    const float shift = first ? mx[n] : fmaxf(mx[n], 0.f);     // m_ref never decreases
                    const float alpha = first ? 0.f : __builtin_amdgcn_exp2f(-shift);
                    m_ref[n] += shift;
                    l_run[n] *= alpha;
    #pragma unroll
                    for (int i = 0; i < 2; ++i)
    #pragma unroll
                        for (int r = 0; r < 16; ++r) o[n][i][r] *= alpha;
    #pragma unroll
                    for (int kb = 0; kb < 2; ++kb)
    #pragma unroll
                        for (int r = 0; r < 16; ++r) s[n][kb][r] -= shift;
                }
                first = false;
            }
            bf16x8 pf[NQ][4];   // P^T fragments, k-step j = kb*2 + (r>>3)
    #pragma unroll
            for (int n = 0; n < NQ; ++n) {
                f32x2 rs2 = {0.f, 0.f};
    #pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    uint32_t pk[8];
    #pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        f32x2 p = {__builtin_amdgcn_exp2f(s[n][kb][r]), __builtin_amdgcn_exp2f(s[n][kb][r + 1])};
                        rs2 += p;
                        pk[r >> 1] = pack_bf16x2(p[0], p[1]);
                    }
                    uint4 lo = make_uint4(pk[0], pk[1], pk[2], pk[3]), hi4 = make_uint4(pk[4], pk[5], pk[6], pk[7]);
                    pf[n][kb * 2] = *reinterpret_cast<bf16x8*>(&lo);
                    pf[n][kb * 2 + 1] = *reinterpret_cast<bf16x8*>(&hi4);
                }
                l_run[n] += rs2[0] + rs2[1];
            }

            // ---- O^T += Vt P^T.  k-step j contracts keys j*16 + {4hi..4hi+3, 8+4hi..8+4hi+3}: with the permuted
            //      key order of Vt that is the single 16-byte chunk 2j + hi of row d
    #pragma unroll
            for (int j = 0; j < 4; ++j) {
    #pragma unroll
                for (int db = 0; db < 2; ++db) {
                    bf16x8 vf;
                    if constexpr (VROW) {
                        // lane (group g = lane >> 4, i = lane & 15) supplies key row j*16 + 4 hi + 8 rd + (i >> 2), d columns
                        // db*32 + 16 (g & 1) + 4 (i & 3) .. + 3; it receives keys {4hi..4hi+3} (rd 0) and {8+4hi..} (rd 1) of d = l31
                        const int li = lane & 15, gg = lane >> 4;
                        const int d0 = db * 32 + 16 * (gg & 1) + 4 * (li & 3);
                        const int off = (j * 16 + 4 * hi + (li >> 2)) * 128 + (((d0 >> 3) ^ (4 * ((li >> 3) & 1))) << 4) + ((d0 & 7) << 1);
                        const a_s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((a_lds_s16x4_t*)(sv + off));
                        const a_s16x4_t hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((a_lds_s16x4_t*)(sv + off + 8 * 128));
                        const a_s16x8_t v8 = __builtin_shufflevector(lo, hi4, 0, 1, 2, 3, 4, 5, 6, 7);
                        vf = __builtin_bit_cast(bf16x8, v8);
                    } else {
                        vf = *reinterpret_cast<const bf16x8*>(smem + 8192 + db * 4096 + (kos ^ (j << 5)));
                    }
    #pragma unroll
                    for (int n = 0; n < NQ; ++n)
                        o[n][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[n][j], o[n][db], 0, 0, 0);
                }
            }

        }
    };
    if (NWV == 4 && NQ == 1 && NST == 3 && (qt + 1) * NWV * QW > Lq && !(xflags & 4)) tiles(std::true_type{});      // M324_ATTN_EXP bit 2: A/B
    else tiles(std::false_type{});

    // ---- normalise and store.  o[n][db][r]: d = db*32 + (r&3) + 8*(r>>2) + 4*hi, q = l31
#pragma unroll
    for (int n = 0; n < NQ; ++n) {
        const float l_tot = l_run[n] + __shfl_xor(l_run[n], 32, 64);
        const float inv = 1.0f / l_tot;
        const int q = q0 + n * QW + l31;
        if (lse && q < Lq && hi == 0) lse[((long)b * H + h) * Lq + q] = m_ref[n] + log2f(l_tot);   // log2-domain LSE
        // The two lanes of a query (l, l + 32) hold alternating 4-value groups of its row.  v_permlane32_swap trades the
        // odd groups of the lower lanes for the even groups of the upper ones, so every lane has whole 8-value (16-byte)
        // chunks; those bounce through a wave-private, XOR-swizzled 32 x 128-byte LDS block so that a store instruction
        // writes 8 whole 128-byte rows instead of 32 quarter rows (partial-line writes were what the short attentions --
        // one to six K / V tiles per workgroup -- spent their time on: decoder 48 us with 8-byte stores, 32 with 16-byte).
        // The block lives in ring stages no tile occupies any more (tiles nt and nt + 1 were never issued); the one-stage
        // form has no such stage and owns a 4-KiB block per wave behind its only stage.
        unsigned char* scr = NST == 1 ? smem + ASTAGE + wave * 4096 : smem + ((nt + (wave >> 2)) % NST) * ASTAGE + (wave & 3) * 4096;
        if (n > 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // NQ = 2: the block is reused per query block
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) {
                uint32_t a[2], c[2];             // a: group 2 gp (even), c: group 2 gp + 1 (odd)
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    a[k] = pack_bf16x2(o[n][db][(2 * gp) * 4 + 2 * k] * inv, o[n][db][(2 * gp) * 4 + 2 * k + 1] * inv);
                    c[k] = pack_bf16x2(o[n][db][(2 * gp + 1) * 4 + 2 * k] * inv, o[n][db][(2 * gp + 1) * 4 + 2 * k + 1] * inv);
                    // a.lanes[32..63] <-> c.lanes[0..31]: lower lanes now hold (own even, partner's even), upper (partner's odd, own odd)
                    const auto sw = __builtin_amdgcn_permlane32_swap(a[k], c[k], false, false);
                    a[k] = sw[0];
                    c[k] = sw[1];
                }
                const int chunk = db * 4 + 2 * gp + hi;
                if (NST == 1 && (xflags & 2)) {  // A/B (M324_ATTN_EXP bit 1): round 2's direct 16-byte stores of the one-tile form
                    if (q < Lq) *reinterpret_cast<uint4*>(O + ((long)b * Lq + q) * ldo + h * 64 + chunk * 8) = make_uint4(a[0], a[1], c[0], c[1]);
                } else {
                    *reinterpret_cast<uint4*>(scr + l31 * 128 + ((chunk ^ (l31 & 7)) << 4)) = make_uint4(a[0], a[1], c[0], c[1]);
                }
            }
        if (!(NST == 1 && (xflags & 2))) {
            const int r8 = lane >> 3, c8 = lane & 7;
            bf16_t* obase = O + ((long)b * Lq + q0 + n * QW) * ldo + h * 64 + c8 * 8;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int r = p * 8 + r8;
                const uint4 v = *reinterpret_cast<const uint4*>(scr + r * 128 + ((c8 ^ (r & 7)) << 4));
                if (q0 + n * QW + r < Lq) *reinterpret_cast<uint4*>(obase + (long)r * ldo) = v;
            }
        }
    }
}

// One key tile (Lk <= 64), ONE query set shared by every batch (q_bstride == 0), FB consecutive batches per workgroup: the
// decoder's cross-attention (the same 2048 mesh points under each of the 32 frames' 64 latent tokens, Pcd_motion.py:556-560).
// The one-tile form of attn_bf16_kernel starts 6144 workgroups that each load the SAME query fragments, wait out one LDS-DMA
// round trip, compute one tile and store 16 KiB.  Here a workgroup keeps its 128 queries' fragments in registers and walks FB
// frames: K / Vt of frame j + 1 fly (second LDS stage) while frame j is computed, and frame j - 1's output rows -- parked in the
// wave's LDS block -- are stored behind the wait at the top of iteration j, so that wait (vmcnt(0): loads and stores share the
// counter and retire out of order with respect to each other) never sees a store younger than one whole iteration.
// Measured on the decoder's shape (T = 32, H = 12, 2048 x 64; us, interleaved A/B, two boxes): one workgroup per frame 38.1 / 39.5,
// FB = 8: 34.7 / 36.1, FB = 4: 35.3 / 37.2, FB = 2: 32.8 / 34.3, FB = 2 with nontemporal stores 32.0 / 32.5 (NT: the 100 MB of
// output rows are read next by the out projection, from HBM / MALL either way; a 100-MB fill takes 16 us).
template <int FB, bool NT>
__global__ __launch_bounds__(256, 3) void attn_frames_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K,
                                                             const bf16_t* __restrict__ Vt, bf16_t* __restrict__ O, long ldo, int H, int Lq,
                                                             int Lk, int Lkp, float* __restrict__ lse) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * ASTAGE + 4 * 4096];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int h = blockIdx.y, qt = blockIdx.x, b0 = blockIdx.z * FB;
    const int q0 = (qt * 4 + wave) * QW;
    const int q = q0 + l31;
    const bf16_t* Qh = Q + (long)h * Lq * 64;
    bf16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        uint4 v = q < Lq ? *reinterpret_cast<const uint4*>(Qh + (long)q * 64 + ks * 16 + hi * 8) : make_uint4(0, 0, 0, 0);
        qf[ks] = *reinterpret_cast<bf16x8*>(&v);
    }
    unsigned vk[2], vv[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int srow = (wave * 2 + i) * 8 + (lane >> 3);
        const int scol = ((lane & 7) ^ ((srow >> 1) & 7)) * 8;
        vk[i] = (unsigned)((srow * 64 + scol) * 2);
        vv[i] = (unsigned)((srow * Lkp + scol) * 2);
    }
    auto issue = [&](int j) {                                       // K rows past Lk read as zeros (masked below), Vt is zero padded
        const long bh = (long)(b0 + j) * H + h;
        const __amdgpu_buffer_rsrc_t rk = dma_rsrc(K + bh * (long)Lk * 64, (long)Lk * 128);
        const __amdgpu_buffer_rsrc_t rv = dma_rsrc(Vt + bh * 64 * Lkp, 64l * Lkp * 2);
        unsigned char* sk = smem + (j & 1) * ASTAGE + wave * 2048;
#pragma unroll
        for (int i = 0; i < 2; ++i) dma_piece(rk, sk + i * 1024, vk[i], 0u);
#pragma unroll
        for (int i = 0; i < 2; ++i) dma_piece(rv, sk + 8192 + i * 1024, vv[i], 0u);
    };
    unsigned char* scr = smem + 2 * ASTAGE + wave * 4096;
    const int r8 = lane >> 3, c8 = lane & 7;
    auto flush = [&](int j) {                                       // frame j's 32 rows of this wave: 8 whole 128-byte rows per store
        bf16_t* obase = O + ((long)(b0 + j) * Lq + q0) * ldo + h * 64 + c8 * 8;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int r = p * 8 + r8;
            const uint4 v = *reinterpret_cast<const uint4*>(scr + r * 128 + ((c8 ^ (r & 7)) << 4));
            if (q0 + r < Lq) {
                if (NT) __builtin_nontemporal_store(*reinterpret_cast<const f32x4*>(&v), reinterpret_cast<f32x4*>(obase + (long)r * ldo));
                else *reinterpret_cast<uint4*>(obase + (long)r * ldo) = v;
            }
        }
    };
    const int ko0 = k_off(l31, hi);
    issue(0);
#pragma unroll 1
    for (int j = 0; j < FB; ++j) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // frame j's pieces (issued an iteration ago); stores of frame j - 2
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (j + 1 < FB) issue(j + 1);
        if (j > 0) flush(j - 1);
        const int kos = ko0 + (j & 1) * ASTAGE;
        f32x16 s[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(smem + kb * 4096 + (kos ^ (ks << 5)));
                s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s[kb], 0, 0, 0);
            }
        }
        if (Lk < KV) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi >= Lk) s[kb][r] = -INFINITY;
        }
        float mx = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kb][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));                    // the row's maximum (one tile: the plain softmax)
        f32x2 rs2 = {0.f, 0.f};
        bf16x8 pf[4];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            uint32_t pk[8];
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                f32x2 p = {__builtin_amdgcn_exp2f(s[kb][r] - mx), __builtin_amdgcn_exp2f(s[kb][r + 1] - mx)};
                rs2 += p;
                pk[r >> 1] = pack_bf16x2(p[0], p[1]);
            }
            uint4 lo = make_uint4(pk[0], pk[1], pk[2], pk[3]), hi4 = make_uint4(pk[4], pk[5], pk[6], pk[7]);
            pf[kb * 2] = *reinterpret_cast<bf16x8*>(&lo);
            pf[kb * 2 + 1] = *reinterpret_cast<bf16x8*>(&hi4);
        }
        f32x16 o[2];
#pragma unroll
        for (int db = 0; db < 2; ++db) {
#pragma unroll
            for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const bf16x8 vf = *reinterpret_cast<const bf16x8*>(smem + 8192 + db * 4096 + (kos ^ (jj << 5)));
                o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[jj], o[db], 0, 0, 0);
            }
        }
        float l_tot = rs2[0] + rs2[1];
        l_tot += __shfl_xor(l_tot, 32, 64);
        const float inv = 1.0f / l_tot;
        if (lse && q < Lq && hi == 0) lse[((long)(b0 + j) * H + h) * Lq + q] = mx + log2f(l_tot);
        store_row_chunks(o, inv, hi, true, [&](int chunk, uint4 v) {
            *reinterpret_cast<uint4*>(scr + l31 * 128 + ((chunk ^ (l31 & 7)) << 4)) = v;
        });
    }
    flush(FB - 1);
}

// Two other forms of the eight-wave kernel were built and measured in round 3 and are not kept (DESIGN.md section 6,
// "what limits the global attention"; tools/coissue_lab, tools/attn_trace.py):
//  * ping-pong: matrix phase {P.V of tile t, S of tile t + 1} / vector phase {softmax}, waves 4-7 one phase behind waves 0-3,
//    four ring stages, two barriers per tile: 365-380 us against 330-345.  On gfx950 a plain VALU instruction of one wave does
//    not issue while another wave of the SIMD has an MFMA pending (coissue_lab: v_fma_f32 next to an MFMA stream 34 cycles
//    per instruction instead of 2.3), so the two phases serialise instead of overlapping.
//  * half tiles: a lazy-reference check per 32 keys, the exponentials of one half between the MFMAs of the next step (only
//    v_exp_f32 runs in an MFMA's shadow: two per MFMA are free), fragments read one block ahead: 339-383 us against 319-358.
//    The compiler's schedule of the kernel above already puts two thirds of a tile's exponentials between its P.V MFMAs; the
//    second check, the extra branches and the shorter MFMA runs cost more than the remaining third buys.

// =====================================================================================================
// Backward (bf16 MFMA).  Same swapped formulation and LDS tile images as the forward:
//   dQ kernel : a wave owns 32 queries (lane = query), walks the keys:
//       S^T = K Qs^T, dP^T = V dO^T          (A = K / V rows from LDS, B = Qs / dO rows in registers)
//       dS^T = exp2(S^T - lse) * (dP^T - D)   (per-lane lse, D)
//       dQ^T += Kt dS^T                        (A = Kt rows [d][kv] in the permuted key order, B = dS^T registers)
//   dK/dV kernel : a wave owns 32 keys (lane = key), walks the queries:
//       S = Qs K^T, dP = dO V^T               (A = Qs / dO rows from LDS, B = K / V rows in registers)
//       P = exp2(S - lse[q]), dS = P * (dP - D[q])      (lse, D vary along the registers)
//       dV^T += dOt P, dK^T += Qst dS          (A = dOt / Qst rows [d][q], permuted query order)
// Every operand is staged by LDS-DMA into XOR-swizzled [64][128 B] tiles; two stages, one barrier per tile.
// Scale conventions as m324_attention_bwd (include/m324.h): dQ = scale * dS K, dK = ln2 * dS^T Qs.
// one 64-row x 128-byte tile by LDS-DMA: the 8 row groups are dealt to the NWV waves (2 pieces per wave at 4 waves, 1 at 8)
template <int NWV>
__device__ __forceinline__ void dma_rows(unsigned char* part, __amdgpu_buffer_rsrc_t rs, long row_stride, int row0, long col0, int wave,
                                         int lane) {
    constexpr int G = 8 / NWV;
    const unsigned soff = (unsigned)(((long)row0 * row_stride + col0) * 2);
#pragma unroll
    for (int i = 0; i < G; ++i) {
        const int r = (wave * G + i) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t*)(part + (wave * G + i) * 8 * 128), 16,
                                                 (unsigned)(((long)r * row_stride + c * 8) * 2), soff, 0, 0);
    }
}

__device__ __forceinline__ void pack_frags(const f32x16 (&x)[2], bf16x8 (&f)[4]) {
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        uint32_t pk[8];
#pragma unroll
        for (int r = 0; r < 16; r += 2) pk[r >> 1] = pack_bf16x2(x[kb][r], x[kb][r + 1]);
        uint4 lo = make_uint4(pk[0], pk[1], pk[2], pk[3]), hi4 = make_uint4(pk[4], pk[5], pk[6], pk[7]);
        f[kb * 2] = *reinterpret_cast<bf16x8*>(&lo);
        f[kb * 2 + 1] = *reinterpret_cast<bf16x8*>(&hi4);
    }
}

template <int NWV>
__global__ __launch_bounds__(NWV * 64) void attn_bwd_dq_mfma_kernel(const bf16_t* __restrict__ Qs, long q_bstride,
                                                               const bf16_t* __restrict__ K, const bf16_t* __restrict__ V,
                                                               const bf16_t* __restrict__ Kt, const bf16_t* __restrict__ dO,
                                                               const float* __restrict__ lse, const float* __restrict__ D,
                                                               bf16_t* __restrict__ dQ, int H, int Lq, int Lk, int Lkp, float scale) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * 3 * 8192];   // [stage][K | V | Kt]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int b = blockIdx.z, h = blockIdx.y;
    const int q = (blockIdx.x * NWV + wave) * QW + l31;
    const long bh = (long)b * H + h;
    const bool qok = q < Lq;
    const bf16_t* Qh = Qs + (long)b * q_bstride + (long)h * Lq * 64;
    const bf16_t* dOh = dO + bh * (long)Lq * 64;
    const bf16_t* Kh = K + bh * (long)Lk * 64;
    const bf16_t* Vh = V + bh * (long)Lk * 64;
    const bf16_t* Kth = Kt + bh * 64 * (long)Lkp;
    bf16x8 qf[4], dof[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        uint4 a = qok ? *reinterpret_cast<const uint4*>(Qh + (long)q * 64 + ks * 16 + hi * 8) : make_uint4(0, 0, 0, 0);
        uint4 c = qok ? *reinterpret_cast<const uint4*>(dOh + (long)q * 64 + ks * 16 + hi * 8) : make_uint4(0, 0, 0, 0);
        qf[ks] = *reinterpret_cast<bf16x8*>(&a);
        dof[ks] = *reinterpret_cast<bf16x8*>(&c);
    }
    const float l2 = qok ? lse[bh * Lq + q] : 0.f, dl = qok ? D[bh * Lq + q] : 0.f;
    const __amdgpu_buffer_rsrc_t rK = dma_rsrc(Kh, (long)Lk * 128), rV = dma_rsrc(Vh, (long)Lk * 128), rKt = dma_rsrc(Kth, 64l * Lkp * 2);
    auto issue = [&](int t) {                                   // rows past Lk read as zeros (masked below)
        unsigned char* st = smem + (t & 1) * 24576;
        dma_rows<NWV>(st, rK, 64, t * KV, 0, wave, lane);
        dma_rows<NWV>(st + 8192, rV, 64, t * KV, 0, wave, lane);
        dma_rows<NWV>(st + 16384, rKt, Lkp, 0, (long)t * KV, wave, lane);
    };
    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const int nt = (Lk + KV - 1) / KV;
    issue(0);
    for (int t = 0; t < nt; ++t) {
        // the LDS-DMA of tile t must have landed: stated explicitly -- __syncthreads() alone is compiled to
        // `s_waitcnt lgkmcnt(0); s_barrier` here (no vmcnt), which let a workgroup read a stage that was still in flight
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t + 1 < nt) issue(t + 1);
        const unsigned char* sk = smem + (t & 1) * 24576;
        const unsigned char* sv = sk + 8192;
        const unsigned char* skt = sk + 16384;
        // eight waves (one workgroup per CU): all fragments of a phase in flight before its first MFMA, as in the dK / dV kernel
        constexpr bool PRE = NWV == 8;
        f32x16 s[2], dp[2];
        bf16x8 fr[16], ft[8];
        if constexpr (PRE) {
#pragma unroll
            for (int i = 0; i < 16; ++i)
                fr[i] = *reinterpret_cast<const bf16x8*>((i & 1 ? sv : sk) + k_off((i >> 3) * 32 + l31, ((i >> 1) & 3) * 2 + hi));
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kb][r] = -l2, dp[kb][r] = -dl;      // S - lse and dP - D straight from the MFMA
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                bf16x8 kf, vf;
                if constexpr (PRE) {
                    kf = fr[kb * 8 + ks * 2], vf = fr[kb * 8 + ks * 2 + 1];
                } else {
                    kf = *reinterpret_cast<const bf16x8*>(sk + k_off(kb * 32 + l31, ks * 2 + hi));
                    vf = *reinterpret_cast<const bf16x8*>(sv + k_off(kb * 32 + l31, ks * 2 + hi));
                }
                s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s[kb], 0, 0, 0);
                dp[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, dof[ks], dp[kb], 0, 0, 0);
            }
        }
        if constexpr (PRE) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 8; ++i) ft[i] = *reinterpret_cast<const bf16x8*>(skt + k_off((i & 1) * 32 + l31, 2 * (i >> 1) + hi));
            __builtin_amdgcn_sched_barrier(0);
        }
        const int kv0 = t * KV;
        const bool ragged = kv0 + KV > Lk;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float p = __builtin_amdgcn_exp2f(s[kb][r]);
                if (ragged && kv0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi >= Lk) p = 0.f;
                s[kb][r] = p * dp[kb][r];                                         // dS^T
            }
        bf16x8 dsf[4];
        pack_frags(s, dsf);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                bf16x8 ktf;
                if constexpr (PRE) ktf = ft[j * 2 + db];
                else ktf = *reinterpret_cast<const bf16x8*>(skt + k_off(db * 32 + l31, 2 * j + hi));
                acc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ktf, dsf[j], acc[db], 0, 0, 0);
            }
    }
    {
        bf16_t* orow = dQ + (bh * Lq + (qok ? q : 0)) * 64;
        store_row_chunks(acc, scale, hi, qok, [&](int chunk, uint4 v) { *reinterpret_cast<uint4*>(orow + chunk * 8) = v; });
    }
}

// Round 6 (A/B, M324_ATTN_BWD_NW=2): the dQ kernel with 64 queries per wave -- four waves, one per SIMD, 256 queries per workgroup like the
// eight-wave form.  The eight-wave kernel reads one 16-byte fragment per lane and MFMA (24 KiB of K / V / Kt per wave and tile for 24 MFMAs):
// 128 B per clock and CU, the LDS's whole bandwidth, against MFMAs that would take half that time.  Here every K / V / Kt fragment feeds
// the MFMAs of TWO query blocks, so the reads per MFMA halve; ~330 registers (one wave per SIMD).  Compiler-scheduled.
__global__ __launch_bounds__(256) void attn_bwd_dq2_mfma_kernel(const bf16_t* __restrict__ Qs, long q_bstride, const bf16_t* __restrict__ K,
                                                                const bf16_t* __restrict__ V, const bf16_t* __restrict__ Kt,
                                                                const bf16_t* __restrict__ dO, const float* __restrict__ lse,
                                                                const float* __restrict__ D, bf16_t* __restrict__ dQ, int H, int Lq, int Lk,
                                                                int Lkp, float scale) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * 3 * 8192];   // [stage][K | V | Kt]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int b = blockIdx.z, h = blockIdx.y;
    const long bh = (long)b * H + h;
    const bf16_t* Qh = Qs + (long)b * q_bstride + (long)h * Lq * 64;
    const bf16_t* dOh = dO + bh * (long)Lq * 64;
    const bf16_t* Kh = K + bh * (long)Lk * 64;
    const bf16_t* Vh = V + bh * (long)Lk * 64;
    const bf16_t* Kth = Kt + bh * 64 * (long)Lkp;
    int q[2];
    bool qok[2];
    bf16x8 qf[2][4], dof[2][4];
    float l2[2], dl[2];
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        q[n] = (blockIdx.x * 4 + wave) * 64 + n * 32 + l31;
        qok[n] = q[n] < Lq;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            uint4 a = qok[n] ? *reinterpret_cast<const uint4*>(Qh + (long)q[n] * 64 + ks * 16 + hi * 8) : make_uint4(0, 0, 0, 0);
            uint4 c = qok[n] ? *reinterpret_cast<const uint4*>(dOh + (long)q[n] * 64 + ks * 16 + hi * 8) : make_uint4(0, 0, 0, 0);
            qf[n][ks] = *reinterpret_cast<bf16x8*>(&a);
            dof[n][ks] = *reinterpret_cast<bf16x8*>(&c);
        }
        l2[n] = qok[n] ? lse[bh * Lq + q[n]] : 0.f;
        dl[n] = qok[n] ? D[bh * Lq + q[n]] : 0.f;
    }
    const __amdgpu_buffer_rsrc_t rK = dma_rsrc(Kh, (long)Lk * 128), rV = dma_rsrc(Vh, (long)Lk * 128), rKt = dma_rsrc(Kth, 64l * Lkp * 2);
    auto issue = [&](int t) {                                   // rows past Lk read as zeros (masked below)
        unsigned char* st = smem + (t & 1) * 24576;
        dma_rows<4>(st, rK, 64, t * KV, 0, wave, lane);
        dma_rows<4>(st + 8192, rV, 64, t * KV, 0, wave, lane);
        dma_rows<4>(st + 16384, rKt, Lkp, 0, (long)t * KV, wave, lane);
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[n][i][r] = 0.f;
    const int nt = (Lk + KV - 1) / KV;
    issue(0);
    for (int t = 0; t < nt; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // tile t landed (see the eight-wave kernel)
        __syncthreads();
        if (t + 1 < nt) issue(t + 1);
        const unsigned char* sk = smem + (t & 1) * 24576;
        const unsigned char* sv = sk + 8192;
        const unsigned char* skt = sk + 16384;
        f32x16 s[2][2], dp[2][2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) s[n][kb][r] = -l2[n], dp[n][kb][r] = -dl[n];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(sk + k_off(kb * 32 + l31, ks * 2 + hi));
                const bf16x8 vf = *reinterpret_cast<const bf16x8*>(sv + k_off(kb * 32 + l31, ks * 2 + hi));
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    s[n][kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[n][ks], s[n][kb], 0, 0, 0);
                    dp[n][kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, dof[n][ks], dp[n][kb], 0, 0, 0);
                }
            }
        }
        const int kv0 = t * KV;
        const bool ragged = kv0 + KV > Lk;
        bf16x8 dsf[2][4];
#pragma unroll
        for (int n = 0; n < 2; ++n) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float p = __builtin_amdgcn_exp2f(s[n][kb][r]);
                    if (ragged && kv0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi >= Lk) p = 0.f;
                    s[n][kb][r] = p * dp[n][kb][r];                                   // dS^T
                }
            pack_frags(s[n], dsf[n]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                const bf16x8 ktf = *reinterpret_cast<const bf16x8*>(skt + k_off(db * 32 + l31, 2 * j + hi));
#pragma unroll
                for (int n = 0; n < 2; ++n) acc[n][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ktf, dsf[n][j], acc[n][db], 0, 0, 0);
            }
    }
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        bf16_t* orow = dQ + (bh * Lq + (qok[n] ? q[n] : 0)) * 64;
        store_row_chunks(acc[n], scale, hi, qok[n], [&](int chunk, uint4 v) { *reinterpret_cast<uint4*>(orow + chunk * 8) = v; });
    }
}

template <int NWV>
__global__ __launch_bounds__(NWV * 64, 2) void attn_bwd_dkv_mfma_kernel(const bf16_t* __restrict__ Qs, const bf16_t* __restrict__ Qst,
                                                                long q_bstride, long qt_bstride, const bf16_t* __restrict__ K,
                                                                const bf16_t* __restrict__ V, const bf16_t* __restrict__ dO,
                                                                const bf16_t* __restrict__ dOt, const float* __restrict__ lse,
                                                                const float* __restrict__ D, bf16_t* __restrict__ dK,
                                                                bf16_t* __restrict__ dV, int H, int Lq, int Lk, int Lqp) {
    // [stage][Qs | dO | Qst | dOt], then [stage][lse of the tile's 64 queries | D]: the two row vectors ride with the tile
    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * 4 * 8192 + 2 * 512];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int b = blockIdx.z, h = blockIdx.y;
    const int kv = (blockIdx.x * NWV + wave) * QW + l31;
    const long bh = (long)b * H + h;
    const bool kok = kv < Lk;
    const bf16_t* Qh = Qs + (long)b * q_bstride + (long)h * Lq * 64;
    const bf16_t* Qth = Qst + (long)b * qt_bstride + (long)h * 64 * Lqp;
    const bf16_t* dOh = dO + bh * (long)Lq * 64;
    const bf16_t* dOth = dOt + bh * 64 * (long)Lqp;
    const bf16_t* Kh = K + bh * (long)Lk * 64;
    const bf16_t* Vh = V + bh * (long)Lk * 64;
    const float* lseh = lse + bh * Lq;
    const float* Dh = D + bh * Lq;
    bf16x8 kf[4], vf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        uint4 a = kok ? *reinterpret_cast<const uint4*>(Kh + (long)kv * 64 + ks * 16 + hi * 8) : make_uint4(0, 0, 0, 0);
        uint4 c = kok ? *reinterpret_cast<const uint4*>(Vh + (long)kv * 64 + ks * 16 + hi * 8) : make_uint4(0, 0, 0, 0);
        kf[ks] = *reinterpret_cast<bf16x8*>(&a);
        vf[ks] = *reinterpret_cast<bf16x8*>(&c);
    }
    const __amdgpu_buffer_rsrc_t rQ = dma_rsrc(Qh, (long)Lq * 128), rdO = dma_rsrc(dOh, (long)Lq * 128), rQt = dma_rsrc(Qth, 64l * Lqp * 2),
                                 rdOt = dma_rsrc(dOth, 64l * Lqp * 2), rL = dma_rsrc(lseh, (long)Lq * 4), rD = dma_rsrc(Dh, (long)Lq * 4);
    auto issue = [&](int t) {
        unsigned char* st = smem + (t & 1) * 32768;
        dma_rows<NWV>(st, rQ, 64, t * KV, 0, wave, lane);
        dma_rows<NWV>(st + 8192, rdO, 64, t * KV, 0, wave, lane);
        dma_rows<NWV>(st + 16384, rQt, Lqp, 0, (long)t * KV, wave, lane);
        dma_rows<NWV>(st + 24576, rdOt, Lqp, 0, (long)t * KV, wave, lane);
        // lse / D of the tile's queries: one 4-byte-per-lane LDS-DMA each (256 B), by waves 0 and 1; queries past the end read
        // as zeros here and are masked where they are used
        if (wave < 2)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wave == 0 ? rL : rD, (lds_ptr_t*)(smem + 65536 + (t & 1) * 512 + wave * 256), 4,
                                                     (unsigned)(lane * 4), (unsigned)(t * KV * 4), 0, 0);
    };
    f32x16 ak[2], av[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) ak[i][r] = 0.f, av[i][r] = 0.f;
    const int nt = (Lq + KV - 1) / KV;
    issue(0);
    for (int t = 0; t < nt; ++t) {
        // the LDS-DMA of tile t must have landed: stated explicitly -- __syncthreads() alone is compiled to
        // `s_waitcnt lgkmcnt(0); s_barrier` here (no vmcnt), which let a workgroup read a stage that was still in flight
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t + 1 < nt) issue(t + 1);
        const unsigned char* sq = smem + (t & 1) * 32768;
        const unsigned char* sdo = sq + 8192;
        const unsigned char* sqt = sq + 16384;
        const unsigned char* sdot = sq + 24576;
        const int q0 = t * KV;
        // The rows of S / dP (queries) run along the registers, so lse[q] and D[q] are 2 x 16 values per lane and tile.  Round 2
        // loaded them from global memory as the accumulators' initial values: the first MFMA of every tile waited out 16 L2
        // round trips (in-kernel stamps, tools/attn_bwd_lab.py --trace: 6600 of a tile's 9500 cycles).  Now they arrive with
        // the tile by LDS-DMA, S and dP start from the inline constant 0, and the two vectors are read (broadcast reads, 16
        // bytes per lane) behind the 16 MFMAs -- a subtraction per value instead of a negated move, no register held across.
        // Eight waves (one workgroup per CU, 256 registers per wave): all 16 fragments of a phase are in flight before its
        // first MFMA, and the second phase's fragments are requested before the exponentials.  Read where the compiler puts
        // them -- in front of each MFMA pair -- the 16 MFMAs of the dV / dK phase took 2200 cycles (stamps).
        constexpr bool PRE = NWV == 8;
        f32x16 s[2], dp[2];
        bf16x8 fr[16], ft[16];
        if constexpr (PRE) {
#pragma unroll
            for (int i = 0; i < 16; ++i)
                fr[i] = *reinterpret_cast<const bf16x8*>((i & 1 ? sdo : sq) + k_off((i >> 3) * 32 + l31, ((i >> 1) & 3) * 2 + hi));
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[qb][r] = 0.f, dp[qb][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                bf16x8 qfr, dfr;
                if constexpr (PRE) {
                    qfr = fr[qb * 8 + ks * 2], dfr = fr[qb * 8 + ks * 2 + 1];
                } else {
                    qfr = *reinterpret_cast<const bf16x8*>(sq + k_off(qb * 32 + l31, ks * 2 + hi));
                    dfr = *reinterpret_cast<const bf16x8*>(sdo + k_off(qb * 32 + l31, ks * 2 + hi));
                }
                s[qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qfr, kf[ks], s[qb], 0, 0, 0);
                dp[qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dfr, vf[ks], dp[qb], 0, 0, 0);
            }
        }
        if constexpr (PRE) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 16; ++i)          // order of use: (j, db, dOt | Qst)
                ft[i] = *reinterpret_cast<const bf16x8*>((i & 1 ? sqt : sdot) + k_off(((i >> 1) & 1) * 32 + l31, 2 * (i >> 2) + hi));
            __builtin_amdgcn_sched_barrier(0);
        }
        const float* rowv = reinterpret_cast<const float*>(smem + 65536 + (t & 1) * 512);
        const bool ragged = q0 + KV > Lq;
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int qi = qb * 32 + 8 * g + 4 * hi;             // accumulator rows (r & 3) + 8 (r >> 2) + 4 hi
                const float4 l4 = *reinterpret_cast<const float4*>(rowv + qi);
                const float4 d4 = *reinterpret_cast<const float4*>(rowv + 64 + qi);
                const float lq[4] = {l4.x, l4.y, l4.z, l4.w}, dq[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float p = __builtin_amdgcn_exp2f(s[qb][g * 4 + e] - lq[e]);
                    if (ragged && q0 + qi + e >= Lq) p = 0.f;       // query past the end (wave-uniform outer condition)
                    s[qb][g * 4 + e] = p;
                    dp[qb][g * 4 + e] = p * (dp[qb][g * 4 + e] - dq[e]);
                }
            }
        bf16x8 pf[4], dsf[4];
        pack_frags(s, pf);
        pack_frags(dp, dsf);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                bf16x8 dotf, qtf;
                if constexpr (PRE) {
                    dotf = ft[j * 4 + db * 2], qtf = ft[j * 4 + db * 2 + 1];
                } else {
                    dotf = *reinterpret_cast<const bf16x8*>(sdot + k_off(db * 32 + l31, 2 * j + hi));
                    qtf = *reinterpret_cast<const bf16x8*>(sqt + k_off(db * 32 + l31, 2 * j + hi));
                }
                av[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dotf, pf[j], av[db], 0, 0, 0);
                ak[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qtf, dsf[j], ak[db], 0, 0, 0);
            }
    }
    {
        bf16_t* krow = dK + (bh * Lk + (kok ? kv : 0)) * 64;
        bf16_t* vrow = dV + (bh * Lk + (kok ? kv : 0)) * 64;
        const float ln2 = 0.69314718055994530942f;
        store_row_chunks(ak, ln2, hi, kok, [&](int chunk, uint4 v) { *reinterpret_cast<uint4*>(krow + chunk * 8) = v; });
        store_row_chunks(av, 1.0f, hi, kok, [&](int chunk, uint4 v) { *reinterpret_cast<uint4*>(vrow + chunk * 8) = v; });
    }
}

// ------------------------------------------------------------------------------------------- fp32
constexpr int FLD = 65;   // padded row length (floats) of the fp32 tiles: conflict-free column reads

__global__ __launch_bounds__(256) void attn_f32_kernel(const float* __restrict__ Q, long q_bstride,
                                                       const float* __restrict__ K, const float* __restrict__ Vt,
                                                       float* __restrict__ O, long ldo, int H, int Lq, int Lk, int Lkp,
                                                       float scale_log2e, float* __restrict__ lse) {
    __shared__ float sk[KV * FLD];   // [kv][d]
    __shared__ float sv[64 * FLD];   // [d][kv]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int b = blockIdx.z, h = blockIdx.y;
    const int q0 = blockIdx.x * QB + wave * QW;

    const float* Qh = Q + (long)b * q_bstride + (long)h * Lq * 64;
    const float* Kh = K + ((long)b * H + h) * (long)Lk * 64;
    const float* Vh = Vt + ((long)b * H + h) * 64 * (long)Lkp;

    // lane (q, hi) holds Q[q][2s + hi], s = 0..31
    float qf[32];
    {
        const int q = q0 + l31;
        const bool ok = q < Lq;
#pragma unroll
        for (int s = 0; s < 32; ++s) qf[s] = ok ? Qh[(long)q * 64 + 2 * s + hi] : 0.f;
    }
    f32x16 o[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    const int nt = (Lk + KV - 1) / KV;
    for (int t = 0; t < nt; ++t) {
        const int kv0 = t * KV;
        __syncthreads();   // previous tile fully consumed
#pragma unroll
        for (int i = 0; i < 4; ++i) {   // 64 x 64 floats = 1024 float4 per tile, 4 per thread
            const int id = tid + 256 * i, row = id >> 4, c = id & 15;
            float4 kx = (kv0 + row < Lk) ? *reinterpret_cast<const float4*>(Kh + (long)(kv0 + row) * 64 + c * 4)
                                         : make_float4(0.f, 0.f, 0.f, 0.f);
            float4 vx = *reinterpret_cast<const float4*>(Vh + (long)row * Lkp + kv0 + c * 4);
            float* pk = sk + row * FLD + c * 4;
            // Vt is stored with the key quarters of each 16-key group in the order 0,2,1,3: restore key order
            const int qd = c & 3, lq = qd == 1 ? 2 : (qd == 2 ? 1 : qd);
            float* pv = sv + row * FLD + (c >> 2) * 16 + lq * 4;
            pk[0] = kx.x; pk[1] = kx.y; pk[2] = kx.z; pk[3] = kx.w;
            pv[0] = vx.x; pv[1] = vx.y; pv[2] = vx.z; pv[3] = vx.w;
        }
        __syncthreads();

        f32x16 s[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
#pragma unroll
            for (int st = 0; st < 32; ++st)
                s[kb] = __builtin_amdgcn_mfma_f32_32x32x2f32(sk[(kb * 32 + l31) * FLD + 2 * st + hi], qf[st], s[kb], 0, 0, 0);
        }
        if (kv0 + KV > Lk) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (kv0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi >= Lk) s[kb][r] = -INFINITY;
        }
        float mx = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kb][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx * scale_log2e);
        const float alpha = exp2f(m_run - m_new);
        m_run = m_new;
        float rs = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                s[kb][r] = exp2f(fmaf(s[kb][r], scale_log2e, -m_new));
                rs += s[kb][r];
            }
        l_run = l_run * alpha + rs;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
        // O^T += Vt P^T : MFMA step (kb, r) contracts key kv = kb*32 + (r&3) + 8*(r>>2) + 4*hi
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kv = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
#pragma unroll
                for (int db = 0; db < 2; ++db)
                    o[db] = __builtin_amdgcn_mfma_f32_32x32x2f32(sv[(db * 32 + l31) * FLD + kv], s[kb][r], o[db], 0, 0, 0);
            }
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    const int q = q0 + l31;
    if (lse && q < Lq && hi == 0) lse[((long)b * H + h) * Lq + q] = m_run + log2f(l_tot);
    if (q < Lq) {
        float* orow = O + ((long)b * Lq + q) * ldo + h * 64;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float4 w = make_float4(o[db][g * 4 + 0] * inv, o[db][g * 4 + 1] * inv, o[db][g * 4 + 2] * inv,
                                       o[db][g * 4 + 3] * inv);
                *reinterpret_cast<float4*>(orow + db * 32 + g * 8 + hi * 4) = w;
            }
    }
}

}  // namespace

// attention_pwg.hip: one wave per SIMD, hand-placed instruction stream (long sequences, pre-scaled Q, transposed zero-padded Vt)
void m324_attn_pwg_launch(const void* Q, long q_bstride, const void* K, const void* Vt, void* O, long ldo, int B, int H, int Lq, int Lk,
                          float* lse, bool bounded, hipStream_t s);
static bool use_pwg(bool prescaled, bool vrow, bool nq2, int fnw, int Lq, int Lk) {
    return m324::tunable(m324::TUN_ATTN_PWG) != 0 && prescaled && !vrow && !nq2 && fnw == 0 && Lq >= 2048 && Lk >= 512;
}

// the per-frame attentions (row-major V from the fused q|k|v epilogue, pre-scaled q, four waves, a handful of key tiles): the
// two-stage, four-per-CU instantiation
static bool two_stage(bool vrow, bool prescaled, bool w8, int Lk) {
    (void)vrow;                                   // both V layouts (the training step's per-frame blocks read the transposed Vt)
    return prescaled && !w8 && Lk > KV && Lk <= 16 * KV && m324::tunable(m324::TUN_ATTN_OCC) != 3;
}

extern "C" int m324_attention(const void* Q, long q_bstride, const void* K, const void* Vt, void* O, long ldo, int B,
                              int H, int Lq, int Lk, float scale, int q_prescaled, float* lse, int dtype, void* stream) {
    M324_REQUIRE(Q && K && Vt && O, "m324_attention: null pointer");
    M324_REQUIRE(B > 0 && H > 0 && Lq > 0 && Lk > 0, "m324_attention: empty problem B=%d H=%d Lq=%d Lk=%d", B, H, Lq, Lk);
    M324_REQUIRE(ldo >= (long)H * 64, "m324_attention: ldo too small");
    M324_REQUIRE(H <= 65535 && B <= 65535, "m324_attention: grid too large");
    const int Lkp = (Lk + 63) / 64 * 64;
    dim3 grid(ceil_div(Lq, QB), H, B);
    hipStream_t s = (hipStream_t)stream;
    const bool vrow = (q_prescaled & M324_ATTN_V_ROWMAJOR) != 0;
    const bool bounded = (q_prescaled & M324_ATTN_SCORES_BOUNDED) != 0;
    q_prescaled &= M324_ATTN_Q_PRESCALED;
    M324_REQUIRE(!vrow || dtype == M324_BF16, "m324_attention: row-major V needs the bf16 kernel (transposing LDS reads)");
    const float sl = q_prescaled ? 1.0f : scale * LOG2E;
    if (dtype == M324_BF16) {
        M324_REQUIRE((ldo * 2) % 8 == 0, "m324_attention: ldo misaligned");
        // NQ = 2 (two query blocks per wave) measured slower than NQ = 1 on MI355X (254 VGPRs -> one wave per
        // SIMD); it stays selectable for experiments only.
        const bool nq2 = m324::tunable(m324::TUN_ATTN_NQ2) != 0 && Lq >= 1024 && !vrow;
        // eight waves per workgroup for long query sets (M324_ATTN_NW=4|8 forces: A/B runs, tests)
        const int fnw = m324::tunable(m324::TUN_ATTN_NW);
        const bool w8 = !nq2 && (fnw ? fnw == 8 : (Lq >= 2048 && Lk >= 512));
        if (use_pwg(q_prescaled != 0, vrow, nq2, fnw, Lq, Lk)) {        // M324_ATTN_PWG=0: the eight-wave kernel below (A/B runs, tests)
            M324_REQUIRE((long)ceil_div(Lq, 256) * H * B < (1l << 31), "m324_attention: grid too large");
            m324_attn_pwg_launch(Q, q_bstride, K, Vt, O, ldo, B, H, Lq, Lk, lse, bounded, s);
            M324_CHECK_LAUNCH("m324_attention");
            return M324_OK;
        }
        dim3 g2(ceil_div(Lq, nq2 ? 2 * QB : (w8 ? 2 * QB : QB)), H, B);
        // XCD-aware flat grid for the 8-wave kernel (M324_ATTN_FLAT=0 keeps the 3-D grid: A/B runs)
        // The same flat order for the short sequences with several query tiles per (batch, head) -- the per-frame blocks:
        // 324 / 257 tokens = 3 tiles of 128 queries that walk the SAME K / V.  On the 3-D grid they are consecutive
        // workgroup ids, i.e. they land on three different XCDs and each pulls the head's K / V through its own L2
        // (round 2 counters: 111.8 MB moved for 55.8 MB algorithmic); on the flat grid they are neighbours in ONE XCD's
        // list.  M324_ATTN_FLAT=2 restricts the flat grid to the 8-wave kernel again (A/B runs).
        int nqt = 0;
        const int flat = m324::tunable(m324::TUN_ATTN_FLAT);
        const bool one_tile = q_prescaled && !w8 && !nq2 && !vrow && Lk <= KV;
        if (flat != 0 && (w8 || (flat != 2 && !nq2 && !one_tile && g2.x > 1 && (long)g2.x * H * B >= 512))) {
            nqt = (int)g2.x;
            g2 = dim3(g2.x * H * B, 1, 1);
        }
        // Co-residency: the NQ = 1 kernel fits 3 workgroups per CU (168 VGPRs, 32 KiB LDS).  Interleaved A/B on
        // MI355X: 3 per CU beats 2 per CU (422 vs 453 us on the 10 368-token global attention) even though the
        // grid then ends in a partly filled round -- latency hiding wins over round quantisation.
        // M324_ATTN_OCC=2 pads the LDS allocation to force two per CU (experiments only).
        const unsigned pad = m324::tunable(m324::TUN_ATTN_OCC) == 2 ? 24 * 1024 : 0;
        const int xfl = m324::tunable(m324::TUN_ATTN_EXP);
#define M324_ATTN(PS, NQ, NWV)                                                                                          \
    hipLaunchKernelGGL((attn_bf16_kernel<PS, NQ, NWV>), g2, dim3(NWV * 64), pad, s, (const bf16_t*)Q, q_bstride,         \
                       (const bf16_t*)K, (const bf16_t*)Vt, (bf16_t*)O, ldo, H, Lq, Lk, Lkp, sl, lse, nqt, xfl)
#define M324_ATTN_VR(PS, NWV)                                                                                            \
    hipLaunchKernelGGL((attn_bf16_kernel<PS, 1, NWV, true>), g2, dim3(NWV * 64), pad, s, (const bf16_t*)Q, q_bstride,    \
                       (const bf16_t*)K, (const bf16_t*)Vt, (bf16_t*)O, ldo, H, Lq, Lk, Lkp, sl, lse, nqt, xfl)
        if (two_stage(vrow, q_prescaled != 0, w8, Lk)) {
            // per-frame blocks (round 6): four workgroups per CU instead of three (two LDS stages, 128 registers); microbench, interleaved
            // A/B on one box: L = 324 26.6 -> 24.5 us, L = 257 22.6 -> 21.2 us; M324_ATTN_OCC=3 keeps the three-stage form (A/B)
            if (vrow)
                hipLaunchKernelGGL((attn_bf16_kernel<true, 1, 4, true, 2>), g2, dim3(256), 0, s, (const bf16_t*)Q, q_bstride, (const bf16_t*)K,
                                   (const bf16_t*)Vt, (bf16_t*)O, ldo, H, Lq, Lk, Lkp, sl, lse, nqt, xfl);
            else
                hipLaunchKernelGGL((attn_bf16_kernel<true, 1, 4, false, 2>), g2, dim3(256), 0, s, (const bf16_t*)Q, q_bstride, (const bf16_t*)K,
                                   (const bf16_t*)Vt, (bf16_t*)O, ldo, H, Lq, Lk, Lkp, sl, lse, nqt, xfl);
        } else if (vrow) {
            if (q_prescaled) { if (w8) M324_ATTN_VR(true, 8); else M324_ATTN_VR(true, 4); }
            else { if (w8) M324_ATTN_VR(false, 8); else M324_ATTN_VR(false, 4); }
        } else if (q_prescaled && !w8 && !nq2 && !vrow && Lk <= KV && q_bstride == 0 && B % 2 == 0 && Lq >= 512 && !(xfl & 8)) {
            // shared queries under several batches of one key tile (the decoder): two frames per workgroup (M324_ATTN_EXP bit 3: A/B)
            if (xfl & 16)       // plain stores
                hipLaunchKernelGGL((attn_frames_kernel<2, false>), dim3(ceil_div(Lq, QB), H, B / 2), dim3(256), 0, s, (const bf16_t*)Q,
                                   (const bf16_t*)K, (const bf16_t*)Vt, (bf16_t*)O, ldo, H, Lq, Lk, Lkp, lse);
            else
                hipLaunchKernelGGL((attn_frames_kernel<2, true>), dim3(ceil_div(Lq, QB), H, B / 2), dim3(256), 0, s, (const bf16_t*)Q,
                                   (const bf16_t*)K, (const bf16_t*)Vt, (bf16_t*)O, ldo, H, Lq, Lk, Lkp, lse);
        } else if (q_prescaled && !w8 && !nq2 && Lk <= KV && m324::tunable(m324::TUN_ATTN_OCC) != 1) {      // one tile (M324_ATTN_OCC=1: A/B)
            hipLaunchKernelGGL((attn_bf16_kernel<true, 1, 4, false, 1>), g2, dim3(256), 0, s, (const bf16_t*)Q, q_bstride,
                               (const bf16_t*)K, (const bf16_t*)Vt, (bf16_t*)O, ldo, H, Lq, Lk, Lkp, sl, lse, nqt, xfl);
        } else if (q_prescaled) { if (nq2) M324_ATTN(true, 2, 4); else if (w8) M324_ATTN(true, 1, 8); else M324_ATTN(true, 1, 4); }
        else { if (nq2) M324_ATTN(false, 2, 4); else if (w8) M324_ATTN(false, 1, 8); else M324_ATTN(false, 1, 4); }
#undef M324_ATTN_VR
#undef M324_ATTN
    } else if (dtype == M324_F32) {
        M324_REQUIRE(ldo % 4 == 0, "m324_attention: ldo misaligned");
        hipLaunchKernelGGL(attn_f32_kernel, grid, dim3(256), 0, s, (const float*)Q, q_bstride, (const float*)K,
                           (const float*)Vt, (float*)O, ldo, H, Lq, Lk, Lkp, sl, lse);
    } else {
        M324_FAIL(M324_ERR_UNSUPPORTED, "m324_attention: dtype %d", dtype);
    }
    M324_CHECK_LAUNCH("m324_attention");
    return M324_OK;
}

// Name and grid (threads) of the kernel m324_attention would launch: see m324_gemm_plan.
extern "C" int m324_attention_plan(int B, int H, int Lq, int Lk, int flags, int dtype, char* buf, int n) {
    M324_REQUIRE(buf && n > 0 && B > 0 && H > 0 && Lq > 0 && Lk > 0, "m324_attention_plan: bad arguments");
    if (dtype != M324_BF16) {
        snprintf(buf, (size_t)n, "attn_f32_kernel grid=%ldx%dx%d", (long)ceil_div(Lq, QB) * 256, H, B);
        return 0;
    }
    const bool vrow = (flags & M324_ATTN_V_ROWMAJOR) != 0, ps = (flags & M324_ATTN_Q_PRESCALED) != 0;
    const bool nq2 = m324::tunable(m324::TUN_ATTN_NQ2) != 0 && Lq >= 1024 && !vrow;
    const int fnw = m324::tunable(m324::TUN_ATTN_NW);
    const bool w8 = !nq2 && (fnw ? fnw == 8 : (Lq >= 2048 && Lk >= 512));
    if (use_pwg(ps, vrow, nq2, fnw, Lq, Lk)) {
        snprintf(buf, (size_t)n, "%s grid=%ldx1x1", (flags & M324_ATTN_SCORES_BOUNDED) ? "attn_pwg_bounded_kernel" : "attn_pwg_kernel",
                 (long)ceil_div(Lq, 256) * H * B * 256);
        return 4;
    }
    const long gx = ceil_div(Lq, (nq2 || w8) ? 2 * QB : QB);
    const int nwv = w8 ? 8 : 4;
    const int flat = m324::tunable(m324::TUN_ATTN_FLAT);
    const bool one_tile = ps && !w8 && !nq2 && !vrow && Lk <= KV;
    const int nst = two_stage(vrow, ps, w8, Lk) ? 2 : 3;
    if (flat != 0 && (w8 || (flat != 2 && !nq2 && !one_tile && gx > 1 && gx * H * B >= 512)))
        snprintf(buf, (size_t)n, "attn_bf16_kernel<%s, %d, %d, %s, %d> grid=%ldx1x1", ps ? "true" : "false", nq2 ? 2 : 1, nwv,
                 vrow ? "true" : "false", nst, gx * H * B * nwv * 64);
    else if (one_tile && (flags & 256) && B % 2 == 0 && Lq >= 512 && !(m324::tunable(m324::TUN_ATTN_EXP) & 8))
        snprintf(buf, (size_t)n, "attn_frames_kernel<2, %s> grid=%ldx%dx%d", (m324::tunable(m324::TUN_ATTN_EXP) & 16) ? "false" : "true", gx * 256, H, B / 2);
    else if (ps && !vrow && !w8 && !nq2 && Lk <= KV && m324::tunable(m324::TUN_ATTN_OCC) != 1)
        snprintf(buf, (size_t)n, "attn_bf16_kernel<true, 1, 4, false, 1> grid=%ldx%dx%d", gx * nwv * 64, H, B);
    else
        snprintf(buf, (size_t)n, "attn_bf16_kernel<%s, %d, %d, %s, %d> grid=%ldx%dx%d", ps ? "true" : "false", nq2 ? 2 : 1, nwv,
                 vrow ? "true" : "false", nst, gx * nwv * 64, H, B);
    return nwv;
}

extern "C" int m324_attention_bwd_mfma(const void* Qs, const void* Qst, long q_bstride, long qt_bstride, const void* K,
                                       const void* Kt, const void* V, const void* dO, const void* dOt, const float* lse,
                                       const float* D, void* dQ, void* dK, void* dV, int B, int H, int Lq, int Lk, float scale,
                                       void* stream) {
    M324_REQUIRE(Qs && Qst && K && Kt && V && dO && dOt && lse && D && dQ && dK && dV, "m324_attention_bwd_mfma: null pointer");
    M324_REQUIRE(B > 0 && H > 0 && Lq > 0 && Lk > 0 && H <= 65535 && B <= 65535, "m324_attention_bwd_mfma: bad sizes");
    const int Lkp = (Lk + 63) / 64 * 64, Lqp = (Lq + 63) / 64 * 64;
    hipStream_t s = (hipStream_t)stream;
    // Rounds 3-5 ran eight waves per workgroup for key sets of 1024 and more (every staged tile feeds 256 instead of 128 rows: half the LDS-DMA
    // pieces per wave and tile; measured 1552 us against 1724 at B = 8, L = 3888 at the time -- before lse / D rode with the tiles).
    const int fbw = m324::tunable(m324::TUN_ATTN_BWD_NW);
    // M324_ATTN_BWD_NW: 0 = by key count; 4 | 8 = both kernels with that many waves; 84 = dQ with eight waves, dK / dV with four; 48 = the
    // other way round; 2 = the dQ kernel's 64-queries-per-wave form (four waves) beside the eight-wave dK / dV kernel (A/B runs, tests)
    // Round 6: FOUR waves per workgroup are the default at every size again (two co-resident four-wave workgroups hide more of a tile's waits than
    // one eight-wave workgroup's shared tiles save: dQ + dK/dV at B = 8, L = 3888 1476-1485 us against 1491-1508, B = 32 5757 against 5970;
    // the c3 training step 89.8-90.0 ms against 91.1-91.2, alternated processes on one box; profiles/r06_misc_ab.md section 8).
    const bool big = Lk >= 1024;
    const bool dq8 = fbw == 8 || fbw == 84, dkv8 = fbw == 8 || fbw == 48 || fbw == 2;
    if (fbw == 2 && big)
        hipLaunchKernelGGL(attn_bwd_dq2_mfma_kernel, dim3(ceil_div(Lq, 2 * QB), H, B), dim3(256), 0, s, (const bf16_t*)Qs, q_bstride,
                           (const bf16_t*)K, (const bf16_t*)V, (const bf16_t*)Kt, (const bf16_t*)dO, lse, D, (bf16_t*)dQ, H, Lq, Lk, Lkp, scale);
    else if (dq8)
        hipLaunchKernelGGL(attn_bwd_dq_mfma_kernel<8>, dim3(ceil_div(Lq, 2 * QB), H, B), dim3(512), 0, s, (const bf16_t*)Qs,
                           q_bstride, (const bf16_t*)K, (const bf16_t*)V, (const bf16_t*)Kt, (const bf16_t*)dO, lse, D,
                           (bf16_t*)dQ, H, Lq, Lk, Lkp, scale);
    else
        hipLaunchKernelGGL(attn_bwd_dq_mfma_kernel<4>, dim3(ceil_div(Lq, QB), H, B), dim3(256), 0, s, (const bf16_t*)Qs,
                           q_bstride, (const bf16_t*)K, (const bf16_t*)V, (const bf16_t*)Kt, (const bf16_t*)dO, lse, D,
                           (bf16_t*)dQ, H, Lq, Lk, Lkp, scale);
    if (dkv8)
        hipLaunchKernelGGL(attn_bwd_dkv_mfma_kernel<8>, dim3(ceil_div(Lk, 2 * QB), H, B), dim3(512), 0, s, (const bf16_t*)Qs,
                           (const bf16_t*)Qst, q_bstride, qt_bstride, (const bf16_t*)K, (const bf16_t*)V, (const bf16_t*)dO,
                           (const bf16_t*)dOt, lse, D, (bf16_t*)dK, (bf16_t*)dV, H, Lq, Lk, Lqp);
    else
        hipLaunchKernelGGL(attn_bwd_dkv_mfma_kernel<4>, dim3(ceil_div(Lk, QB), H, B), dim3(256), 0, s, (const bf16_t*)Qs,
                           (const bf16_t*)Qst, q_bstride, qt_bstride, (const bf16_t*)K, (const bf16_t*)V, (const bf16_t*)dO,
                           (const bf16_t*)dOt, lse, D, (bf16_t*)dK, (bf16_t*)dV, H, Lq, Lk, Lqp);
    M324_CHECK_LAUNCH("m324_attention_bwd_mfma");
    return M324_OK;
}
