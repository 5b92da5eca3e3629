// Lab copy of motion324_amd/csrc/gemm_pp.hip's kernel (schedule v14) with in-kernel stamps and ablation switches, as a stand-alone
// program:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form tools/lab_src/pp_lab.hip -o tools/pp_lab
// Stamps (wave 0 of a workgroup): start; per tile: before the X_0 wait, behind it, end of the main loop, end of the epilogue.
#include "../../motion324_amd/csrc/gemm_tile.h"

namespace {

constexpr int CH14 = 128 * ROWB;                  // 16 KiB

// LAB: ABL bits: 1 no epilogue, 2 no MFMA, 4 no LDS-DMA in the loop, 8 no fragment reads
template <typename TOUT, int ACT, int RES, int ABL, int PRIO = 0>
__global__ __launch_bounds__(256, 2) void gemm_pp_kernel(const bf16_t* __restrict__ A, long lda, const bf16_t* __restrict__ W, long ldw,
                                                         TOUT* C, long ldc, int M, int N, int K, Epilogue ep, int ntn, int ntiles,
                                                         int xcd_remap, int skew, long long* trace) {
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
    auto issue2 = [&](int which, int i0, int st, int pos) {
        unsigned char* d = smem + pos * CH14 + wave * 4096 + i0 * 1024;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const unsigned g = which == 0 ? gaa[i0 + i] : (which == 1 ? gw[i0 + i] : gab[i0 + i]);
            if (!(ABL & 4) || st == 0) dma_piece(which == 1 ? rb : ra, d + i * 1024, g, (unsigned)(st * 128));
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
        if constexpr ((ABL & 8) != 0) return;
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
                    if constexpr ((ABL & 2) == 0) acc[ib + ii][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[2 * (ph & 1) + q][j], fa[set][q][ii], acc[ib + ii][j], 0, 0, 0);
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

    long long* tr = trace ? trace + (long)blockIdx.x * 64 : nullptr;
    int ti = 0;
    auto stamp = [&]() { if (tr && lane == 0 && wave == 0 && ti < 64) tr[ti++] = clock64(); };
    stamp();
    if constexpr (PRIO == 2) __builtin_amdgcn_s_setprio(3);
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
        stamp();
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");      // X_0: Aa_0, W_0 landed (Ab_0 may fly)
        M324_BARRIER();
        stamp();
        int pa = 3;                                           // ring position of Aa_s; W_s at pa + 1, Ab_s at pa + 2 (mod 5)
        for (int s = 0; s < NS; ++s) {
            int pw = pa + 1, pb = pa + 2, pan = pa + 3, pwn = pa + 4;      // ... of Aa_{s+1}, W_{s+1}; Ab_{s+1} takes Aa_s's place
            pw = pw >= 5 ? pw - 5 : pw;
            pb = pb >= 5 ? pb - 5 : pb;
            pan = pan >= 5 ? pan - 5 : pan;
            pwn = pwn >= 5 ? pwn - 5 : pwn;
            const int sn = s + 1 < NS ? s + 1 : NS - 1;       // past the end the last stage is fetched again (never read)
            load_frags(0, pa, pw);
            issue2(0, 0, sn, pan); issue2(1, 0, sn, pwn);
            mma8(3);                                          // (s-1, second half, k-steps 2, 3); zeros in a tile's first stage
            sched_a();
            load_frags(1, pa, pw);
            issue2(0, 2, sn, pan); issue2(1, 2, sn, pwn);
            mma8(0);
            sched_a();
            asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");    // Y_s: Ab_s landed, Aa_s is in registers
            M324_BARRIER();
            load_frags(2, pb, pw);
            issue2(2, 0, sn, pa);
            mma8(1);
            sched_b();
            load_frags(3, pb, pw);
            issue2(2, 2, sn, pa);
            mma8(2);
            sched_b();
            asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");    // X_{s+1}: Aa_{s+1}, W_{s+1} landed; stage s is in registers
            M324_BARRIER();
            pa = pan;
        }
        mma8(3);                                              // (NS-1, second half, k-steps 2, 3)
        stamp();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // no LDS-DMA may outlive the main loop: the ring becomes scratch
        M324_BARRIER();
        if (more) {                                           // the next tile's first chunks land under this tile's epilogue
            tile_setup(t + gridDim.x);
            issue2(0, 0, 0, 3); issue2(0, 2, 0, 3);
            issue2(1, 0, 0, 4); issue2(1, 2, 0, 4);
        }
        if constexpr (PRIO == 1) __builtin_amdgcn_s_setprio(3);
        if constexpr (PRIO == 2) __builtin_amdgcn_s_setprio(0);
        if constexpr ((ABL & 1) == 0)
            store_tile_lds<TOUT, ACT, RES, 4>(acc, reinterpret_cast<float*>(smem) + wave * ep_wave_floats(ACT), C, ldc, M, N, mt + wm * 128,
                                              nt + wn * 64, lane, ep);
        else if (acc[0][0][0] == 1.2345f) C[0] = (TOUT)1;
        if constexpr (PRIO == 1) __builtin_amdgcn_s_setprio(0);
        if constexpr (PRIO == 2) __builtin_amdgcn_s_setprio(3);
        stamp();
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

#include <cstdio>
#include <cstdlib>
#include <vector>
#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)
void m324_set_error(const char*, ...) {}
int m324::tunable(int) { return 0; }

template <int ACT, int ABL, int PRIO = 0>
static void run(const char* name, const bf16_t* A, const bf16_t* W, bf16_t* C, const float* bias, int M, int N, int K, int skew, int xcd, long long* trace, bool dump) {
    Epilogue ep{};
    ep.bias = bias;
    ep.act = ACT == 1 ? M324_ACT_GELU : 0;
    const int ntn = (N + 127) / 128, ntiles = ntn * ((M + 255) / 256);
    const int grid = ntiles < 512 ? ntiles : 512;
    hipEvent_t e0, e1;
    HIP_OK(hipEventCreate(&e0));
    HIP_OK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL((gemm_pp_kernel<bf16_t, ACT, 0, ABL, PRIO>), dim3(grid), dim3(256), 0, 0, A, (long)K, W, (long)K, C, (long)N, M, N, K, ep, ntn, ntiles, xcd, skew, (long long*)nullptr);
        HIP_OK(hipEventRecord(e0));
        for (int i = 0; i < 10; ++i)
            hipLaunchKernelGGL((gemm_pp_kernel<bf16_t, ACT, 0, ABL, PRIO>), dim3(grid), dim3(256), 0, 0, A, (long)K, W, (long)K, C, (long)N, M, N, K, ep, ntn, ntiles, xcd, skew, (long long*)nullptr);
        HIP_OK(hipEventRecord(e1));
        HIP_OK(hipEventSynchronize(e1));
        float ms;
        HIP_OK(hipEventElapsedTime(&ms, e0, e1));
        if (ms / 10 < best) best = ms / 10;
    }
    printf("%-44s skew %2d: %7.1f us  %6.0f TF/s\n", name, skew, best * 1e3, 2.0 * M * N * K / (best * 1e-3) / 1e12);
    if (!dump) return;
    HIP_OK(hipMemset(trace, 0, 512 * 64 * 8));
    hipLaunchKernelGGL((gemm_pp_kernel<bf16_t, ACT, 0, ABL, PRIO>), dim3(grid), dim3(256), 0, 0, A, (long)K, W, (long)K, C, (long)N, M, N, K, ep, ntn, ntiles, xcd, skew, trace);
    HIP_OK(hipDeviceSynchronize());
    std::vector<long long> h(512 * 64);
    HIP_OK(hipMemcpy(h.data(), trace, 512 * 64 * 8, hipMemcpyDeviceToHost));
    for (int pair = 0; pair < 2; ++pair) {
        const int b0 = pair * 8 * 5;                  // workgroups b0 and b0 + 256 share a CU (tools/place_lab)
        const long long t0 = h[b0 * 64];
        for (int b : {b0, b0 + 256}) {
            if (b >= grid) continue;
            printf("  wg %3d:", b);
            const long long* t = &h[(long)b * 64];
            // stamps: [start], then per tile: before X_0 wait, after X_0, main loop end, epilogue end
            printf(" start %+lld |", t[0] - t0);
            for (int i = 1; i + 3 < 64 && t[i]; i += 4)
                printf(" tile: wait %lld, main %lld (%.0f / stage), epi %lld, ends at %+lld |", t[i + 1] - t[i], t[i + 2] - t[i + 1], (double)(t[i + 2] - t[i + 1]) / (K / 64),
                       t[i + 3] - t[i + 2], t[i + 3] - t0);
            printf("\n");
        }
    }
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 10368, N = argc > 2 ? atoi(argv[2]) : 3072, K = argc > 3 ? atoi(argv[3]) : 768;
    const int xcd = argc > 4 ? atoi(argv[4]) : 3;
    setvbuf(stdout, nullptr, _IONBF, 0);
    bf16_t *A, *W, *C;
    float* bias;
    long long* trace;
    HIP_OK(hipMalloc(&A, (size_t)M * K * 2));
    HIP_OK(hipMalloc(&W, (size_t)N * K * 2));
    HIP_OK(hipMalloc(&C, (size_t)M * N * 2));
    HIP_OK(hipMalloc(&bias, (size_t)N * 4));
    HIP_OK(hipMalloc(&trace, 512 * 64 * 8));
    std::vector<bf16_t> h((size_t)(M > N ? M : N) * K);
    unsigned x = 12345;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (bf16_t)(0x3c00 + ((x >> 9) & 0x3ff) + ((x >> 31) << 15) - ((x >> 13) & 0x400)); }
    HIP_OK(hipMemcpy(A, h.data(), (size_t)M * K * 2, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(W, h.data() + 1000, (size_t)N * K * 2 - 2000, hipMemcpyHostToDevice));
    HIP_OK(hipMemset(bias, 0, (size_t)N * 4));
    printf("M = %d N = %d K = %d, xcd mode %d; stamps in clock64 ticks of wave 0\n", M, N, K, xcd);
    for (int skew : {0}) {
        run<0, 0>("plain", A, W, C, bias, M, N, K, skew, xcd, trace, true);
        run<1, 0>("GELU", A, W, C, bias, M, N, K, skew, xcd, trace, true);
    }
    run<0, 0, 1>("plain, epilogue at priority 3", A, W, C, bias, M, N, K, 0, xcd, trace, true);
    run<1, 0, 1>("GELU, epilogue at priority 3", A, W, C, bias, M, N, K, 0, xcd, trace, true);
    run<0, 0, 2>("plain, main loop at priority 3", A, W, C, bias, M, N, K, 0, xcd, trace, true);
    run<1, 0, 2>("GELU, main loop at priority 3", A, W, C, bias, M, N, K, 0, xcd, trace, true);
    run<1, 1>("GELU kernel without epilogue", A, W, C, bias, M, N, K, 0, xcd, trace, true);
    run<1, 1 | 8>("... and without fragment reads", A, W, C, bias, M, N, K, 0, xcd, trace, false);
    run<1, 1 | 4>("... without LDS-DMA in the loop", A, W, C, bias, M, N, K, 0, xcd, trace, false);
    run<1, 1 | 2>("... without MFMAs", A, W, C, bias, M, N, K, 0, xcd, trace, false);
    run<1, 2>("GELU, no MFMAs (epilogue + traffic)", A, W, C, bias, M, N, K, 0, xcd, trace, true);
    return 0;
}
