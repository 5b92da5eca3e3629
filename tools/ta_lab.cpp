// Texture-path lab, round 5: what does the operand traffic of a K = 768 GEMM cost a CU when the operands are L2 / MALL resident
// (the model's shapes: A 10368 x 768, W 3072 x 768 -- dma_lab's 4096^3 operands are not), per path, and beside an MFMA stream?
// One persistent workgroup per CU walks the 256 x 256 tiles of the fc1 GEMM in m324_gemm's XCD-aware order and issues, per
// K-stage of 64, the loads the tile needs -- no fragment reads, no epilogue:
//   A: LDS-DMA pieces (buffer form, 8 rows x 128 B, swizzled like the ring kernels)
//   W: LDS-DMA pieces | fragment-packed global loads straight to VGPRs (1 KiB contiguous per wave-instruction; each fragment by
//      ONE wave, or by the two waves of a 2 x 4 wave grid that share it)
// with 0 or 32 (8 waves) / 64 (4 waves) register-only MFMAs per wave and stage between the loads.
//     hipcc --offload-arch=gfx950 -O3 tools/ta_lab.cpp -o tools/ta_lab ; run on the GPU box.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define HIP_OK(x)                                                                     \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(2);                                                                  \
        }                                                                             \
    } while (0)

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;
typedef __attribute__((address_space(3))) void lds_t;
typedef unsigned short u16;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int K = 768, NS = K / 64;
constexpr int CHUNK = 256 * 128;          // one operand's 256 rows x 64 k

__device__ __forceinline__ void tile_of(int bid, int nblocks, int ntm, int ntn, int mode, int& tm, int& tn) {
    int p = bid;
    tm = tn = 0;
    if (mode & 1) {
        const int q = nblocks >> 3, r = nblocks & 7, x = bid & 7, loc = bid >> 3;
        p = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + loc;
    }
    if (!(mode & 2)) {
        tm = p / ntn;
        tn = p - tm * ntn;
        return;
    }
    const int rq = ntm >> 2, rr = ntm & 3, c0 = (ntn + 1) >> 1;
    int rs = 0;
    for (int rg = 0; rg < 4; ++rg) {
        const int rows = rq + (rg < rr);
        for (int cg = 0; cg < 2; ++cg) {
            const int cols = cg ? ntn - c0 : c0, size = rows * cols;
            if (p >= 0 && p < size) {
                const int rl = p / cols;
                tm = rs + rl;
                tn = (cg ? c0 : 0) + (p - rl * cols);
            }
            p -= size;
        }
        rs += rows;
    }
}

// (target builtins go through __device__ helpers: called directly from a __global__ template the host pass drops the kernel stub)
__device__ __forceinline__ void dma_piece(__amdgpu_buffer_rsrc_t rs, unsigned char* lds, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_t*)lds, 16, voff, soff, 0, 0);
}

// AMODE: 0 none, 1 LDS-DMA.  WMODE: 0 none, 1 LDS-DMA, 2 packed fragments to VGPRs (each by one wave), 3 packed fragments, every
// fragment by two waves.  MF: MFMAs per wave and stage.  TN: tile width in columns (256 | 128: the W chunk is TN rows).
template <int NW, int AMODE, int WMODE, int MF, int TN>
__global__ __launch_bounds__(NW * 64) void k(const u16* __restrict__ A, const u16* __restrict__ W, const u16* __restrict__ Wp, int M, int N,
                                             int xcd_mode, float* out, long long* cyc) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[5 * CHUNK];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntm = (M + 255) / 256, ntn = N / TN, ntiles = ntm * ntn;
    constexpr int PA = AMODE ? 32 / NW : 0;                     // LDS-DMA pieces of A per wave and stage
    constexpr int PW = WMODE == 1 ? (TN / 8) / NW : 0;          // ... of W
    constexpr int FW = WMODE == 2 ? (TN / 8) / NW : (WMODE == 3 ? 2 * (TN / 8) / NW : 0);   // packed 1-KiB fragment loads
    constexpr int PV = PA + PW + FW;                            // vector-memory instructions per wave and stage
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 fa, fb;
    for (int i = 0; i < 8; ++i) {
        fa[i] = (__bf16)(0.01f * (float)((lane * 7 + i * 3) % 97 - 48));
        fb[i] = (__bf16)(0.02f * (float)((lane * 5 + i * 11) % 89 - 44));
    }
    u32x4 wv[FW > 0 ? FW : 1];
    for (int i = 0; i < (FW > 0 ? FW : 1); ++i) wv[i] = (u32x4)(0u);
    unsigned va[PA > 0 ? PA : 1], vb[PW > 0 ? PW : 1];
    const long long t0 = clock64();
    int q = 0;                                                  // chunk counter: ring position q % 5
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        int tm, tn;
        tile_of(t, ntiles, ntm, ntn, xcd_mode, tm, tn);
        const int m0 = tm * 256, n0 = tn * TN;
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const int r = (wave * PA + i) * 8 + (lane >> 3);
            va[i] = (unsigned)(((long)min(r, M - 1 - m0) * K + ((lane & 7) ^ ((r >> 1) & 7)) * 8) * 2);
        }
#pragma unroll
        for (int i = 0; i < PW; ++i) {
            const int r = (wave * PW + i) * 8 + (lane >> 3);
            vb[i] = (unsigned)(((long)r * K + ((lane & 7) ^ ((r >> 1) & 7)) * 8) * 2);
        }
        const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(A + (long)m0 * K), 0, 0x7FFFFFFF, 0x00020000);
        const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(W + (long)n0 * K), 0, 0x7FFFFFFF, 0x00020000);
        // packed W: [N / 32][K / 16][64 lanes][8]; this wave's fragments of a stage: n-blocks x 4 k-steps
        //   WMODE 2: wave w owns n-blocks w * (TN / 32 / NW) ...;  WMODE 3: a 2 x (NW / 2) wave grid, wave column w % (NW / 2)
        constexpr int NBW = FW / 4 > 0 ? FW / 4 : 1;            // n-blocks per wave
        const int nb0 = n0 / 32 + (WMODE == 3 ? (wave % (NW / 2)) : wave) * NBW;
        const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(Wp), 0, 0x7FFFFFFF, 0x00020000);
        for (int s = 0; s < NS; ++s) {
            unsigned char* ca = smem + (q % 5) * CHUNK + wave * PA * 1024;
            if (AMODE) ++q;
            unsigned char* cw = smem + (q % 5) * CHUNK + wave * PW * 1024;
            if (WMODE == 1) ++q;
            auto mfmas = [&](int n) {
#pragma unroll
                for (int i = 0; i < n; ++i) acc[i & 7] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[i & 7], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            };
            constexpr int PVD = PV > 0 ? PV : 1, MPP = PV > 0 ? MF / PVD : 0;           // MFMAs behind every vector-memory instruction
#pragma unroll
            for (int i = 0; i < PA; ++i) {
                dma_piece(ra, ca + i * 1024, va[i], (unsigned)(s * 128));
                __builtin_amdgcn_sched_barrier(0);
                mfmas(MPP);
            }
#pragma unroll
            for (int i = 0; i < PW; ++i) {
                dma_piece(rb, cw + i * 1024, vb[i], (unsigned)(s * 128));
                __builtin_amdgcn_sched_barrier(0);
                mfmas(MPP);
            }
#pragma unroll
            for (int i = 0; i < FW; ++i) {
                const unsigned off = (unsigned)((((nb0 + i / 4) * (K / 16) + s * 4 + (i & 3)) * 1024 + lane * 16));
                // "+v": the destination stays owned by wv[i] from here to the final wait -- with a plain output the compiler reuses the
                // register of an earlier iteration while its load is still in flight (the late data then lands in an address)
                asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "+v"(wv[i]) : "v"(off), "s"(rp) : "memory");
                __builtin_amdgcn_sched_barrier(0);
                mfmas(MPP);
            }
            mfmas(MF - MPP * PV);
            // at most two stages of this wave in flight
            if (PV > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PV < 32 ? 2 * PV : 63) : "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t1 = clock64();
    float sres = 0;
    for (int i = 0; i < 8; ++i) sres += acc[i][0] + acc[i][5];
    for (int i = 0; i < (FW > 0 ? FW : 1); ++i) sres += (float)(wv[i].x & 1);
    out[blockIdx.x * NW * 64 + tid] = sres + (float)smem[tid];
    if (tid == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int NW, int AMODE, int WMODE, int MF, int TN = 256>
static void run(const char* name, const u16* A, const u16* W, const u16* Wp, int M, int N, int xcd_mode, float* out, long long* cyc) {
    hipEvent_t e0, e1;
    HIP_OK(hipEventCreate(&e0));
    HIP_OK(hipEventCreate(&e1));
    const int ntiles = ((M + 255) / 256) * (N / TN), grid = ntiles < 256 ? ntiles : 256;
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL((k<NW, AMODE, WMODE, MF, TN>), dim3(grid), dim3(NW * 64), 0, 0, A, W, Wp, M, N, xcd_mode, out, cyc);
        HIP_OK(hipEventRecord(e0));
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((k<NW, AMODE, WMODE, MF, TN>), dim3(grid), dim3(NW * 64), 0, 0, A, W, Wp, M, N, xcd_mode, out, cyc);
        HIP_OK(hipEventRecord(e1));
        HIP_OK(hipEventSynchronize(e1));
        float ms;
        HIP_OK(hipEventElapsedTime(&ms, e0, e1));
        if (ms / 10 < best) best = ms / 10;
    }
    long long c;
    HIP_OK(hipDeviceSynchronize());
    HIP_OK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
    const double stage_bytes = (AMODE ? 32768.0 : 0.0) + (WMODE == 1 || WMODE == 2 ? TN * 128.0 : (WMODE == 3 ? 2 * TN * 128.0 : 0.0));
    const int rounds = (ntiles + grid - 1) / grid;               // tiles of the busiest workgroup
    const double cu_bytes = stage_bytes * NS * rounds;
    const double mhz = c / (best * 1e3);
    const double mfma_cycles = (double)rounds * NS * MF * (NW / 4.0) * 32.0;
    printf("%-58s %d waves  %7.1f us  %6.0f MHz  %8lld ticks: %6.0f per stage  %5.1f B/clk/CU  %5.1f cycles/KiB  MFMA alone %6.0f per stage\n", name, NW,
           best * 1e3, mhz, c, (double)c / (rounds * NS), cu_bytes / (double)c, (double)c / (cu_bytes / 1024.0 + 1e-9),
           mfma_cycles / (rounds * NS));
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 10368, N = argc > 2 ? atoi(argv[2]) : 3072;
    setvbuf(stdout, nullptr, _IONBF, 0);
    u16 *A, *W, *Wp;
    float* out;
    long long* cyc;
    HIP_OK(hipMalloc(&A, (size_t)M * K * 2));
    HIP_OK(hipMalloc(&W, (size_t)N * K * 2));
    HIP_OK(hipMalloc(&Wp, (size_t)N * K * 2));
    std::vector<u16> h((size_t)M * K);
    unsigned x = 12345;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (u16)(0x3c00 + ((x >> 9) & 0x3ff) + ((x >> 31) << 15)); }
    HIP_OK(hipMemcpy(A, h.data(), (size_t)M * K * 2, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(W, h.data(), (size_t)N * K * 2, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(Wp, h.data() + 4096, (size_t)N * K * 2, hipMemcpyHostToDevice));
    HIP_OK(hipMalloc(&out, 256 * 512 * 4));
    HIP_OK(hipMalloc(&cyc, 8));
    printf("M = %d, N = %d, K = %d; 256 x 256 tiles unless noted, persistent, XCD mode as given\n", M, N, K);
    for (int xm = 1; xm <= 3; xm += 2) {
        printf("-- tile order mode %d\n", xm);
#define R(NW, AM, WM, MF, name) run<NW, AM, WM, MF>(name, A, W, Wp, M, N, xm, out, cyc)
        R(8, 1, 1, 0, "A + W by LDS-DMA");
        R(4, 1, 1, 0, "A + W by LDS-DMA");
        R(8, 1, 0, 0, "A by LDS-DMA only");
        R(4, 1, 0, 0, "A by LDS-DMA only");
        R(8, 0, 1, 0, "W by LDS-DMA only");
        R(8, 0, 2, 0, "W packed -> VGPR only (each fragment once)");
        R(4, 0, 2, 0, "W packed -> VGPR only (each fragment once)");
        R(8, 1, 2, 0, "A LDS-DMA + W packed -> VGPR (once)");
        R(4, 1, 2, 0, "A LDS-DMA + W packed -> VGPR (once)");
        R(8, 1, 3, 0, "A LDS-DMA + W packed -> VGPR (twice: 2 x 4 wave grid)");
        R(8, 0, 0, 32, "MFMA only");
        R(4, 0, 0, 64, "MFMA only");
        R(8, 1, 1, 32, "A + W by LDS-DMA + MFMA");
        R(4, 1, 1, 64, "A + W by LDS-DMA + MFMA");
        R(8, 1, 2, 32, "A LDS-DMA + W packed -> VGPR (once) + MFMA");
        R(4, 1, 2, 64, "A LDS-DMA + W packed -> VGPR (once) + MFMA");
        R(8, 1, 3, 32, "A LDS-DMA + W packed -> VGPR (twice) + MFMA");
#undef R
        // 256 x 128 tiles (the traffic of two half-workgroups out of phase): MFMA per stage halves
        run<4, 1, 1, 0, 128>("256 x 128 tiles: A + W by LDS-DMA", A, W, Wp, M, N, xm, out, cyc);
        run<4, 1, 1, 32, 128>("256 x 128 tiles: A + W by LDS-DMA + MFMA", A, W, Wp, M, N, xm, out, cyc);
        run<4, 1, 2, 32, 128>("256 x 128 tiles: A LDS-DMA + W packed -> VGPR + MFMA", A, W, Wp, M, N, xm, out, cyc);
    }
    return 0;
}
