// Shader-clock lab for gfx950: does the core clock drop when LDS / texture traffic is added next to the MFMA stream?
// Every mode runs 256 workgroups x NW waves of a K-loop of 32x32x16 bf16 MFMAs on register operands; modes add
// fragment reads (ds_read_b128) and LDS-DMA pieces at the ratio of a 256 x 256 GEMM tile.  clock64() counts shader
// cycles, wall time comes from HIP events: cycles / time = the clock the kernel actually ran at.
//     hipcc --offload-arch=gfx950 -O2 tools/clk_lab.cpp -o tools/clk_lab
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

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
typedef __attribute__((address_space(1))) void glb_t;
typedef __attribute__((address_space(3))) void lds_t;

constexpr int ITER = 4096;     // k-steps per wave

// NW waves per workgroup; per k-step a wave does MF MFMAs, RD fragment reads and (every other k-step) DMA LDS-DMA pieces
template <int NW, int MF, int RD, int DMA>
__global__ __launch_bounds__(NW * 64) void k(const unsigned short* __restrict__ g, float* out, long long* cyc) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[131072];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 131072 / 4; i += NW * 64) reinterpret_cast<unsigned*>(smem)[i] = 0x3c003c00u + i;
    __syncthreads();
    constexpr int NA = MF >= 16 ? 4 : 4, NB = MF / NA;
    f32x16 acc[NA][NB];
    for (int i = 0; i < NA; ++i)
        for (int j = 0; j < NB; ++j)
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    bf16x8 fr[8];
    for (int i = 0; i < 8; ++i) fr[i] = *reinterpret_cast<const bf16x8*>(smem + ((lane * 16 + i * 4096 + wave * 1024) & 131071));
    const unsigned short* gp = g + ((long)blockIdx.x * 4096 + wave * 512 + lane * 8);
    const long long t0 = clock64();
    for (int it = 0; it < ITER; ++it) {
        const int base = ((it & 15) * 8192 + wave * 1024 + (lane & 31) * 128 + ((lane >> 5) ^ ((lane >> 1) & 7)) * 16) & 131071;
#pragma unroll
        for (int i = 0; i < RD; ++i) fr[i] = *reinterpret_cast<const bf16x8*>(smem + ((base + i * 4096) & 131071));
        if (DMA) {
#pragma unroll
            for (int d = 0; d < DMA; ++d)
                __builtin_amdgcn_global_load_lds((glb_t*)(gp + (it & 63) * 64 + d * 32), (lds_t*)(smem + ((it & 15) * 8192 + wave * 1024) % 131072),
                                                 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NA; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[4 + (j & 3)], fr[i & 3], acc[i][j], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t1 = clock64();
    float s = 0;
    for (int i = 0; i < NA; ++i)
        for (int j = 0; j < NB; ++j) s += acc[i][j][0] + acc[i][j][7];
    out[blockIdx.x * NW * 64 + tid] = s;
    if (tid == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int NW, int MF, int RD, int DMA>
static void run(const char* name, const unsigned short* g, float* out, long long* cyc) {
    hipEvent_t e0, e1;
    HIP_OK(hipEventCreate(&e0));
    HIP_OK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<NW, MF, RD, DMA>), dim3(256), dim3(NW * 64), 0, 0, g, out, cyc);
    HIP_OK(hipEventRecord(e0));
    for (int w = 0; w < 5; ++w) hipLaunchKernelGGL((k<NW, MF, RD, DMA>), dim3(256), dim3(NW * 64), 0, 0, g, out, cyc);
    HIP_OK(hipEventRecord(e1));
    HIP_OK(hipEventSynchronize(e1));
    float ms;
    HIP_OK(hipEventElapsedTime(&ms, e0, e1));
    ms /= 5;
    long long c;
    HIP_OK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
    const double flops = 256.0 * NW * ITER * MF * 32768.0;
    printf("%-44s %2d waves: %8.1f us  %7.0f TF/s  %10lld clock64 ticks = %6.1f MHz tick rate  (MFMA pipe busy %4.1f %% of ticks)\n", name,
           NW, ms * 1e3, flops / (ms * 1e-3) / 1e12, c, c / (ms * 1e3), 100.0 * (NW / 4.0) * ITER * MF * 32.0 / c);
}

int main() {
    unsigned short* g;
    float* out;
    long long* cyc;
    HIP_OK(hipMalloc(&g, 256 * 4096 * 2 * 4));
    HIP_OK(hipMemset(g, 0x3c, 256 * 4096 * 2 * 4));
    HIP_OK(hipMalloc(&out, 256 * 1024 * 4));
    HIP_OK(hipMalloc(&cyc, 8));
    run<8, 8, 0, 0>("MFMA only (8 per k-step)", g, out, cyc);
    run<8, 8, 6, 0>("MFMA + 6 ds_read_b128 per 8 MFMA", g, out, cyc);
    run<8, 8, 6, 2>("MFMA + 6 reads + 2 LDS-DMA pieces", g, out, cyc);
    run<8, 8, 0, 2>("MFMA + 2 LDS-DMA pieces", g, out, cyc);
    run<4, 16, 0, 0>("MFMA only (16 per k-step)", g, out, cyc);
    run<4, 16, 8, 0>("MFMA + 8 ds_read_b128 per 16 MFMA", g, out, cyc);
    run<4, 16, 8, 4>("MFMA + 8 reads + 4 LDS-DMA pieces", g, out, cyc);
    run<4, 16, 0, 4>("MFMA + 4 LDS-DMA pieces", g, out, cyc);
    return 0;
}
