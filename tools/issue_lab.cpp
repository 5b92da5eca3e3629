// Issue lab for gfx950 (round 4): does plain VALU work issue in the shadow of a wave's OWN MFMAs?  Round 3's coissue_lab said no
// (32 + 2.3 N cycles per MFMA with N v_fma_f32 behind it, compiler-scheduled, two accumulators in VGPRs).  This lab hand-places
// the stream in inline asm and varies what the first lab held fixed: where the accumulators live (VGPR / AGPR), how many
// independent accumulators rotate (2 / 4 / 8), the filler instruction, and waves per SIMD.
//   hipcc --offload-arch=gfx950 -O2 tools/issue_lab.cpp -o tools/issue_lab
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

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
constexpr int ITER = 2000;

// filler kinds
enum { F_FMA = 0, F_ADD = 1, F_MAX3 = 2, F_EXP = 3, F_CVT = 4, F_MOV = 5, F_PKADD = 6, F_SUB = 7 };

template <int KIND>
__device__ __forceinline__ void filler(float& x, float k1, float k2) {
    if (KIND == F_FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(k1), "v"(k2));
    if (KIND == F_ADD) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(k2));
    if (KIND == F_MAX3) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(k1), "v"(k2));
    if (KIND == F_EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
    if (KIND == F_CVT) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(x) : "v"(k1));
    if (KIND == F_MOV) asm volatile("v_mov_b32 %0, %1" : "+v"(x) : "v"(k1));
    if (KIND == F_SUB) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x) : "v"(k2));
}

template <bool AGPR>
__device__ __forceinline__ void mfma(f32x16& c, bf16x8 a, bf16x8 b) {
    if (AGPR) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}

// NACC independent accumulators rotate; every MFMA is followed by NF fillers of KIND on NF distinct registers (independent of
// each other and of the MFMA).  EXTRA = 1 adds two v_exp_f32 per gap on top (the softmax's ratio).
template <bool AGPR, int NACC, int NF, int KIND, int EXTRA>
__global__ __launch_bounds__(512) void kmix(float seed, float* out, long long* stamp, int nwaves) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (wave >= nwaves) return;
    f32x16 c[NACC];
    for (int j = 0; j < NACC; ++j)
        for (int i = 0; i < 16; ++i) c[j][i] = seed * (j + 1);
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) a[i] = (__bf16)(seed + i + lane), b[i] = (__bf16)(seed - i);
    float x[8], y[2];
    for (int i = 0; i < 8; ++i) x[i] = seed * (lane + i + 1) * 1e-3f;
    y[0] = seed * 1e-4f * lane, y[1] = seed * 2e-4f * lane;
    float k1 = 0.999f + seed * 1e-9f, k2 = 0.001f * seed;
    asm volatile("" : "+v"(k1), "+v"(k2));
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            mfma<AGPR>(c[u % NACC], a, b);
#pragma unroll
            for (int f = 0; f < NF; ++f) filler<KIND>(x[f & 7], k1, k2);
            if (EXTRA) {
                filler<F_EXP>(y[0], k1, k2);
                filler<F_EXP>(y[1], k1, k2);
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float acc = 0.f;
    for (int j = 0; j < NACC; ++j)
        for (int i = 0; i < 16; ++i) acc += c[j][i];
    for (int i = 0; i < 8; ++i) acc += x[i];
    acc += y[0] + y[1];
    out[(blockIdx.x & 255) * 512 + threadIdx.x] = acc;
    if (lane == 0 && blockIdx.x == 17) stamp[wave] = t1 - t0;
}

static const char* KNAME[] = {"v_fma_f32", "v_add_f32", "v_max3_f32", "v_exp_f32", "v_cvt_pk_bf16", "v_mov_b32", "v_pk_add", "v_sub_f32"};

template <bool AGPR, int NACC, int NF, int KIND, int EXTRA>
static void run(float* d, long long* st) {
    for (int cfg = 0; cfg < 3; ++cfg) {   // 1, 2, 4 waves per SIMD
        const int nw = cfg == 0 ? 4 : 8, blocks = cfg == 2 ? 512 : 256;
        if (cfg == 2 && NACC * 16 + 40 > 128) continue;
        HIP_OK(hipMemset(st, 0, 64 * 8));
        hipEvent_t e0, e1;
        HIP_OK(hipEventCreate(&e0));
        HIP_OK(hipEventCreate(&e1));
        hipLaunchKernelGGL((kmix<AGPR, NACC, NF, KIND, EXTRA>), dim3(blocks), dim3(512), 0, 0, 1.0f, d, st, nw);
        HIP_OK(hipEventRecord(e0));
        hipLaunchKernelGGL((kmix<AGPR, NACC, NF, KIND, EXTRA>), dim3(blocks), dim3(512), 0, 0, 1.0f, d, st, nw);
        HIP_OK(hipEventRecord(e1));
        HIP_OK(hipEventSynchronize(e1));
        float ms;
        HIP_OK(hipEventElapsedTime(&ms, e0, e1));
        long long h[8];
        HIP_OK(hipMemcpy(h, st, 64, hipMemcpyDeviceToHost));
        const int wps = cfg == 0 ? 1 : (cfg == 1 ? 2 : 4);
        // kernel time per MFMA issued on one SIMD, in units of the wave's own cycle counter: the wave's elapsed cycles / its MFMAs
        // (1 wave per SIMD), or kernel-time based for several waves (they run concurrently): us * 1e-6 * f / (MFMAs per SIMD)
        const double per_wave = (double)h[0] / (ITER * 8.0);
        printf("  acc %s x%d | MFMA + %d %-14s%s | %d wave/SIMD: %6.1f cyc per MFMA in wave 0's stream, %7.1f us kernel = %5.2f ns per SIMD-MFMA\n",
               AGPR ? "AGPR" : "VGPR", NACC, NF, KNAME[KIND], EXTRA ? " + 2 exp" : "        ", wps, per_wave, ms * 1e3,
               ms * 1e6 / (ITER * 8.0 * wps * (cfg == 2 ? 2 : 1)));
    }
}

int main() {
    float* d;
    long long* st;
    HIP_OK(hipMalloc(&d, 256 * 512 * 4));
    HIP_OK(hipMalloc(&st, 64 * 8));
    printf("bare MFMA streams\n");
    run<false, 2, 0, F_FMA, 0>(d, st);
    run<false, 4, 0, F_FMA, 0>(d, st);
    run<true, 2, 0, F_FMA, 0>(d, st);
    run<true, 4, 0, F_FMA, 0>(d, st);
    printf("v_fma_f32 fillers, VGPR accumulators\n");
    run<false, 2, 2, F_FMA, 0>(d, st);
    run<false, 2, 4, F_FMA, 0>(d, st);
    run<false, 4, 2, F_FMA, 0>(d, st);
    run<false, 4, 4, F_FMA, 0>(d, st);
    run<false, 4, 6, F_FMA, 0>(d, st);
    run<false, 8, 4, F_FMA, 0>(d, st);
    printf("v_fma_f32 fillers, AGPR accumulators\n");
    run<true, 2, 2, F_FMA, 0>(d, st);
    run<true, 2, 4, F_FMA, 0>(d, st);
    run<true, 4, 2, F_FMA, 0>(d, st);
    run<true, 4, 4, F_FMA, 0>(d, st);
    run<true, 4, 6, F_FMA, 0>(d, st);
    run<true, 8, 4, F_FMA, 0>(d, st);
    printf("other fillers (4 accumulators)\n");
    run<false, 4, 4, F_ADD, 0>(d, st);
    run<true, 4, 4, F_ADD, 0>(d, st);
    run<false, 4, 4, F_MAX3, 0>(d, st);
    run<true, 4, 4, F_MAX3, 0>(d, st);
    run<false, 4, 4, F_MOV, 0>(d, st);
    run<true, 4, 4, F_MOV, 0>(d, st);
    run<false, 4, 4, F_CVT, 0>(d, st);
    run<true, 4, 4, F_CVT, 0>(d, st);
    run<false, 4, 2, F_EXP, 0>(d, st);
    run<true, 4, 2, F_EXP, 0>(d, st);
    printf("softmax ratio: 3-4 plain + 2 exp per MFMA\n");
    run<false, 4, 3, F_FMA, 1>(d, st);
    run<true, 4, 3, F_FMA, 1>(d, st);
    run<false, 4, 4, F_FMA, 1>(d, st);
    run<true, 4, 4, F_FMA, 1>(d, st);
    return 0;
}
