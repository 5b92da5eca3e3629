// Lab for schedule v15 (csrc/gemm_hp.hip): the product stream beside its timing-only ablation streams (gen_gemm_hp.py --lab) and the
// stamped stream.  Timing-only streams write wrong / no results.
//   python3 motion324_amd/csrc/gen_gemm_hp.py --lab
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form tools/lab_src/hp_lab.hip -o tools/hp_lab
#include "../../motion324_amd/csrc/gemm_tile.h"

namespace {
typedef __attribute__((ext_vector_type(4))) int i32x4;
__device__ __forceinline__ i32x4 rsrc_words(const void* base, long bytes) {
    const unsigned long p = (unsigned long)base;
    i32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)p);
    r[1] = __builtin_amdgcn_readfirstlane((int)((p >> 32) & 0xffffu));
    r[2] = __builtin_amdgcn_readfirstlane((int)(bytes > 0x7FFFFFFFl ? 0x7FFFFFFFl : bytes));
    r[3] = 0x00020000;
    return r;
}
constexpr int HP_STAGE = 49152;
constexpr int HP_TABLE_BYTES = 8192;
constexpr int HP_SCRATCH = 2048;
#define HP_TRACE 0
#define HP_ST_FLAG ""
#define HP_ONE(NAME, INC) 
#define HP_KERNEL k_full
#define HP_ASM_INC "../../motion324_amd/csrc/gemm_hp_gelu.inc"
#include "gemm_hp_lab_kernel.inl"
#undef HP_KERNEL
#undef HP_ASM_INC
#define HP_KERNEL k_plain
#define HP_ASM_INC "../../motion324_amd/csrc/gemm_hp_plain.inc"
#include "gemm_hp_lab_kernel.inl"
#undef HP_KERNEL
#undef HP_ASM_INC
#define HP_KERNEL k_lab1
#define HP_ASM_INC "gemm_hp_lab1.inc"
#include "gemm_hp_lab_kernel.inl"
#undef HP_KERNEL
#undef HP_ASM_INC
#define HP_KERNEL k_lab2
#define HP_ASM_INC "gemm_hp_lab2.inc"
#include "gemm_hp_lab_kernel.inl"
#undef HP_KERNEL
#undef HP_ASM_INC
#define HP_KERNEL k_lab4
#define HP_ASM_INC "gemm_hp_lab4.inc"
#include "gemm_hp_lab_kernel.inl"
#undef HP_KERNEL
#undef HP_ASM_INC
#define HP_KERNEL k_lab5
#define HP_ASM_INC "gemm_hp_lab5.inc"
#include "gemm_hp_lab_kernel.inl"
#undef HP_KERNEL
#undef HP_ASM_INC
#define HP_KERNEL k_lab6
#define HP_ASM_INC "gemm_hp_lab6.inc"
#include "gemm_hp_lab_kernel.inl"
#undef HP_KERNEL
#undef HP_ASM_INC
#define HP_KERNEL k_lab7
#define HP_ASM_INC "gemm_hp_lab7.inc"
#include "gemm_hp_lab_kernel.inl"
#undef HP_KERNEL
#undef HP_ASM_INC
#define HP_KERNEL k_lab8
#define HP_ASM_INC "gemm_hp_lab8.inc"
#include "gemm_hp_lab_kernel.inl"
#undef HP_KERNEL
#undef HP_ASM_INC
#define HP_KERNEL k_lab9
#define HP_ASM_INC "gemm_hp_lab9.inc"
#include "gemm_hp_lab_kernel.inl"
#undef HP_KERNEL
#undef HP_ASM_INC
#define HP_KERNEL k_lab10
#define HP_ASM_INC "gemm_hp_lab10.inc"
#include "gemm_hp_lab_kernel.inl"
#undef HP_KERNEL
#undef HP_ASM_INC
#define HP_KERNEL k_lab11
#define HP_ASM_INC "gemm_hp_lab11.inc"
#include "gemm_hp_lab_kernel.inl"
#undef HP_KERNEL
#undef HP_ASM_INC
#undef HP_TRACE
#define HP_TRACE 2
#define HP_KERNEL k_small
#define HP_ASM_INC "../../motion324_amd/csrc/gemm_hp_gelu.inc"
#include "gemm_hp_lab_kernel.inl"
#undef HP_KERNEL
#undef HP_ASM_INC
#undef HP_TRACE
#define HP_TRACE 1
#define HP_KERNEL k_trace
#define HP_ASM_INC "gemm_hp_lab3.inc"
#include "gemm_hp_lab_kernel.inl"
#undef HP_KERNEL
#undef HP_ASM_INC
}  // namespace

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)
void m324_set_error(const char*, ...) {}
int m324::tunable(int) { return 0; }
static float bf2f(bf16_t v) { unsigned u = (unsigned)v << 16; float f; memcpy(&f, &u, 4); return f; }

typedef void (*kern_t)(const bf16_t*, long, const bf16_t*, long, bf16_t*, long, int, int, const float*, const float*, const float2*, int, int, int, unsigned*);

static void run(const char* name, kern_t k, const bf16_t* A, const bf16_t* W, bf16_t* C, const float* bias, int M, int N, int xcd, unsigned* trace, int grid_cap) {
    const int K = 768, ntn = N / 128, ntiles = ntn * ((M + 255) / 256);
    const int grid = ntiles < grid_cap ? ntiles : grid_cap;
    hipEvent_t e0, e1;
    HIP_OK(hipEventCreate(&e0));
    HIP_OK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, A, (long)K, W, (long)K, C, (long)N, M, N, bias, (const float*)nullptr, (const float2*)nullptr, ntn, ntiles, xcd, (unsigned*)nullptr);
        HIP_OK(hipEventRecord(e0));
        for (int i = 0; i < 10; ++i)
            hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, A, (long)K, W, (long)K, C, (long)N, M, N, bias, (const float*)nullptr, (const float2*)nullptr, ntn, ntiles, xcd, (unsigned*)nullptr);
        HIP_OK(hipEventRecord(e1));
        HIP_OK(hipEventSynchronize(e1));
        float ms;
        HIP_OK(hipEventElapsedTime(&ms, e0, e1));
        if (ms / 10 < best) best = ms / 10;
    }
    const double tiles_per_wg = (double)ntiles / grid;
    printf("%-64s %7.1f us  %6.0f TF/s   %.2f us per tile round\n", name, best * 1e3, 2.0 * M * N * K / (best * 1e-3) / 1e12, best * 1e3 / ceil(tiles_per_wg));
    if (trace) {
        HIP_OK(hipMemset(trace, 0, 1024 * 16));
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, A, (long)K, W, (long)K, C, (long)N, M, N, bias, (const float*)nullptr, (const float2*)nullptr, ntn, ntiles, xcd, trace);
        HIP_OK(hipDeviceSynchronize());
        std::vector<unsigned> h(1024 * 4);
        HIP_OK(hipMemcpy(h.data(), trace, 1024 * 16, hipMemcpyDeviceToHost));
        for (int b : {0, 1, 40, 255}) {
            if (b >= grid) continue;
            const unsigned* t = &h[(size_t)b * 4];
            const int nt = (ntiles - b + grid - 1) / grid;
            const unsigned total = t[2] - t[1];
            printf("  wg %3d: %d tiles, %u ticks first top .. last top (%.0f per stage), %u ticks in top waits + barriers (%.0f per stage)\n", b, nt, total,
                   (double)total / (nt * 12 - 1), t[0], (double)t[0] / (nt * 12 - 1));
        }
    }
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 10368, N = argc > 2 ? atoi(argv[2]) : 3072, K = 768;
    const int xcd = argc > 3 ? atoi(argv[3]) : 3;
    const int cap = argc > 4 ? atoi(argv[4]) : 256;
    setvbuf(stdout, nullptr, _IONBF, 0);
    bf16_t *A, *W, *C;
    float* bias;
    unsigned* trace;
    HIP_OK(hipMalloc(&A, (size_t)M * K * 2));
    HIP_OK(hipMalloc(&W, (size_t)N * K * 2));
    HIP_OK(hipMalloc(&C, (size_t)M * N * 2));
    HIP_OK(hipMalloc(&bias, (size_t)N * 4));
    HIP_OK(hipMalloc(&trace, 1024 * 16));
    HIP_OK(hipMemset(bias, 0, (size_t)N * 4));
    std::vector<bf16_t> hA((size_t)M * K), hW((size_t)N * K);
    unsigned x = 12345;
    auto fill = [&](std::vector<bf16_t>& v, int eb) {
        for (auto& e : v) { x = x * 1664525u + 1013904223u; e = (bf16_t)(((127 + eb) << 7) + ((x >> 9) & 0x7f) + ((x >> 31) << 15)); }
    };
    fill(hA, -1);
    fill(hW, -5);
    HIP_OK(hipMemcpy(A, hA.data(), (size_t)M * K * 2, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(W, hW.data(), (size_t)N * K * 2, hipMemcpyHostToDevice));
    printf("M = %d N = %d K = %d, xcd mode %d, grid cap %d\n", M, N, K, xcd, cap);
    run("v15 bias + GELU (product stream)", k_full, A, W, C, bias, M, N, xcd, nullptr, cap);
    {
        std::vector<bf16_t> hc((size_t)M * N);
        HIP_OK(hipMemcpy(hc.data(), C, (size_t)M * N * 2, hipMemcpyDeviceToHost));
        double worst = 0;
        for (int it = 0; it < 4000; ++it) {
            x = x * 1664525u + 1013904223u;
            const int m = (x >> 8) % M;
            x = x * 1664525u + 1013904223u;
            const int n = (x >> 8) % N;
            double s = 0;
            for (int k = 0; k < K; ++k) s += (double)bf2f(hA[(size_t)m * K + k]) * bf2f(hW[(size_t)n * K + k]);
            s = 0.5 * s * (1.0 + erf(s / sqrt(2.0)));
            const double d = fabs(bf2f(hc[(size_t)m * N + n]) - s) / (fabs(s) + 0.05);
            worst = d > worst ? d : worst;
        }
        printf("  check: worst rel err %.2e%s\n", worst, worst < 1e-2 ? "" : "  <-- WRONG");
    }
    run("v15 bias only", k_plain, A, W, C, bias, M, N, xcd, nullptr, cap);
    run("stamped stream (GELU)", k_trace, A, W, C, bias, M, N, xcd, trace, cap);
    run("GELU, nt stores", k_lab10, A, W, C, bias, M, N, xcd, nullptr, cap);
    run("GELU, sc1 stores", k_lab11, A, W, C, bias, M, N, xcd, nullptr, cap);
    run("GELU, every tile of a workgroup stored over the same 64 KiB", k_small, A, W, C, bias, M, N, xcd, nullptr, cap);
    run("GELU with packed fp32 arithmetic (v_pk_*)", k_lab8, A, W, C, bias, M, N, xcd, nullptr, cap);
    run("GELU packed, no stores", k_lab9, A, W, C, bias, M, N, xcd, nullptr, cap);
    run("GELU, no stores", k_lab2, A, W, C, bias, M, N, xcd, nullptr, cap);
    run("no epilogue", k_lab1, A, W, C, bias, M, N, xcd, nullptr, cap);
    run("no epilogue, no barriers", k_lab6, A, W, C, bias, M, N, xcd, nullptr, cap);
    run("no epilogue, no LDS-DMA (MFMA + fragment reads)", k_lab4, A, W, C, bias, M, N, xcd, nullptr, cap);
    run("no epilogue, no fragment reads (MFMA + LDS-DMA)", k_lab5, A, W, C, bias, M, N, xcd, nullptr, cap);
    run("MFMAs only", k_lab7, A, W, C, bias, M, N, xcd, nullptr, cap);
    return 0;
}
