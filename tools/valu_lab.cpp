// VALU issue-rate lab for gfx950: how many cycles does a wave64 instruction of each kind occupy its SIMD, and do
// transcendental and packed-FMA instructions overlap?  (Decides how the attention softmax should compute exp2.)
//     hipcc --offload-arch=gfx950 -O2 tools/valu_lab.cpp -o tools/valu_lab ; run on the GPU box.
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

typedef float f2 __attribute__((ext_vector_type(2)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

constexpr int ITER = 2000, UNROLL = 8;

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, float seed) {
    float a[UNROLL];
    f2 p[UNROLL];
    h2 q[UNROLL];
    for (int i = 0; i < UNROLL; ++i) {
        a[i] = seed * (threadIdx.x + i + 1) * 1e-3f;
        p[i] = (f2){a[i], a[i] * 0.5f};
        q[i] = (h2){(_Float16)a[i], (_Float16)(a[i] * 0.5f)};
    }
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < UNROLL; ++i) {
            if (MODE == 0) a[i] = __builtin_amdgcn_exp2f(a[i]);                       // v_exp_f32
            if (MODE == 1) a[i] = __builtin_fmaf(a[i], 0.999f, 0.001f);               // v_fma_f32
            if (MODE == 2) p[i] = __builtin_elementwise_fma(p[i], (f2)(0.999f), (f2)(0.001f));   // v_pk_fma_f32
            if (MODE == 3) q[i] = __builtin_elementwise_fma(q[i], (h2)((_Float16)0.999f), (h2)((_Float16)0.001f));   // v_pk_fma_f16
            if (MODE == 4) {                                                          // exp + pk_fma interleaved
                a[i] = __builtin_amdgcn_exp2f(a[i]);
                p[i] = __builtin_elementwise_fma(p[i], (f2)(0.999f), (f2)(0.001f));
            }
            if (MODE == 5) {                                                          // exp + 2 pk_fma
                a[i] = __builtin_amdgcn_exp2f(a[i]);
                p[i] = __builtin_elementwise_fma(p[i], (f2)(0.999f), (f2)(0.001f));
                p[i] = __builtin_elementwise_fma(p[i], (f2)(0.998f), (f2)(0.002f));
            }
            if (MODE == 6) {                                                          // v_exp_f16
                _Float16 h = (_Float16)a[i];
                asm volatile("v_exp_f16 %0, %1" : "=v"(h) : "v"(h));
                a[i] = (float)h;
            }
            if (MODE == 7) {                                                          // exp + 4 pk_fma
                a[i] = __builtin_amdgcn_exp2f(a[i]);
                p[i] = __builtin_elementwise_fma(p[i], (f2)(0.999f), (f2)(0.001f));
                p[i] = __builtin_elementwise_fma(p[i], (f2)(0.998f), (f2)(0.002f));
                p[i] = __builtin_elementwise_fma(p[i], (f2)(0.997f), (f2)(0.003f));
                p[i] = __builtin_elementwise_fma(p[i], (f2)(0.996f), (f2)(0.004f));
            }
            if (MODE == 8) a[i] = __builtin_amdgcn_rcpf(a[i]);                        // v_rcp_f32
        }
    }
    float s = 0;
    for (int i = 0; i < UNROLL; ++i) s += a[i] + p[i].x + p[i].y + (float)q[i].x + (float)q[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
static void run(const char* name, int insts_per_iter, float* d) {
    hipEvent_t e0, e1;
    HIP_OK(hipEventCreate(&e0));
    HIP_OK(hipEventCreate(&e1));
    const int blocks = 256 * 4;       // 4 workgroups of 4 waves per CU: 4 waves per SIMD
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0f);
    HIP_OK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0f);
    HIP_OK(hipEventRecord(e1));
    HIP_OK(hipEventSynchronize(e1));
    float ms;
    HIP_OK(hipEventElapsedTime(&ms, e0, e1));
    // wave-instructions per SIMD = 4 waves * ITER * UNROLL * insts_per_iter
    const double winst = 4.0 * ITER * UNROLL * insts_per_iter;
    printf("%-28s %8.1f us   %6.2f ns per wave-instruction per SIMD  (= %5.1f cycles at 2.0 GHz)\n", name, ms * 1e3,
           ms * 1e6 / winst, ms * 1e6 / winst * 2.0);
}

int main() {
    float* d;
    HIP_OK(hipMalloc(&d, 256 * 4 * 256 * 4));
    run<0>("v_exp_f32", 1, d);
    run<8>("v_rcp_f32", 1, d);
    run<6>("v_exp_f16 (+2 cvt)", 3, d);
    run<1>("v_fma_f32", 1, d);
    run<2>("v_pk_fma_f32", 1, d);
    run<3>("v_pk_fma_f16", 1, d);
    run<4>("exp + 1 pk_fma (2 inst)", 2, d);
    run<5>("exp + 2 pk_fma (3 inst)", 3, d);
    run<7>("exp + 4 pk_fma (5 inst)", 5, d);
    return 0;
}
