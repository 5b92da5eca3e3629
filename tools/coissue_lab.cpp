// Co-issue lab for gfx950: do the matrix pipe and the vector ALUs of ONE SIMD run at the same time when the instructions come
// from two different waves?  A 512-thread workgroup per CU puts waves w and w + 4 on SIMD w; waves 0-3 run role A, waves 4-7
// role B (0 = exit at once, 1 = v_mfma_f32_32x32x16_bf16 stream on two accumulators, 2 = v_fma_f32 stream, 3 = v_exp_f32 stream,
// 4 = softmax-like mix: exp, add, cvt_pk).  Each wave stamps s_memtime around its loop; the table shows cycles per instruction of
// each role alone and beside the other.   hipcc --offload-arch=gfx950 -O2 tools/coissue_lab.cpp -o tools/coissue_lab
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

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
constexpr int ITER = 4000;

template <int r>
__device__ __forceinline__ void role(float seed, float* out, long long* stamp, int slot) {
    const int lane = threadIdx.x & 63;
    float acc = 0.f;
    long long t0 = 0, t1 = 0;
    if (r == 1) {
        f32x16 c0, c1;
        for (int i = 0; i < 16; ++i) c0[i] = seed, c1[i] = seed * 2;
        bf16x8 a, b;
        for (int i = 0; i < 8; ++i) a[i] = (__bf16)(seed + i), b[i] = (__bf16)(seed - i);
        t0 = __builtin_readcyclecounter();
        for (int it = 0; it < ITER; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
            }
        }
        asm volatile("s_nop 0" ::"v"(c0), "v"(c1));
        t1 = __builtin_readcyclecounter();
        for (int i = 0; i < 16; ++i) acc += c0[i] + c1[i];
    } else if (r >= 2) {
        float x[8];
        for (int i = 0; i < 8; ++i) x[i] = seed * (lane + i + 1) * 1e-3f;
        t0 = __builtin_readcyclecounter();
        for (int it = 0; it < ITER; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (r == 2) x[i] = __builtin_fmaf(x[i], 0.999f, 0.001f);
                if (r == 3) x[i] = __builtin_amdgcn_exp2f(x[i]);
                if (r == 4) {
                    const float e = __builtin_amdgcn_exp2f(x[i]);
                    acc += e;
                    unsigned pk;
                    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk) : "v"(e), "v"(acc));
                    x[i] = __builtin_fmaf(x[i], 0.5f, __uint_as_float(pk & 0x3f800000u) * 1e-9f);
                }
            }
        }
        asm volatile("s_nop 0" ::"v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]));
        t1 = __builtin_readcyclecounter();
        for (int i = 0; i < 8; ++i) acc += x[i];
    }
    out[blockIdx.x * 512 + threadIdx.x] = acc;
    if (lane == 0 && blockIdx.x == 17) stamp[slot] = t1 - t0;
}

// One wave's own stream: every MFMA followed by NF independent v_fma_f32 and NE v_exp_f32 ("fillers" in the MFMA's shadow).
template <int NF, int NE>
__global__ __launch_bounds__(512) void kmix(float seed, float* out, long long* stamp, int nwaves) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (wave >= nwaves) return;
    f32x16 c0, c1;
    for (int i = 0; i < 16; ++i) c0[i] = seed, c1[i] = seed * 2;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) a[i] = (__bf16)(seed + i), b[i] = (__bf16)(seed - i);
    float x[16], y[4];
    for (int i = 0; i < 16; ++i) x[i] = seed * (lane + i + 1) * 1e-3f;
    for (int i = 0; i < 4; ++i) y[i] = seed * (lane + i + 1) * 1e-4f;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (u & 1) c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
            else c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
#pragma unroll
            for (int f = 0; f < NF; ++f) x[f & 15] = __builtin_fmaf(x[f & 15], 0.999f, 0.001f);
#pragma unroll
            for (int e = 0; e < NE; ++e) y[e & 3] = __builtin_amdgcn_exp2f(y[e & 3]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_nop 0" ::"v"(c0), "v"(c1));
    const long long t1 = __builtin_readcyclecounter();
    float acc = 0.f;
    for (int i = 0; i < 16; ++i) acc += c0[i] + c1[i] + x[i];
    for (int i = 0; i < 4; ++i) acc += y[i];
    out[(blockIdx.x & 255) * 512 + threadIdx.x] = acc;
    if (lane == 0 && blockIdx.x == 17) stamp[wave] = t1 - t0;
}

template <int NF, int NE>
static void run_mix(float* d, long long* st) {
    for (int cfg = 0; cfg < 3; ++cfg) {       // 1, 2, 4 waves per SIMD
        const int nw = cfg == 0 ? 4 : 8, blocks = cfg == 2 ? 512 : 256;
        HIP_OK(hipMemset(st, 0, 64 * 8));
        hipEvent_t e0, e1;
        HIP_OK(hipEventCreate(&e0));
        HIP_OK(hipEventCreate(&e1));
        hipLaunchKernelGGL((kmix<NF, NE>), dim3(blocks), dim3(512), 0, 0, 1.0f, d, st, nw);
        HIP_OK(hipEventRecord(e0));
        hipLaunchKernelGGL((kmix<NF, NE>), dim3(blocks), dim3(512), 0, 0, 1.0f, d, st, nw);
        HIP_OK(hipEventRecord(e1));
        HIP_OK(hipEventSynchronize(e1));
        float ms;
        HIP_OK(hipEventElapsedTime(&ms, e0, e1));
        long long h[8];
        HIP_OK(hipMemcpy(h, st, 64, hipMemcpyDeviceToHost));
        const int wps = cfg == 0 ? 1 : (cfg == 1 ? 2 : 4);
        // SIMD cycles per MFMA = kernel cycles / (MFMAs per SIMD); the clock is read off the wave's own stamp
        const double per_wave = (double)h[0] / (ITER * 8.0);
        printf("  MFMA + %2d fma + %d exp | %d wave(s)/SIMD: %6.1f cycles per MFMA in the wave's stream = %5.1f SIMD-cycles per MFMA  (%.0f us)\n",
               NF, NE, wps, per_wave, per_wave / wps, ms * 1e3);
    }
}

// What does issuing an LDS-DMA piece (1 KiB: 64 lanes x 16 B, global memory -> LDS) cost a SIMD whose waves are streaming MFMAs?
// Per loop iteration a wave issues 8 MFMAs and NP pieces, addressed in one of three ways: MODE 0 per-lane 64-bit addresses
// (global_load_lds), MODE 1 wave-uniform base + 32-bit per-lane offset (the same instruction in its saddr form, if the compiler
// picks it), MODE 2 a buffer resource + 32-bit offset (buffer_load ... lds).  Sources: a 2 MiB window (L2 resident).
typedef __attribute__((address_space(3))) void lds_v;
typedef __attribute__((address_space(1))) const void glb_v;
template <int NP, int MODE>
__global__ __launch_bounds__(512) void kdma(float seed, const unsigned char* __restrict__ src, float* out, long long* stamp, int nwaves) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[64 * 1024];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (wave >= nwaves) return;
    f32x16 c0, c1;
    for (int i = 0; i < 16; ++i) c0[i] = seed, c1[i] = seed * 2;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) a[i] = (__bf16)(seed + i), b[i] = (__bf16)(seed - i);
    const unsigned voff = (unsigned)((lane >> 3) * 4096 + (lane & 7) * 16 + wave * 32768 + (blockIdx.x & 7) * 262144);
    const unsigned char* lanep = src + voff;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(src), 0, 1 << 30, 0x00020000);
    unsigned char* dst = smem + wave * 8192;
    uint4 ld[4] = {make_uint4(1, 2, 3, 4), make_uint4(5, 6, 7, 8), make_uint4(9, 10, 11, 12), make_uint4(13, 14, 15, 16)};
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (u & 1) c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
            else c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
            if (u < NP) {
                const unsigned step = (unsigned)((it & 31) * 128 + u * 1024 * 1024 / 8);
                if (MODE == 0) __builtin_amdgcn_global_load_lds((glb_v*)(lanep + step), (lds_v*)(dst + u * 1024), 16, 0, 0);
                if (MODE == 1) __builtin_amdgcn_global_load_lds((glb_v*)(src + step + voff), (lds_v*)(dst + u * 1024), 16, 0, 0);
                if (MODE == 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_v*)(dst + u * 1024), 16, voff, step, 0, 0);
                // plain 16-byte loads into registers and 16-byte stores, per-lane 64-bit address vs buffer resource
                if (MODE == 3) ld[u & 3] = *reinterpret_cast<const uint4*>(lanep + step);
                if (MODE == 4) ld[u & 3] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, step, 0));
                if (MODE == 5) *reinterpret_cast<uint4*>(const_cast<unsigned char*>(lanep) + step + (4 << 20)) = ld[u & 3];
                if (MODE == 6) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, ld[u & 3]), rsrc, voff + (4 << 20), step, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)\n s_nop 0" ::"v"(c0), "v"(c1));
    const long long t1 = __builtin_readcyclecounter();
    float acc = 0.f;
    for (int i = 0; i < 16; ++i) acc += c0[i] + c1[i];
    out[(blockIdx.x & 255) * 512 + threadIdx.x] = acc + smem[threadIdx.x] + (float)(ld[0].x + ld[1].y + ld[2].z + ld[3].w);
    if (lane == 0 && blockIdx.x == 17) stamp[wave] = t1 - t0;
}

template <int NP, int MODE>
static void run_dma(const char* what, const unsigned char* src, float* d, long long* st) {
    HIP_OK(hipMemset(st, 0, 64 * 8));
    hipEvent_t e0, e1;
    HIP_OK(hipEventCreate(&e0));
    HIP_OK(hipEventCreate(&e1));
    hipLaunchKernelGGL((kdma<NP, MODE>), dim3(256), dim3(512), 0, 0, 1.0f, src, d, st, 8);
    HIP_OK(hipEventRecord(e0));
    hipLaunchKernelGGL((kdma<NP, MODE>), dim3(256), dim3(512), 0, 0, 1.0f, src, d, st, 8);
    HIP_OK(hipEventRecord(e1));
    HIP_OK(hipEventSynchronize(e1));
    float ms;
    HIP_OK(hipEventElapsedTime(&ms, e0, e1));
    long long h[8];
    HIP_OK(hipMemcpy(h, st, 64, hipMemcpyDeviceToHost));
    // two waves per SIMD: SIMD cycles per 8-MFMA iteration of BOTH waves = the younger wave's total / ITER (it finishes last)
    const double older = (double)h[0] / ITER, younger = (double)h[4] / ITER;
    printf("  8 MFMAs + %d pieces, %-34s | cycles per iteration: older wave %6.1f, younger %6.1f  (%.0f us; 2 x 8 bare MFMAs = 512)\n", NP, what,
           older, younger, ms * 1e3);
}

__global__ __launch_bounds__(512) void k(int ra, int rb, float seed, float* out, long long* stamp) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = wave < 4 ? ra : rb;
    if (r == 1) role<1>(seed, out, stamp, wave);
    else if (r == 2) role<2>(seed, out, stamp, wave);
    else if (r == 3) role<3>(seed, out, stamp, wave);
    else if (r == 4) role<4>(seed, out, stamp, wave);
}

static const char* NAME[] = {"-", "mfma 32x32x16", "v_fma_f32", "v_exp_f32", "exp+add+cvt+fma"};
static const int PER_ITER[] = {1, 8, 8, 8, 32};   // role 4: exp, add, cvt_pk, and, (mul), fma per element ~ 4 plain + 1 exp; counted as 4

int main() {
    float* d;
    long long* st;
    HIP_OK(hipMalloc(&d, 256 * 512 * 4));
    HIP_OK(hipMalloc(&st, 64 * 8));
    const int pairs[][2] = {{1, 0}, {0, 1}, {2, 0}, {3, 0}, {4, 0}, {1, 1}, {1, 2}, {2, 1}, {1, 3}, {3, 1}, {1, 4}, {4, 1}, {2, 2}, {4, 4}};
    printf("%-18s %-18s | cycles per instruction: waves 0-3, waves 4-7   (kernel us)\n", "waves 0-3 (older)", "waves 4-7");
    for (auto& p : pairs) {
        HIP_OK(hipMemset(st, 0, 64 * 8));
        hipEvent_t e0, e1;
        HIP_OK(hipEventCreate(&e0));
        HIP_OK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, p[0], p[1], 1.0f, d, st);
        HIP_OK(hipEventRecord(e0));
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, p[0], p[1], 1.0f, d, st);
        HIP_OK(hipEventRecord(e1));
        HIP_OK(hipEventSynchronize(e1));
        float ms;
        HIP_OK(hipEventElapsedTime(&ms, e0, e1));
        long long h[8];
        HIP_OK(hipMemcpy(h, st, 64, hipMemcpyDeviceToHost));
        const double ca = p[0] ? (double)h[0] / (ITER * PER_ITER[p[0]]) : 0, cb = p[1] ? (double)h[4] / (ITER * PER_ITER[p[1]]) : 0;
        printf("%-18s %-18s | %7.2f  %7.2f   (%.1f us)\n", NAME[p[0]], NAME[p[1]], ca, cb, ms * 1e3);
    }
    printf("own fillers behind every MFMA (one stream per wave):\n");
    run_mix<0, 0>(d, st);
    run_mix<4, 0>(d, st);
    run_mix<8, 0>(d, st);
    run_mix<12, 0>(d, st);
    run_mix<16, 0>(d, st);
    run_mix<0, 2>(d, st);
    run_mix<0, 4>(d, st);
    run_mix<6, 2>(d, st);
    run_mix<8, 2>(d, st);
    run_mix<10, 2>(d, st);
    printf("LDS-DMA pieces beside MFMAs (8 waves = 2 per SIMD, one workgroup per CU):\n");
    unsigned char* src;
    HIP_OK(hipMalloc(&src, 16 << 20));
    HIP_OK(hipMemset(src, 1, 16 << 20));
    run_dma<0, 0>("(none)", src, d, st);
    run_dma<2, 0>("per-lane 64-bit addresses", src, d, st);
    run_dma<2, 1>("uniform base + 32-bit lane offset", src, d, st);
    run_dma<2, 2>("buffer resource + 32-bit offset", src, d, st);
    run_dma<4, 0>("per-lane 64-bit addresses", src, d, st);
    run_dma<4, 1>("uniform base + 32-bit lane offset", src, d, st);
    run_dma<4, 2>("buffer resource + 32-bit offset", src, d, st);
    printf("16-byte loads into registers / 16-byte stores beside MFMAs (the same harness; 'pieces' = instructions per 8 MFMAs):\n");
    run_dma<2, 3>("global_load_dwordx4 (64-bit addr)", src, d, st);
    run_dma<2, 4>("buffer_load_dwordx4", src, d, st);
    run_dma<2, 5>("global_store_dwordx4 (64-bit addr)", src, d, st);
    run_dma<2, 6>("buffer_store_dwordx4", src, d, st);
    run_dma<4, 5>("global_store_dwordx4 (64-bit addr)", src, d, st);
    run_dma<4, 6>("buffer_store_dwordx4", src, d, st);
    return 0;
}
