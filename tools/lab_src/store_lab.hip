// What does a global store cost the wave that issues it, alone and between MFMAs?  (round 5: schedule v15's stores do not hide)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/lab_src/store_lab.hip -o tools/store_lab
// One workgroup per CU (256 CUs), NW waves; every wave writes its own contiguous region of `bytes_per_wave`, 1 KiB per store instruction
// (64 lanes x 16 B).  MODE 0: stores back to back.  MODE k > 0: k MFMAs (32x32x16 bf16, independent accumulators) between stores.
// WHO: 0 = every wave stores; 1 = only wave 0 of the workgroup stores (the others run the same MFMAs).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int FLAVOUR>
__device__ __forceinline__ void store16(i32x4 v, unsigned off, __amdgpu_buffer_rsrc_t r) {
    if (FLAVOUR == 0) __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, 0);
    else if (FLAVOUR == 1) __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, 2);        // nt (slc)
    else if (FLAVOUR == 2) __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, 16);       // sc1
    else __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, 1);                          // sc0
}

// layout LINES = 1: a store = 8 rows x 128 B of a [rows][ld] matrix (ld = 6144 B); 0: 1 KiB contiguous
template <int MODE, int FLAVOUR, int WHO, int LINES>
__global__ __launch_bounds__(256, 1) void k(char* out, long bytes_per_wg, int nstores, long long* cyc) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    char* base = out + (long)blockIdx.x * bytes_per_wg;
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7fffffff, 0x00020000);
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x16)(0.f);
    bf16x8 a = (bf16x8)(short)(0x3f80 + lane), b = (bf16x8)(short)(0x3f80 + wave);
    i32x4 v = {lane, wave, 3, 4};
    const bool me = WHO == 0 || (WHO == 1 && wave == 0);
    const long t0 = clock64();
    for (int s = 0; s < nstores; ++s) {
        if (MODE > 0) {
#pragma unroll
            for (int m = 0; m < MODE; ++m) acc[m & 7] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m & 7], 0, 0, 0);
        }
        if (me) {
            unsigned off;
            if (LINES) off = (unsigned)(((s * 4 + wave) * 8 + (lane >> 3)) * 6144 + (lane & 7) * 16);
            else off = (unsigned)((s * 4 + wave) * 1024 + lane * 16);
            store16<FLAVOUR>(v, off, r);
        }
    }
    const long t1 = clock64();
    asm volatile("s_waitcnt vmcnt(0)");
    const long t2 = clock64();
    float sum = 0;
    for (int i = 0; i < 8; ++i) sum += acc[i][0];
    if (sum == 1.2345f) out[0] = 1;
    if (lane == 0) { cyc[(blockIdx.x * 4 + wave) * 2] = t1 - t0; cyc[(blockIdx.x * 4 + wave) * 2 + 1] = t2 - t0; }
}

template <int MODE, int FLAVOUR, int WHO, int LINES>
static void run(const char* name, char* out, long long* cyc, int nstores) {
    const long bytes_per_wg = LINES ? (long)nstores * 4 * 8 * 6144 : (long)nstores * 4 * 1024;
    hipEvent_t e0, e1;
    HIP_OK(hipEventCreate(&e0));
    HIP_OK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        HIP_OK(hipEventRecord(e0));
        hipLaunchKernelGGL((k<MODE, FLAVOUR, WHO, LINES>), dim3(256), dim3(256), 0, 0, out, bytes_per_wg, nstores, cyc);
        HIP_OK(hipEventRecord(e1));
        HIP_OK(hipEventSynchronize(e1));
        float ms;
        HIP_OK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    std::vector<long long> h(256 * 4 * 2);
    HIP_OK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
    double issue = 0, done = 0;
    for (int i = 0; i < 256 * 4; ++i) { issue += h[2 * i]; done += h[2 * i + 1]; }
    issue /= 1024; done /= 1024;
    const int storing = WHO == 0 ? 4 : (WHO == 1 ? 1 : 0);
    const double bytes_cu = (double)nstores * storing * 1024;
    printf("%-58s %8.1f us | loop %8.0f ticks (%6.1f per iteration), drained %8.0f | %5.1f B/tick/CU issued, %6.1f GB/s chip\n", name, best * 1e3, issue,
           issue / nstores, done, bytes_cu / issue, bytes_cu * 256 / (best * 1e-3) / 1e9);
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    char* out;
    long long* cyc;
    const int NS = 64;
    HIP_OK(hipMalloc(&out, 256l * NS * 4 * 8 * 6144 + 4096));
    HIP_OK(hipMalloc(&cyc, 256 * 4 * 2 * 8));
        run<8, 0, 1, 1>("warm-up", out, cyc, NS);
    run<0, 0, 0, 0>("stores only, 4 waves, 1 KiB contiguous", out, cyc, NS);
    run<0, 0, 0, 1>("stores only, 4 waves, 8 rows x 128 B", out, cyc, NS);
    run<0, 1, 0, 1>("stores only, 4 waves, rows, nt", out, cyc, NS);
    run<0, 2, 0, 1>("stores only, 4 waves, rows, sc1", out, cyc, NS);
    run<0, 3, 0, 1>("stores only, 4 waves, rows, sc0", out, cyc, NS);
    run<0, 0, 1, 1>("stores only, 1 wave of 4, rows", out, cyc, NS);
    run<8, 0, 2, 1>("8 MFMAs, no store", out, cyc, NS);
    run<32, 0, 2, 1>("32 MFMAs, no store", out, cyc, NS);
    run<8, 0, 0, 1>("8 MFMAs + 1 store, every wave", out, cyc, NS);
    run<8, 0, 1, 1>("8 MFMAs + 1 store, wave 0 only", out, cyc, NS);
    run<16, 0, 0, 1>("16 MFMAs + 1 store, every wave", out, cyc, NS);
    run<16, 0, 1, 1>("16 MFMAs + 1 store, wave 0 only", out, cyc, NS);
    run<32, 0, 0, 1>("32 MFMAs + 1 store, every wave", out, cyc, NS);
    run<32, 0, 1, 1>("32 MFMAs + 1 store, wave 0 only", out, cyc, NS);
    run<32, 1, 0, 1>("32 MFMAs + 1 store nt, every wave", out, cyc, NS);
    run<32, 2, 0, 1>("32 MFMAs + 1 store sc1, every wave", out, cyc, NS);
    run<32, 0, 0, 0>("32 MFMAs + 1 store (1 KiB contiguous), every wave", out, cyc, NS);
    return 0;
}
