// Placement lab: which workgroups of a 2-per-CU grid (256 threads, 80 KiB of LDS each) share a CU, and in which order do they start?
//     hipcc --offload-arch=gfx950 -O2 tools/place_lab.cpp -o tools/place_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

__global__ __launch_bounds__(256, 2) void k(unsigned* out, long long* t, int spin) {
    __shared__ unsigned char smem[80 * 1024];
    const long long t0 = clock64();
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    smem[threadIdx.x] = (unsigned char)hw;
    __syncthreads();
    for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(64);
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = hw;
        out[2 * blockIdx.x + 1] = xcc + (smem[5] & 0);
        t[blockIdx.x] = t0;
    }
}

int main(int argc, char** argv) {
    const int grid = argc > 1 ? atoi(argv[1]) : 512;
    unsigned* out;
    long long* t;
    HIP_OK(hipMalloc(&out, grid * 8));
    HIP_OK(hipMalloc(&t, grid * 8));
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, out, t, 200);
        HIP_OK(hipDeviceSynchronize());
    }
    std::vector<unsigned> h(2 * grid);
    std::vector<long long> ht(grid);
    HIP_OK(hipMemcpy(h.data(), out, grid * 8, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(ht.data(), t, grid * 8, hipMemcpyDeviceToHost));
    std::map<unsigned long long, std::vector<int>> cu;
    long long tmin = ht[0];
    for (int b = 0; b < grid; ++b) tmin = ht[b] < tmin ? ht[b] : tmin;
    for (int b = 0; b < grid; ++b) {
        const unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 0xf;
        const unsigned cuid = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        cu[((unsigned long long)xcc << 32) | (se << 8) | (sh << 4) | cuid].push_back(b);
        if (b < 24 || (b >= 256 && b < 264))
            printf("wg %3d: hw_id %08x  xcc %u se %u sh %u cu %u simd %u wave %u tg %u  start +%lld\n", b, hw, xcc, se, sh, cuid, (hw >> 4) & 3, hw & 15,
                   (hw >> 16) & 15, ht[b] - tmin);
    }
    printf("%zu distinct (xcc, se, sh, cu)\n", cu.size());
    int shown = 0, same_b256 = 0, same_b8 = 0, pairs = 0;
    for (auto& kv : cu) {
        auto& v = kv.second;
        if (shown++ < 16) {
            printf("cu %llx:", kv.first);
            for (int b : v) printf(" %d", b);
            printf("\n");
        }
        if (v.size() == 2) {
            ++pairs;
            const int d = abs(v[1] - v[0]);
            same_b256 += d == grid / 2;
            same_b8 += d == 8;
        }
    }
    printf("CUs with exactly two workgroups: %d; partner = b + grid/2: %d; partner = b + 8: %d\n", pairs, same_b256, same_b8);
    return 0;
}
