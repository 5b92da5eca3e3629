// ds_read_b64_tr_b16 semantics lab (gfx950): LDS holds bf16-sized words whose value is their own index; every lane
// supplies a byte address, the four 16-bit results per lane are printed for a few address patterns.
//     hipcc --offload-arch=gfx950 -O2 tools/tr_lab.cpp -o tools/tr_lab
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

__global__ void k(const int* addr, unsigned short* out) {
    __shared__ unsigned short lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const unsigned a = (unsigned)(size_t)lds + (unsigned)addr[threadIdx.x];   // LDS byte address
    uint2 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(a) : "memory");
    out[threadIdx.x * 4 + 0] = r.x & 0xFFFF;
    out[threadIdx.x * 4 + 1] = r.x >> 16;
    out[threadIdx.x * 4 + 2] = r.y & 0xFFFF;
    out[threadIdx.x * 4 + 3] = r.y >> 16;
}

static void run(const char* name, int (*f)(int)) {
    int h[64];
    for (int l = 0; l < 64; ++l) h[l] = f(l);
    int* d;
    unsigned short* o;
    HIP_OK(hipMalloc(&d, sizeof h));
    HIP_OK(hipMalloc(&o, 64 * 4 * 2));
    HIP_OK(hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o);
    unsigned short r[256];
    HIP_OK(hipMemcpy(r, o, sizeof r, hipMemcpyDeviceToHost));
    printf("== %s   (lane: byte address -> 4 element indices)\n", name);
    for (int l = 0; l < 64; ++l)
        printf("  l%02d a=%4d -> %4d %4d %4d %4d%s", l, h[l], r[l * 4], r[l * 4 + 1], r[l * 4 + 2], r[l * 4 + 3], (l % 4 == 3) ? "\n" : " |");
}

int main() {
    run("lane * 8 (each lane its own 8 contiguous bytes)", [](int l) { return l * 8; });
    run("row pitch 128 B: lane i of a 16-group -> row i>>2, piece i&3; groups 64 B apart",
        [](int l) { return ((l & 15) >> 2) * 128 + (l & 3) * 8 + (l >> 4) * 32; });
    run("all lanes address 0", [](int) { return 0; });
    run("row pitch 512 B, groups take rows 0-3, 4-7, 8-11, 12-15",
        [](int l) { return (((l & 15) >> 2) + (l >> 4) * 4) * 512 + (l & 3) * 8; });
    return 0;
}
