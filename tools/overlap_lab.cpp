// Do two m324_gemm launches from two HIP streams share the chip?  A 256 x 256-tile GEMM whose grid is 1.5 rounds of the
// 256 CUs leaves half of them idle in its second round; if the queues overlap, a second stream's launch fills them.
//     hipcc -O2 -std=c++17 tools/overlap_lab.cpp -o tools/overlap_lab -ldl
// Run (GPU box):  tools/overlap_lab [--lib path] [--M 8224] [--N 3072] [--K 768] [--iters 200]
// Prints us per launch: one stream back-to-back, and the same number of launches dealt over two streams.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../include/m324.h"

#define HIP_OK(x)                                                                       \
    do {                                                                                \
        hipError_t e_ = (x);                                                            \
        if (e_ != hipSuccess) {                                                         \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));   \
            exit(2);                                                                    \
        }                                                                               \
    } while (0)

static unsigned short f2bf(float f) {
    unsigned u;
    memcpy(&u, &f, 4);
    u += 0x7FFF + ((u >> 16) & 1);
    return (unsigned short)(u >> 16);
}

int main(int argc, char** argv) {
    int M = 8224, N = 3072, K = 768, iters = 200, variant = 0, f32res = 0;
    std::string libp = "motion324_amd/libm324.so";
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--M") && i + 1 < argc) M = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--N") && i + 1 < argc) N = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--K") && i + 1 < argc) K = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--iters") && i + 1 < argc) iters = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--lib") && i + 1 < argc) libp = argv[++i];
        else if (!strcmp(argv[i], "--variant") && i + 1 < argc) variant = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--res")) f32res = 1;          // fp32 output with in-place residual (out-projection / fc2)
    }
    void* h = dlopen(libp.c_str(), RTLD_NOW);
    if (!h) { fprintf(stderr, "dlopen %s: %s\n", libp.c_str(), dlerror()); return 2; }
    auto gemm = (int (*)(const m324_gemm_args*, void*))dlsym(h, "m324_gemm");
    auto settun = (int (*)(const char*, int))dlsym(h, "m324_set_tunable");
    if (variant) settun("M324_GEMM", variant);
    hipStream_t st[2];
    hipEvent_t e0, e1, ej;
    for (auto& s : st) HIP_OK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1)); HIP_OK(hipEventCreate(&ej));
    m324_gemm_args a[2];
    for (int s = 0; s < 2; ++s) {
        void *dA, *dW, *dC;
        HIP_OK(hipMalloc(&dA, (size_t)M * K * 2)); HIP_OK(hipMalloc(&dW, (size_t)N * K * 2)); HIP_OK(hipMalloc(&dC, (size_t)M * N * 4)); HIP_OK(hipMemset(dC, 0, (size_t)M * N * 4));
        // random operands: constant data toggles fewer wires, and the power-limited chip then clocks ~25 % higher
        std::vector<unsigned short> hA((size_t)M * K), hW((size_t)N * K);
        for (auto& v : hA) v = f2bf((rand() / (float)RAND_MAX - 0.5f) * 2.f);
        for (auto& v : hW) v = f2bf((rand() / (float)RAND_MAX - 0.5f) * 0.1f);
        HIP_OK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice));
        HIP_OK(hipMemcpy(dW, hW.data(), hW.size() * 2, hipMemcpyHostToDevice));
        memset(&a[s], 0, sizeof a[s]);
        a[s].A = dA; a[s].lda = K; a[s].W = dW; a[s].ldw = K; a[s].C = dC; a[s].ldc = N;
        a[s].M = M; a[s].N = N; a[s].K = K; a[s].in_dtype = M324_BF16; a[s].out_dtype = f32res ? M324_F32 : M324_BF16;
        if (f32res) { a[s].residual = (const float*)dC; a[s].ldr = N; }
    }
    auto serial = [&]() {
        HIP_OK(hipEventRecord(e0, st[0]));
        for (int i = 0; i < 2 * iters; ++i) gemm(&a[i & 1], st[0]);
        HIP_OK(hipEventRecord(e1, st[0]));
        HIP_OK(hipEventSynchronize(e1));
        float ms; HIP_OK(hipEventElapsedTime(&ms, e0, e1));
        return ms * 1e3 / (2 * iters);
    };
    auto dual = [&]() {
        HIP_OK(hipEventRecord(e0, st[0]));
        HIP_OK(hipStreamWaitEvent(st[1], e0, 0));
        for (int i = 0; i < iters; ++i) { gemm(&a[0], st[0]); gemm(&a[1], st[1]); }
        HIP_OK(hipEventRecord(ej, st[1]));
        HIP_OK(hipStreamWaitEvent(st[0], ej, 0));
        HIP_OK(hipEventRecord(e1, st[0]));
        HIP_OK(hipEventSynchronize(e1));
        float ms; HIP_OK(hipEventElapsedTime(&ms, e0, e1));
        return ms * 1e3 / (2 * iters);
    };
    // the same two-stream pattern captured into a hipGraph (fork / join by events) and replayed
    hipGraph_t graph;
    hipGraphExec_t exec;
    HIP_OK(hipStreamBeginCapture(st[0], hipStreamCaptureModeGlobal));
    HIP_OK(hipEventRecord(e0, st[0]));
    HIP_OK(hipStreamWaitEvent(st[1], e0, 0));
    for (int i = 0; i < iters; ++i) { gemm(&a[0], st[0]); gemm(&a[1], st[1]); }
    HIP_OK(hipEventRecord(ej, st[1]));
    HIP_OK(hipStreamWaitEvent(st[0], ej, 0));
    HIP_OK(hipStreamEndCapture(st[0], &graph));
    HIP_OK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    hipGraph_t graph1;
    hipGraphExec_t exec1;
    HIP_OK(hipStreamBeginCapture(st[0], hipStreamCaptureModeGlobal));
    for (int i = 0; i < 2 * iters; ++i) gemm(&a[i & 1], st[0]);
    HIP_OK(hipStreamEndCapture(st[0], &graph1));
    HIP_OK(hipGraphInstantiate(&exec1, graph1, nullptr, nullptr, 0));
    hipEvent_t g0, g1;
    HIP_OK(hipEventCreate(&g0)); HIP_OK(hipEventCreate(&g1));
    auto replay = [&](hipGraphExec_t ex) {
        HIP_OK(hipGraphLaunch(ex, st[0]));
        HIP_OK(hipStreamSynchronize(st[0]));
        HIP_OK(hipEventRecord(g0, st[0]));
        HIP_OK(hipGraphLaunch(ex, st[0]));
        HIP_OK(hipEventRecord(g1, st[0]));
        HIP_OK(hipEventSynchronize(g1));
        float ms; HIP_OK(hipEventElapsedTime(&ms, g0, g1));
        return ms * 1e3 / (2 * iters);
    };
    serial(); dual();
    printf("graph replay: serial chain %.2f us/launch   two branches %.2f us/launch\n", replay(exec1), replay(exec));
    for (int r = 0; r < 3; ++r) printf("v%d M=%d N=%d K=%d  one stream %.2f us/launch   two streams %.2f us/launch\n", variant, M, N, K, serial(), dual());
    return 0;
}
