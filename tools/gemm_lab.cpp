// Standalone GEMM lab: times m324_gemm variants (forced through m324_set_tunable) on the shapes of the c2 clip without
// paying a Python / torch start-up on the GPU box.  Build (in the container, cross-compiles):
//     hipcc -O2 -std=c++17 tools/gemm_lab.cpp -o tools/gemm_lab -ldl
// Run (GPU box):  tools/gemm_lab [--variants 0,2,6,7] [--iters 20] [--only substr]
// Variant 0 = the library's own choice.  Every variant is checked against variant 2's output (max |diff|).
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cmath>
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

struct Shape {
    const char* name;
    int M, N, K;
    int epi;   // 0 plain bf16 out, 1 bias+gelu bf16 out, 2 fp32 residual (in place) fp32 out, 3 bias bf16 out,
               // 4 fused q|k|v heads (RMSNorm, transposed V; L = M, H = N / 192), 5 the same with bias and row-major V, L = 257
               // 6 fp32 out + residual broadcast over 2048 rows (the decoder's out-projection), 7 fp32 out, no residual
};

static unsigned short f2bf(float f) {
    unsigned u;
    memcpy(&u, &f, 4);
    u += 0x7FFF + ((u >> 16) & 1);
    return (unsigned short)(u >> 16);
}

int main(int argc, char** argv) {
    std::vector<int> variants = {0};
    int iters = 20;
    std::string only, libover;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--variants") && i + 1 < argc) {
            variants.clear();
            for (char* t = strtok(argv[++i], ","); t; t = strtok(nullptr, ",")) variants.push_back(atoi(t));
        } else if (!strcmp(argv[i], "--iters") && i + 1 < argc) iters = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--only") && i + 1 < argc) only = argv[++i];
        else if (!strcmp(argv[i], "--lib") && i + 1 < argc) libover = argv[++i];
    }
    std::string self = argv[0];
    std::string dir = self.substr(0, self.find_last_of('/') == std::string::npos ? 0 : self.find_last_of('/'));
    std::string libp = (dir.empty() ? std::string(".") : dir) + "/../motion324_amd/libm324.so";
    if (!libover.empty()) libp = libover;
    void* h = dlopen(libp.c_str(), RTLD_NOW);
    if (!h) { fprintf(stderr, "dlopen %s: %s\n", libp.c_str(), dlerror()); return 2; }
    auto gemm = (int (*)(const m324_gemm_args*, void*))dlsym(h, "m324_gemm");
    auto lasterr = (int (*)(char*, int))dlsym(h, "m324_last_error");
    auto settun = (int (*)(const char*, int))dlsym(h, "m324_set_tunable");

    const Shape shapes[] = {
        {"trunk qkv", 10368, 2304, 768, 0},   {"trunk fc+res", 10368, 768, 768, 2},  {"trunk fc1 gelu", 10368, 3072, 768, 1},
        {"trunk fc2+res", 10368, 768, 3072, 2}, {"dino qkv", 8224, 2304, 768, 3},     {"dino fc1 gelu", 8224, 3072, 768, 1},
        {"dino fc2+res", 8224, 768, 3072, 2},  {"dino fc+res", 8224, 768, 768, 2}, {"dec fc+res", 65536, 768, 768, 2}, {"dec fc+bres", 65536, 768, 768, 6}, {"dec fc f32", 65536, 768, 768, 7}, {"dec fc bf16", 65536, 768, 768, 0}, {"dec fc b16+bres", 65536, 768, 768, 8}, {"dec fc2 b16 inpl", 65536, 768, 3072, 9},    {"dec fc1 gelu", 65536, 3072, 768, 1},
        {"dec fc1 plain", 65536, 3072, 768, 0}, {"dec fc2+res", 65536, 768, 3072, 2}, {"square 4096", 4096, 4096, 4096, 0},
        {"square 8192", 8192, 8192, 8192, 0},   {"pcd fc2", 64, 768, 3072, 2},         {"pcd fc1 gelu", 64, 3072, 768, 1},
        {"pcd qkv", 64, 2304, 768, 0},          {"pcd fc", 64, 768, 768, 2},
        {"trunk qkv heads", 10368, 2304, 768, 4}, {"dino qkv heads", 8224, 2304, 768, 5},
        {"deckv proj", 2048, 1536, 768, 0}, {"decq proj", 2048, 768, 768, 0},
    };
    hipStream_t st;
    HIP_OK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    HIP_OK(hipEventCreate(&e0));
    HIP_OK(hipEventCreate(&e1));
    srand(1);
    for (const Shape& s : shapes) {
        if (!only.empty() && !strstr(s.name, only.c_str())) continue;
        const size_t nA = (size_t)s.M * s.K, nW = (size_t)s.N * s.K, nC = (size_t)s.M * s.N;
        std::vector<unsigned short> hA(nA), hW(nW);
        for (auto& v : hA) v = f2bf((rand() / (float)RAND_MAX - 0.5f) * 2.f);
        for (auto& v : hW) v = f2bf((rand() / (float)RAND_MAX - 0.5f) * 0.1f);
        std::vector<float> hb(s.N), hR;
        for (auto& v : hb) v = (rand() / (float)RAND_MAX - 0.5f);
        void *dA, *dW, *dC, *dRef, *dR0 = nullptr;
        float* db;
        const bool f32out = s.epi == 2 || s.epi == 6 || s.epi == 7;
        const size_t osz = f32out ? 4 : 2;
        HIP_OK(hipMalloc(&dA, nA * 2));
        HIP_OK(hipMalloc(&dW, nW * 2));
        HIP_OK(hipMalloc(&dC, nC * osz));
        HIP_OK(hipMalloc(&dRef, nC * osz));
        HIP_OK(hipMalloc((void**)&db, s.N * 4));
        HIP_OK(hipMemcpy(dA, hA.data(), nA * 2, hipMemcpyHostToDevice));
        HIP_OK(hipMemcpy(dW, hW.data(), nW * 2, hipMemcpyHostToDevice));
        HIP_OK(hipMemcpy(db, hb.data(), s.N * 4, hipMemcpyHostToDevice));
        if (s.epi == 2) {   // residual stream: x <- x + A.W^T in place; keep a pristine copy to restore
            hR.resize(nC);
            for (auto& v : hR) v = (rand() / (float)RAND_MAX - 0.5f);
            HIP_OK(hipMalloc(&dR0, nC * 4));
            HIP_OK(hipMemcpy(dR0, hR.data(), nC * 4, hipMemcpyHostToDevice));
        }
        m324_gemm_args a;
        memset(&a, 0, sizeof a);
        a.A = dA; a.lda = s.K; a.W = dW; a.ldw = s.K; a.C = dC; a.ldc = s.N;
        a.M = s.M; a.N = s.N; a.K = s.K; a.in_dtype = M324_BF16; a.out_dtype = f32out ? M324_F32 : M324_BF16;
        a.bias = (s.epi == 1 || s.epi == 3) ? db : nullptr;
        a.act = s.epi == 1 ? M324_ACT_GELU : M324_ACT_NONE;
        if (s.epi == 2) { a.residual = (const float*)dC; a.ldr = s.N; }
        void* dBR = nullptr;
        if (s.epi == 9) { a.residual = (const float*)dC; a.ldr = s.N; }
        if (s.epi == 6 || s.epi == 8) {
            std::vector<float> hbr((size_t)2048 * s.N);
            for (auto& v : hbr) v = (rand() / (float)RAND_MAX - 0.5f);
            HIP_OK(hipMalloc(&dBR, hbr.size() * 4));
            HIP_OK(hipMemcpy(dBR, hbr.data(), hbr.size() * 4, hipMemcpyHostToDevice));
            a.residual = (const float*)dBR; a.ldr = s.N; a.res_rows = 2048;
        }
        void *dQ = nullptr, *dK = nullptr, *dV = nullptr;
        float* dnw = nullptr;
        if (s.epi == 4 || s.epi == 5) {      // outputs go to Q / K / V; the correctness column then compares nothing (C untouched)
            const size_t third = (size_t)s.M * (s.N / 3) * 2;
            HIP_OK(hipMalloc(&dQ, third)); HIP_OK(hipMalloc(&dK, third)); HIP_OK(hipMalloc(&dV, third));
            std::vector<float> ones(64, 1.0f);
            HIP_OK(hipMalloc((void**)&dnw, 256));
            HIP_OK(hipMemcpy(dnw, ones.data(), 256, hipMemcpyHostToDevice));
            a.bias = s.epi == 5 ? db : nullptr;
            a.aux_mode = s.epi == 4 ? M324_AUX_QKV_HEADS_VT : M324_AUX_QKV_HEADS;
            a.qkv_q = dQ; a.qkv_k = dK; a.qkv_v = dV;
            a.qkv_qw = s.epi == 4 ? dnw : nullptr; a.qkv_kw = s.epi == 4 ? dnw : nullptr;
            a.qkv_eps = 1e-5f; a.qkv_qscale = 0.18f; a.qkv_H = s.N / 192;
            a.qkv_L = s.epi == 4 ? s.M : 257;
        }
        auto run = [&](int variant, void* out) {
            settun("M324_GEMM", variant);
            m324_gemm_args b = a;
            b.C = out;
            if (s.epi == 2 || s.epi == 9) b.residual = (const float*)out;
            int rc = gemm(&b, st);
            if (rc) { char buf[256]; lasterr(buf, 256); fprintf(stderr, "m324_gemm v%d: %s\n", variant, buf); exit(3); }
        };
        // reference output from v2
        if (s.epi == 2) HIP_OK(hipMemcpyAsync(dRef, dR0, nC * 4, hipMemcpyDeviceToDevice, st));
        run(2, dRef);
        HIP_OK(hipStreamSynchronize(st));
        std::vector<unsigned char> href(nC * osz), hout(nC * osz);
        HIP_OK(hipMemcpy(href.data(), dRef, nC * osz, hipMemcpyDeviceToHost));
        std::vector<double> best(variants.size(), 1e30), sum(variants.size(), 0.0);
        std::vector<double> err(variants.size(), 0.0);
        for (size_t vi = 0; vi < variants.size(); ++vi) {   // correctness first
            if (s.epi == 2) HIP_OK(hipMemcpyAsync(dC, dR0, nC * 4, hipMemcpyDeviceToDevice, st));
            else HIP_OK(hipMemsetAsync(dC, 0xFF, nC * osz, st));
            run(variants[vi], dC);
            HIP_OK(hipStreamSynchronize(st));
            HIP_OK(hipMemcpy(hout.data(), dC, nC * osz, hipMemcpyDeviceToHost));
            double mx = 0;
            for (size_t i = 0; i < nC; i += 7) {
                float x, y;
                if (osz == 4) { x = ((float*)hout.data())[i]; y = ((float*)href.data())[i]; }
                else {
                    unsigned ux = ((unsigned short*)hout.data())[i] << 16, uy = ((unsigned short*)href.data())[i] << 16;
                    memcpy(&x, &ux, 4); memcpy(&y, &uy, 4);
                }
                double d = fabs((double)x - (double)y);
                if (!(d <= mx)) mx = d;      // NaN-propagating max
            }
            err[vi] = mx;
        }
        for (int rep = 0; rep < 3; ++rep)                    // interleaved timing rounds
            for (size_t vi = 0; vi < variants.size(); ++vi) {
                for (int w = 0; w < 2; ++w) run(variants[vi], dC);
                HIP_OK(hipEventRecord(e0, st));
                for (int i = 0; i < iters; ++i) run(variants[vi], dC);
                HIP_OK(hipEventRecord(e1, st));
                HIP_OK(hipEventSynchronize(e1));
                float ms;
                HIP_OK(hipEventElapsedTime(&ms, e0, e1));
                double us = ms * 1e3 / iters;
                if (us < best[vi]) best[vi] = us;
                sum[vi] += us;
            }
        printf("%-16s M=%6d N=%5d K=%5d |", s.name, s.M, s.N, s.K);
        for (size_t vi = 0; vi < variants.size(); ++vi)
            printf(" v%d %7.1f us %6.0f TF/s (avg %6.1f, err %.3g) |", variants[vi], best[vi],
                   2.0 * s.M * s.N * s.K / best[vi] / 1e6, sum[vi] / 3, err[vi]);
        printf("\n");
        fflush(stdout);
        hipFree(dA); hipFree(dW); hipFree(dC); hipFree(dRef); hipFree(db);
        if (dR0) hipFree(dR0);
        if (dBR) hipFree(dBR);
        if (dQ) { hipFree(dQ); hipFree(dK); hipFree(dV); hipFree(dnw); }
    }
    return 0;
}
