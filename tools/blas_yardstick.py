"""Yardstick only (never on the product path): what does the vendor GEMM reach on this box for the shapes m324_gemm
runs?  Prints TF/s of torch.matmul (hipBLASLt / rocBLAS behind it) next to m324_gemm for the same operands."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from motion324_amd import ops

SHAPES = [("square 4096", 4096, 4096, 4096), ("square 8192", 8192, 8192, 8192), ("dec fc1", 65536, 3072, 768),
          ("dec fc2", 65536, 768, 3072), ("dec fc", 65536, 768, 768), ("trunk qkv", 10368, 2304, 768),
          ("trunk fc1", 10368, 3072, 768), ("trunk fc2", 10368, 768, 3072)]


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    best = 1e30
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters)
    return best


def main():
    dev = torch.device("cuda:0")
    for name, M, N, K in SHAPES:
        a = (torch.rand(M, K, device=dev) - 0.5).bfloat16()
        w = ((torch.rand(N, K, device=dev) - 0.5) * 0.1).bfloat16()
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        t_blas = timeit(lambda: torch.matmul(a, w.t(), out=out))
        out2 = torch.empty_like(out)
        t_ours = timeit(lambda: ops.gemm(a, w, out2))
        fl = 2.0 * M * N * K
        print(f"{name:12s} M={M:6d} N={N:5d} K={K:5d} | vendor {t_blas*1e3:8.1f} us {fl/t_blas/1e9:6.0f} TF/s | "
              f"m324 {t_ours*1e3:8.1f} us {fl/t_ours/1e9:6.0f} TF/s", flush=True)


if __name__ == "__main__":
    main()
