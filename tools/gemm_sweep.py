#!/usr/bin/env python3
"""K-sweep of the forward GEMM at M=131072: separates per-tile fixed cost (prologue / epilogue) from k-loop rate."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cm3p_amd import kernels as K
from tools.bench_kernels import timeit
DEV = "cuda"
g = torch.Generator(device=DEV).manual_seed(0)
for M, N, Kd in ((131072, 2304, 768), (131072, 2304, 1536), (131072, 2304, 3072), (65536, 2304, 6144), (131072, 768, 768), (131072, 768, 3072)):
    x = torch.randn(M, Kd, device=DEV, generator=g).to(torch.bfloat16)
    w = (torch.randn(N, Kd, device=DEV, generator=g) * 0.02).to(torch.bfloat16)
    ms = timeit(lambda: K.linear_fwd(x, w), 10)
    print(f"fwd bf16 [{M}x{N}x{Kd}] {ms:7.3f} ms {2.0*M*N*Kd/ms/1e9:7.1f} TF/s")
