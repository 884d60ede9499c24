#!/usr/bin/env python3
"""Timing of the f32-core GEMM on Versa's wide skinny-K products (development aid): [1408, 64] x [64 -> N] and back."""
import sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iisan_amd import _lib
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
import itertools
for M, mode in itertools.product((1408, 11264), (0, 1)):
    _lib.dev_set("gemm32_k64", mode)
    print("k64 mode", mode)
    for name, N, K, ta, tb in [("up  [M,64]x[N,64]^T", 8192, 64, 0, 0), ("up  [M,64]x[N,64]^T", 1024, 64, 0, 0),
                               ("dF  [M,64]x[64,N]", 8192, 64, 0, 1)]:
        A = torch.randn(M, K, device="cuda")
        B = torch.randn((K, N) if tb else (N, K), device="cuda")
        C_ = torch.zeros(M, N, device="cuda")
        for _ in range(3):
            lib.iisan_gemm32(A.data_ptr(), B.data_ptr(), None, C_.data_ptr(), M, N, K, ta, tb, 0, 0, st)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        it = 200
        for _ in range(it):
            lib.iisan_gemm32(A.data_ptr(), B.data_ptr(), None, C_.data_ptr(), M, N, K, ta, tb, 0, 0, st)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / it
        by = 4.0 * (M * K + N * K + M * N)
        print(f"M={M:6d} N={N:5d} K={K:5d} {name:22s}: {dt*1e6:7.1f} us  {2.0*M*N*K/dt/1e12:6.2f} TF  {by/dt/1e9:7.0f} GB/s", flush=True)
