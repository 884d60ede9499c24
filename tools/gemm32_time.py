#!/usr/bin/env python3
"""Timing of the fp32 MFMA GEMM primitive on the side-network shapes (development aid)."""
import sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iisan_amd import _lib
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
for M in (1408, 11264):
    for name, N, K, ta, tb, acc in [("down  [M,768]x[64,768]^T", 64, 768, 0, 0, 0), ("up    [M,64]x[768,64]^T", 768, 64, 0, 0, 0),
                                    ("dX    [M,768]x[768,64]", 64, 768, 0, 1, 0), ("dF    [M,64]x[64,768]", 768, 64, 0, 1, 0), ("dWd   [64,M]x[M,768] (+=)", 768, M, 1, 1, 1)]:
        if name.startswith("dWd"):
            Mm, Nn, Kk = 64, 768, M
        else:
            Mm, Nn, Kk = M, N, K
        A = torch.randn((Kk, Mm) if ta else (Mm, Kk), device="cuda")
        B = torch.randn((Kk, Nn) if tb else (Nn, Kk), device="cuda")
        C_ = torch.zeros(Mm, Nn, device="cuda")
        for _ in range(3):
            lib.iisan_gemm32(A.data_ptr(), B.data_ptr(), None, C_.data_ptr(), Mm, Nn, Kk, ta, tb, 0, acc, st)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        it = 200
        for _ in range(it):
            lib.iisan_gemm32(A.data_ptr(), B.data_ptr(), None, C_.data_ptr(), Mm, Nn, Kk, ta, tb, 0, acc, st)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / it
        fl = 2.0 * Mm * Nn * Kk
        by = 4.0 * (Mm * Kk + Nn * Kk + Mm * Nn)
        print(f"M={M:6d} {name:28s}: {dt*1e6:7.1f} us  {fl/dt/1e12:6.2f} TF  {by/dt/1e9:7.0f} GB/s")
