#!/usr/bin/env python3
"""Time the 128x128 fp16 MFMA GEMM with the fp32 epilogue (split-operand products) alone: shapes of Versa's dim-align."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iisan_amd import _lib  # noqa: E402

lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, M, N, K in [("align fwd 2 terms", 1408, 1024, 16384), ("align fwd 3 terms", 1408, 1024, 24576),
                      ("align dW 2 terms", 1024, 8192, 2816), ("fc fwd 3 terms", 11264, 768, 2304),
                      ("fc dW 3 terms", 768, 768, 33792)]:
    Mp = (M + 127) // 128 * 128
    A = (torch.randn(Mp, K, device="cuda") * 0.5).half()
    W = (torch.randn(N, K, device="cuda") * 0.5).half()
    C = torch.zeros(M, N, device="cuda")
    for ks in (1, 2, 4, 8, 16):
        if K // 64 // ks < 4:
            continue
        t = timeit(lambda: _lib.check(lib.iisan_gemm16_f32(A.data_ptr(), W.data_ptr(), C.data_ptr(), M, N, K, ks, st), "g16f32"))
        print(f"{name:20s} M={M} N={N} K={K} ksplit={ks:2d}: {t:8.1f} us  {2.0 * M * N * K / t / 1e6:7.1f} TF(16-bit)")
