#!/usr/bin/env python3
"""Time the split-operand GEMM (csrc/split.hip) against the f32-matrix-core GEMM (gemm32.hip) on the side network's
large products.  Usage on the GPU box: python tools/x3_time.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iisan_amd import _lib  # noqa: E402

lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream

CASES = [  # name, M, N, K, ta, tb, acc
    ("fc fwd  [4373,768]x[768,768]", 4373, 768, 768, 0, 0, 0),
    ("fc dX   [4373,768]x[768,768]", 4373, 768, 768, 0, 1, 0),
    ("fc dW   [768,4373]x[4373,768]", 768, 768, 4373, 1, 1, 1),
    ("fc fwd  [2816,768]x[768,768]", 2816, 768, 768, 0, 0, 0),
    ("fc dX   [2816,768]x[768,768]", 2816, 768, 768, 0, 1, 0),
    ("fc dW   [768,2816]x[2816,768]", 768, 768, 2816, 1, 1, 1),
    ("fc fwd  [11264,768]x[768,768]", 11264, 768, 768, 0, 0, 0),
    ("fc dX   [11264,768]x[768,768]", 11264, 768, 768, 0, 1, 0),
    ("fc dW   [768,11264]x[11264,768]", 768, 768, 11264, 1, 1, 1),
    ("fc fwd  bs=128 [1408,768]x[768,768]", 1408, 768, 768, 0, 0, 0),
    ("fc dW   bs=128", 768, 768, 1408, 1, 1, 1),
    ("align fwd [1408,8192]->1024", 1408, 1024, 8192, 0, 0, 0),
    ("align dW  [1024,1408]x[1408,8192]", 1024, 8192, 1408, 1, 1, 1),
    ("align fwd bs=1024 [11264,8192]->1024", 11264, 1024, 8192, 0, 0, 0),
]


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
    return e0.elapsed_time(e1) / n * 1e3      # us


for name, M, N, K, ta, tb, acc in CASES:
    A = torch.randn((K, M) if ta else (M, K), device="cuda") * 0.5
    B = torch.randn((K, N) if tb else (N, K), device="cuda") * 0.03
    C = torch.zeros(M, N, device="cuda")
    ws = torch.empty(lib.iisan_gemm_x3_ws_bytes(M, N, K), dtype=torch.uint8, device="cuda")
    t3 = timeit(lambda: _lib.check(lib.iisan_gemm_x3(A.data_ptr(), B.data_ptr(), None, C.data_ptr(), M, N, K, ta, tb, acc,
                                                     ws.data_ptr(), ws.numel(), st), "x3"))
    t32 = timeit(lambda: _lib.check(lib.iisan_gemm32(A.data_ptr(), B.data_ptr(), None, C.data_ptr(), M, N, K, ta, tb, 0, acc, st), "g32"))
    fl = 2.0 * M * N * K
    print(f"{name:42s} x3 {t3:8.1f} us ({fl / t3 / 1e6:7.1f} TF fp32-equiv)   gemm32 {t32:8.1f} us ({fl / t32 / 1e6:6.1f} TF)   x{t32 / t3:.2f}")
