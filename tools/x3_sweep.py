#!/usr/bin/env python3
"""Same-box sweep of the split-operand GEMM's tile width and K-split count (dev knobs x3_force_bn / x3_force_ks) on the Cached fc products.
usage (GPU box): python tools/x3_sweep.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iisan_amd import _lib  # noqa: E402

lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
CASES = [("fc fwd  [11264,768]x[768,768]", 11264, 768, 768, 0, 0, 0), ("fc dX   [11264,768]x[768,768]", 11264, 768, 768, 0, 1, 0),
         ("fc dW   [768,11264]x[11264,768]", 768, 768, 11264, 1, 1, 1), ("align fwd [1408,8192]->1024", 1408, 1024, 8192, 0, 0, 0),
         ("align dW  [1024,1408]x[1408,8192]", 1024, 8192, 1408, 1, 1, 1)]


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


for name, M, N, K, ta, tb, acc in CASES:
    A = torch.randn((K, M) if ta else (M, K), device="cuda") * 0.5
    B = torch.randn((K, N) if tb else (N, K), device="cuda") * 0.03
    C = torch.zeros(M, N, device="cuda")
    for bn in (0, 128, 192):
        for ks in (0, 1, 2, 3, 4, 5, 7, 8, 11, 14, 16):
            if bn == 192 and N % 192:
                continue
            with _lib.dev(x3_force_bn=bn, x3_force_ks=ks):
                ws = torch.empty(lib.iisan_gemm_x3_ws_bytes(M, N, K), dtype=torch.uint8, device="cuda")
                t = timeit(lambda: _lib.check(lib.iisan_gemm_x3(A.data_ptr(), B.data_ptr(), None, C.data_ptr(), M, N, K, ta, tb, acc,
                                                                ws.data_ptr(), ws.numel(), st), "x3"))
            print(f"{name:36s} bn {bn or 'auto':>4} ks {ks or 'auto':>4}: {t:7.1f} us (amax + split + product + reduce)")
