#!/usr/bin/env python3
"""Micro-benchmark of the 16-bit GEMM variants on the encoder shapes (development aid)."""
import sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iisan_amd import _lib
lib = _lib.load()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 277376
shapes = [("qkv", 2304, 768, 0), ("o", 768, 768, 0), ("fc1", 3072, 768, 1), ("fc2", 768, 3072, 0)]
st = torch.cuda.current_stream().cuda_stream
Mp = (M + 255) // 256 * 256
for name, N, K, mode in shapes:
    A = (torch.randn(Mp, K, device="cuda") * 0.5).half()
    W = (torch.randn(N, K, device="cuda") * 0.05).half()
    b = torch.randn(N, device="cuda")
    out = torch.empty(Mp, N, device="cuda", dtype=torch.float32 if mode == 2 else torch.float16)
    res = {}
    XV = [3 + ((8 + (u << 8)) << 8) for u in (2, 4, 5, 6, 8)]
    for var in [1, 2, 3, 3 + 256, 3 + 1024] + XV:
        _lib.dev_set("gemm16_variant", var)
        for _ in range(2):
            lib.iisan_gemm16(0, mode, A.data_ptr(), W.data_ptr(), b.data_ptr(), out.data_ptr(), out.data_ptr() if mode == 2 else None, M, N, K, st)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        it = 10
        for _ in range(it):
            lib.iisan_gemm16(0, mode, A.data_ptr(), W.data_ptr(), b.data_ptr(), out.data_ptr(), out.data_ptr() if mode == 2 else None, M, N, K, st)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / it
        res[var] = (dt * 1e3, 2.0 * M * N * K / dt / 1e12)
    print(f"{name:4s} M={M} N={N} K={K}: v1 {res[1][0]:.3f} ms {res[1][1]:.0f} TF | p256 {res[2][0]:.3f} ms {res[2][1]:.0f} TF | s256 {res[3][0]:.3f} ms {res[3][1]:.0f} TF | s256 no-epilogue {res[259][1]:.0f} TF | s256 desync {res[1027][0]:.3f} ms {res[1027][1]:.0f} | xcd-phase " + " ".join(f"{res[v][1]:.0f}" for v in XV))
_lib.dev_set("gemm16_variant", 0)
