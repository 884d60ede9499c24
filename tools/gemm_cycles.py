#!/usr/bin/env python3
"""Cycles per K-step of gemm16_s256 workgroups (debug bit 16 makes each workgroup write its cycle count into out[])."""
import sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iisan_amd import _lib
lib = _lib.load()
M = 277376
st = torch.cuda.current_stream().cuda_stream
for name, N, K, mode in [("qkv", 2304, 768, 0), ("fc2", 768, 3072, 0)]:
    A = (torch.randn(M + 256, K, device="cuda") * 0.5).half()
    W = (torch.randn(N, K, device="cuda") * 0.05).half()
    b = torch.randn(N, device="cuda")
    out = torch.empty(M + 256, N, device="cuda", dtype=torch.float16)
    for dbg, label in [(0, "full"), (1, "no-epilogue")]:
        _lib.dev_set("gemm16_variant", 3 + ((dbg | 16) << 8))
        for _ in range(3):
            t0 = time.perf_counter()
            lib.iisan_gemm16(0, mode, A.data_ptr(), W.data_ptr(), b.data_ptr(), out.data_ptr(), None, M, N, K, st)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        c = out.view(-1).view(torch.int64)[:512].cpu().view(256, 2)
        cyc, steps = c[:, 0].double(), c[:, 1].double()
        print(f"{name} {label:12s}: {dt*1e3:.3f} ms  cycles/WG {cyc.mean():.0f} (max {cyc.max():.0f})  K-steps/WG {steps.mean():.0f}  cycles/K-step {(cyc/steps).mean():.0f}  -> clock {cyc.max()/dt/1e9:.2f} GHz")
_lib.dev_set("gemm16_variant", 0)
