#!/usr/bin/env python3
"""Slot timeline of workgroup 0 of gemm16_s256 (debug bit 16): per slot, work time and barrier-wait time of both groups."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iisan_amd import _lib
lib = _lib.load()
M, N, K, mode = 277376, 2304, 768, 0
A = (torch.randn(M + 256, K, device="cuda") * 0.5).half()
W = (torch.randn(N, K, device="cuda") * 0.05).half()
b = torch.randn(N, device="cuda")
out = torch.empty(M + 256, N, device="cuda", dtype=torch.float16)
st = torch.cuda.current_stream().cuda_stream
for dbg, label in [(3, "neither"), (1, "no-epilogue"), (0, "full")]:
    lib.iisan_set_gemm16_variant(3 + ((dbg | 16) << 8))
    for _ in range(2):
        lib.iisan_gemm16(0, mode, A.data_ptr(), W.data_ptr(), b.data_ptr(), out.data_ptr(), None, M, N, K, st)
        torch.cuda.synchronize()
    t = out.view(-1).view(torch.int64)[4096:4096 + 2048].cpu().view(2, 1024)
    print(f"== {label}: stamps come in pairs (before barrier, after barrier); 4 per K-step: R-end, R-barrier-exit, M-end, M-barrier-exit")
    for g in range(2):
        x = t[g, 96:96 + 4 * 26].view(-1, 4).tolist()     # K-steps 24.. (third tile)
        prev = t[g, 95].item()
        rows = []
        for r_end, r_bar, m_end, m_bar in x:
            rows.append((r_end - prev, r_bar - r_end, m_end - r_bar, m_bar - m_end))
            prev = m_bar
        print(f" group {'AB'[g]}: (R work, R wait, M work, M wait) per K-step:")
        print("   " + " ".join(f"({a},{b_},{c},{d})" for a, b_, c, d in rows))
lib.iisan_set_gemm16_variant(0)
