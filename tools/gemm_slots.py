#!/usr/bin/env python3
"""Slot timeline of workgroup 0 of gemm16_s256: per slot, work time and barrier-wait time of both groups.
Needs the library built with -DS256_TIMELINE (make -C iisan_amd/csrc EXTRA=-DS256_TIMELINE); run with debug bit 16."""
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
CASES = [(1, "no-epilogue"), (0, "full")] if len(sys.argv) < 2 else [(int(a), f"debug {a}") for a in sys.argv[1:]]
for dbg, label in CASES:
    _lib.dev_set("gemm16_variant", 3 + ((dbg | 16) << 8))
    for _ in range(2):
        lib.iisan_gemm16(0, mode, A.data_ptr(), W.data_ptr(), b.data_ptr(), out.data_ptr(), None, M, N, K, st)
        torch.cuda.synchronize()
    t = out.view(-1).view(torch.int32)[8192:8192 + 2048].cpu().view(2, 1024).long()
    print(f"== {label}: per K-step (R work, barrier wait, M work, rest of slot incl. epilogue + barrier)")
    for g in range(2):
        x = t[g, 3 * 24:3 * 24 + 3 * 26].view(-1, 3).tolist()     # K-steps 24.. (third tile): stamps R-end, after barrier, M-end
        prev_m_end = t[g, 3 * 24 - 1].item()
        rows = []
        for r_end, r_bar, m_end in x:
            rows.append((r_end - prev_m_end, r_bar - r_end, m_end - r_bar))
            prev_m_end = m_end
        print(f" group {'AB'[g]}: (time from previous M end to this R end, wait at R barrier, M work):")
        print("   " + " ".join(f"({a},{b_},{c})" for a, b_, c in rows))
_lib.dev_set("gemm16_variant", 0)
