#!/usr/bin/env python3
"""Does the 4/3/3/3 split of ViT's 13 query blocks over a workgroup's four waves cost its critical path or only its throughput?
Time the attention kernel at S = 176 (11 blocks: 3/3/3/2), 192 (12: 3/3/3/3), 197 (13: 4/3/3/3), 208 (13, no pad keys): if a launch
scales with max-blocks-per-wave the barrier per head is the bound (a balanced split would pay); if it scales with total blocks it is
throughput.   python tools/attn_balance.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iisan_amd import _lib
lib = _lib.load()
items, heads = 1408, 12
st = torch.cuda.current_stream().cuda_stream
res = {}
for rnd in range(3):
    for S in (176, 192, 197, 208):
        qkv = torch.randn(items, heads, 3, S, 64, device="cuda").half()
        ctx = torch.empty(items * S, heads * 64, device="cuda", dtype=torch.float16)
        for _ in range(2):
            lib.iisan_attention16(0, qkv.data_ptr(), None, ctx.data_ptr(), items, S, heads, st)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            lib.iisan_attention16(0, qkv.data_ptr(), None, ctx.data_ptr(), items, S, heads, st)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10 * 1e6
        nb = (S + 15) // 16
        print(f"round {rnd} S {S:4d}: {dt:7.1f} us  blocks {nb} (max per wave {(nb + 3) // 4})  us per block-row {dt / nb:6.2f}  us per max-block {dt / ((nb + 3) // 4):6.2f}  bytes {items * heads * S * 64 * 2 * 4 / 1e6:.0f} MB", flush=True)
