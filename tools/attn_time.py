#!/usr/bin/env python3
"""Micro-benchmark / ablation of the encoder attention kernel (development aid)."""
import sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iisan_amd import _lib
if os.environ.get("IISAN_LIB"):            # an ablation build: make -C iisan_amd/csrc EXTRA=-DATTN_DEBUG_BITS, copied aside
    _lib.LIB_PATH = os.path.abspath(os.environ["IISAN_LIB"])
lib = _lib.load()
items, S, heads = 1408, 197, 12
qkv = (torch.randn(items, heads, 3, S, 64, device="cuda")).half()
ctx = torch.empty(items * S, heads * 64, device="cuda", dtype=torch.float16)
st = torch.cuda.current_stream().cuda_stream
flops = items * heads * 4.0 * S * S * 64
for name, dbg in (("full", 0), ("no Vt write", 1), ("no K write", 16), ("no store", 8), ("no LDS writes, no store", 1 + 16 + 8),
                  ("no global loads", 32), ("no loads, no store", 32 + 8), ("no query blocks (loads + staging)", 64),
                  ("no query blocks, no LDS writes", 64 + 1 + 16), ("full again", 0)):
    _lib.dev_set("attn_debug", dbg)
    for _ in range(2):
        lib.iisan_attention16(0, qkv.data_ptr(), None, ctx.data_ptr(), items, S, heads, st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        lib.iisan_attention16(0, qkv.data_ptr(), None, ctx.data_ptr(), items, S, heads, st)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print(f"{name:36s} {dt*1e6:8.1f} us  ({flops/dt/1e12:.0f} TF-equivalent)")
_lib.dev_set("attn_debug", 0)
