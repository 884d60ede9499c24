#!/usr/bin/env python3
"""Same-process A/B of the two attention kernels (dev switch attn_debug bit 5: 0 = register-prefetch kernel, two workgroups per CU, the
product; 32 = no prefetch, three workgroups per CU) on the ViT (197 tokens) and BERT (30 tokens) shapes of the bs = 128 step:
device time per launch, output equality."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iisan_amd import _lib
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
for name, S, bias in (("vit", 197, False), ("bert", 30, True)):
    items, heads = 1408, 12
    qkv = torch.randn(items, heads, 3, S, 64, device="cuda").half()
    kb = None
    if bias:
        kb = torch.zeros(items, S, device="cuda"); kb[:, 20:] = -1.0; kb[5] = -1.0
    out = {}
    for dbg in (0, 32):
        _lib.dev_set("attn_debug", dbg)
        ctx = torch.full((items * S, heads * 64), float("nan"), device="cuda", dtype=torch.float16)
        lib.iisan_attention16(0, qkv.data_ptr(), kb.data_ptr() if bias else None, ctx.data_ptr(), items, S, heads, st)
        torch.cuda.synchronize()
        out[dbg] = ctx
    d = (out[0].float() - out[32].float()).abs().max().item()
    print(f"{name}: new vs old kernel max |d| {d:.3e}, finite {torch.isfinite(out[0]).all().item()}", flush=True)
    for r in range(3):
        for dbg in (0, 32):
            _lib.dev_set("attn_debug", dbg)
            for _ in range(3): lib.iisan_attention16(0, qkv.data_ptr(), kb.data_ptr() if bias else None, ctx.data_ptr(), items, S, heads, st)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(20): lib.iisan_attention16(0, qkv.data_ptr(), kb.data_ptr() if bias else None, ctx.data_ptr(), items, S, heads, st)
            torch.cuda.synchronize()
            print(f"  round {r} {name} {'no-prefetch/3wg' if dbg else 'prefetch/2wg'}: {(time.perf_counter() - t0) / 20 * 1e6:.1f} us", flush=True)
# heads per workgroup x prefetch variant (ViT shape)
items, heads, S = 1408, 12, 197
qkv = torch.randn(items, heads, 3, S, 64, device="cuda").half()
ctx = torch.empty(items * S, heads * 64, device="cuda", dtype=torch.float16)
for r in range(2):
    for pf in (0, 32):
        row = []
        for hpw in (1, 2, 3, 4, 6):
            _lib.dev_set("attn_debug", pf | (hpw << 8))
            for _ in range(3): lib.iisan_attention16(0, qkv.data_ptr(), None, ctx.data_ptr(), items, S, heads, st)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(20): lib.iisan_attention16(0, qkv.data_ptr(), None, ctx.data_ptr(), items, S, heads, st)
            torch.cuda.synchronize()
            row.append(f"hpw {hpw}: {(time.perf_counter() - t0) / 20 * 1e6:.0f}")
        print(f"  round {r} {'no-prefetch/3wg' if pf else 'prefetch/2wg'}: " + "  ".join(row), flush=True)
_lib.dev_set("attn_debug", 0)
