#!/usr/bin/env python3
"""Time iisan_attention16 at the ViT / BERT shapes of the bs = 128 headline (product build; development aid):  python tools/attn_quick.py [reps]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iisan_amd import _lib
if os.environ.get("IISAN_LIB"):            # same-box A/B of two builds: IISAN_LIB=tools/lib_prev.so
    _lib.LIB_PATH = os.path.abspath(os.environ["IISAN_LIB"])
lib = _lib.load()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
st = torch.cuda.current_stream().cuda_stream
for name, items, S, heads, masked in (("ViT", 1408, 197, 12, False), ("BERT", 1408, 30, 12, True)):
    qkv = torch.randn(items, heads, 3, S, 64, device="cuda").half()
    ctx = torch.empty(items * S, heads * 64, device="cuda", dtype=torch.float16)
    kb = torch.zeros(items, S, device="cuda") if masked else None
    args = (0, qkv.data_ptr(), kb.data_ptr() if masked else None, ctx.data_ptr(), items, S, heads, st)
    for _ in range(3):
        assert lib.iisan_attention16(*args) == 0
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for e0, e1 in ev:
        e0.record(); lib.iisan_attention16(*args); e1.record()
    torch.cuda.synchronize()
    ts = sorted(e0.elapsed_time(e1) * 1e3 for e0, e1 in ev)
    print(f"{name}: items {items} S {S}: median {ts[len(ts) // 2]:.1f} us, min {ts[0]:.1f} us; checksum {float(ctx.float().abs().sum()):.6e}")
