#!/usr/bin/env python3
"""Time the frozen-encoder HIP path alone (development aid; bench.py is the contract benchmark)."""
import argparse
import sys, os, time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iisan_amd import _lib, encoders, synth, weights

ap = argparse.ArgumentParser()
ap.add_argument("--items", type=int, default=1408)
ap.add_argument("--iters", type=int, default=3)
ap.add_argument("--dtype", default="fp16")
ap.add_argument("--chunk", type=int, default=0)
ap.add_argument("--bert", type=int, default=1)
a = ap.parse_args()
dt = encoders.DTYPE_NAMES[a.dtype]
vit = encoders.PackedVit(weights.make_vit_weights(), weights.VIT_BASE, "cuda", dt)
bert = encoders.PackedBert(weights.make_bert_weights(), weights.BERT_BASE, "cuda", dt)
M = a.items
img = torch.randn(M, 3, 224, 224, device="cuda").clamp_(-1, 1)
text = torch.zeros(M, 60, dtype=torch.int64, device="cuda")
text[:, :30] = torch.randint(1000, 30000, (M, 30), device="cuda")
text[:, 30:] = 1
sel = [0, 2, 4, 6, 8, 10, 12]
for name, fn, gf in (("vit", lambda: vit.forward_taps(img, sel, a.chunk), 35.126), ("bert", lambda: bert.forward_taps(text, sel, a.chunk), 5.129)):
    if name == "bert" and not a.bert:
        continue
    fn(); torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(a.iters):
        fn()
    torch.cuda.synchronize()
    dtm = (time.time() - t0) / a.iters
    print(f"{name}: {dtm*1e3:.2f} ms / {M} items -> {M/dtm:.0f} items/s, {gf*M/dtm/1e3:.1f} TFLOP/s ({a.dtype}, chunk {a.chunk})")
