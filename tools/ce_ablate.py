#!/usr/bin/env python3
"""Ablation of the fused in-batch CE row pass at the Cached batch size (timing only: results are wrong with a bit set).
Needs a library built with -DCE_ABLATE (make -C iisan_amd/csrc EXTRA=-DCE_ABLATE); the product build compiles the
ablation branches out — measured on MI355X: full 634 us per forward call, no logits MFMAs 520, no d_prec MFMAs 484, none 407."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from iisan_amd import _lib, ops, synth
lib = _lib.load()
bs, S, E = 1024, 10, 64
b = synth.scientific_batch(bs=bs, seed=77, res=2, words=2, dup_items=True)
g = torch.Generator().manual_seed(5)
ids, lm, pop = b.ids.view(-1).cuda(), b.log_mask.float().cuda(), b.pop_prob.float().cuda()
score = (torch.randn(bs * (S + 1), E, generator=g) * 0.3).cuda()
prec = (torch.randn(bs * S, E, generator=g) * 0.3).cuda()
for bits, name in ((0, "full"), (1, "no logits MFMAs"), (2, "no d_prec MFMAs"), (3, "no MFMAs")):
    _lib.dev_set("ce_debug", bits)
    for _ in range(3):
        ops.InbatchCeFn.apply(ids, score, prec, lm, pop)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.InbatchCeFn.apply(ids, score, prec, lm, pop)
    e1.record(); torch.cuda.synchronize()
    print(f"{name:20s} forward (prep + fused row pass + reduce) {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us")
_lib.dev_set("ce_debug", 0)
