"""Same-box sweep of the split-operand loss passes (csrc/ce.hip ce16_*) at the Cached batch size: forward + backward of `ops.InbatchCeFn` alone,
HIP-event timed, over dev-knob settings.  usage (GPU box): python tools/ce_sweep.py [bs]"""
import sys
import torch
sys.path.insert(0, ".")
from iisan_amd import _lib, ops, synth

bs = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
S, E = 10, 64
b = synth.scientific_batch(bs=bs, seed=77, res=2, words=2, dup_items=True)
g = torch.Generator().manual_seed(5)
ids, lm, pop = b.ids.view(-1).cuda(), b.log_mask.float().cuda(), b.pop_prob.float().cuda()
score = (torch.randn(bs * (S + 1), E, generator=g) * 0.3).cuda().requires_grad_(True)
prec = (torch.randn(bs * S, E, generator=g) * 0.3).cuda().requires_grad_(True)


def run(n=20):
    for _ in range(3):
        ops.InbatchCeFn.apply(ids, score, prec, lm, pop).backward()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        ops.InbatchCeFn.apply(ids, score, prec, lm, pop).backward()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for rep in range(2):
    with _lib.dev(ce_fast=4):
        print(f"f32 fused passes: {run():.0f} us")
    for nsub in (1, 2):
        for ys in (0, 4, 5, 6, 8, 10, 11, 12, 16):
            with _lib.dev(ce16_nsub=nsub, ce16_ys=ys):
                print(f"split operands nsub {nsub} ys {ys or 'auto'}: {run():.0f} us")
