#!/usr/bin/env python3
"""A/B of the one-launch SASRec kernels (sasrec_fused.hip) against the per-operator launches (sasrec.hip): outputs, gradients and
device time of forward / backward at the Uncached (bs = 128) and Cached (bs = 1024) batch sizes, eval and training-mode dropout."""
import os, sys, time
import torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import golden_io as gio
from iisan_amd import _lib, ops
lib = _lib.load()
S, E, H, L = 10, 64, 2, 2
P = {k: v for k, v in gio.weights.make_trainable_params(seed=99).items() if k.startswith("user_encoder.")}
order = ops.sasrec_param_order(L)


def run(B, p, fused, iters=0):
    _lib.dev_set("sasrec_fused", fused)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, S, E, generator=g).cuda().requires_grad_(True)
    lm = (torch.rand(B, S, generator=g) > 0.3).float(); lm[:, -1] = 1; lm = lm.cuda()
    w = torch.randn(B, S, E, generator=g).cuda()
    params = [P["user_encoder.transformer_encoder." + k].cuda().requires_grad_(True) for k in order]
    cfg = ops.make_sasrec_cfg(S, E, H, L, p, 123456789012345)
    y = ops.SasrecFn.apply(cfg, x, lm, *params)
    (y * w).sum().backward()
    torch.cuda.synchronize()
    res = [y.detach(), x.grad] + [t.grad for t in params]
    tf = tb = None
    if iters:
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        tf = tb = 0.0
        for _ in range(iters):
            for t in params: t.grad = None
            x.grad = None
            e[0].record(); y = ops.SasrecFn.apply(cfg, x, lm, *params); e[1].record(); (y * w).sum().backward(); e[2].record()
            torch.cuda.synchronize()
            tf += e[0].elapsed_time(e[1]); tb += e[1].elapsed_time(e[2])
        tf /= iters; tb /= iters
    return res, tf, tb


for B in ((37, 128, 1024) if len(sys.argv) < 2 else ()):
    for p in (0.0, 0.1):
        r1, tf1, tb1 = run(B, p, 1, 5)
        r0, tf0, tb0 = run(B, p, 0, 5)
        worst = max(((a - b).norm() / (b.norm() + 1e-12)).item() for a, b in zip(r1, r0))
        names = ["y", "dx"] + order
        wi = max(range(len(r1)), key=lambda i: ((r1[i] - r0[i]).norm() / (r0[i].norm() + 1e-12)).item())
        print(f"B={B} p={p}: fused vs per-operator worst rel {worst:.2e} ({names[wi]}); fwd {tf1:.3f} vs {tf0:.3f} ms, fwd+bwd-autograd {tb1:.3f} vs {tb0:.3f} ms", flush=True)
_lib.dev_set("sasrec_fused", 1)

# which of the two is right where they differ?  both against the fp32 oracle (the fp64 one keeps q.k beside the -1e9 mask that fp32 absorbs) at B = 1024 (eval mode)
from oracle import iisan_oracle as O
B = 1024
g = torch.Generator().manual_seed(5)
x = torch.randn(B, S, E, generator=g)
lm = (torch.rand(B, S, generator=g) > 0.3).float(); lm[:, -1] = 1
w = torch.randn(B, S, E, generator=g)
Po = {k: v.clone().requires_grad_(True) for k, v in P.items()}
xo = x.clone().requires_grad_(True)
yo = O.sasrec(xo, lm, Po, H, L)
(yo * w).sum().backward()
for fused in (1, 0):
    r, _, _ = run(B, 0.0, fused)
    names = ["y", "dx"] + order
    ref = [yo.detach(), xo.grad] + [Po["user_encoder.transformer_encoder." + k].grad for k in order]
    errs = [((a.cpu().double() - b.double()).norm() / (b.norm() + 1e-30)).item() for a, b in zip(r, ref)]
    wi = max(range(len(errs)), key=lambda i: errs[i])
    print(f"B=1024 fused={fused} vs fp32 oracle (the fp64 one keeps q.k beside the -1e9 mask that fp32 absorbs): worst rel {errs[wi]:.2e} ({names[wi]}), median {sorted(errs)[len(errs)//2]:.2e}; y {errs[0]:.2e} dx {errs[1]:.2e}")
_lib.dev_set("sasrec_fused", 1)
