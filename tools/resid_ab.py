#!/usr/bin/env python3
"""A/B of the residual-stream format of the frozen encoders (dev switch resid32): fp32 everywhere (rounds 1-3) against fp32 CLS rows +
fp16 token rows (round 4).  Per-layer tap error against the reference's golden taps (4 items, full-size ViT-B / BERT-base) and
forward time of the production batch (1,408 item slots, every block on every token)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import golden_io as gio
from iisan_amd import _lib, encoders, weights, synth
lib = _lib.load()


def rel(a, b):
    a, b = a.double(), b.double()
    return ((a - b).norm() / b.norm()).item()


z, vw, bw, b = gio.encoders_full_inputs()
ref_c, ref_t = torch.from_numpy(z["taps_cv"]), torch.from_numpy(z["taps_text"])
for dt, name in ((_lib.IISAN_F16, "fp16"), (_lib.IISAN_BF16, "bf16")):
    vit = encoders.PackedVit(vw, weights.VIT_BASE, "cuda", dt)
    bert = encoders.PackedBert(bw, weights.BERT_BASE, "cuda", dt)
    res = {}
    for r32 in (1, 0):
        _lib.dev_set("resid32", r32)
        _lib.dev_set("full_blocks", 1)
        tc = vit.forward_taps(b.images.cuda(), list(range(13))).cpu()
        tt = bert.forward_taps(b.text.cuda(), list(range(13))).cpu()
        res[r32] = (tc, tt)
    _lib.dev_set("resid32", 0); _lib.dev_set("full_blocks", 0)
    print(f"== {name} operands: relative Frobenius error of tap l vs the reference golden   [ViT fp32-stream, ViT mixed | BERT fp32-stream, BERT mixed | mixed vs fp32-stream ViT, BERT]")
    for l in range(13):
        print(f"  tap {l:2d}: {rel(res[1][0][:, l], ref_c[:, l]):.3e} {rel(res[0][0][:, l], ref_c[:, l]):.3e} | "
              f"{rel(res[1][1][:, l], ref_t[:, l]):.3e} {rel(res[0][1][:, l], ref_t[:, l]):.3e} | "
              f"{rel(res[0][0][:, l], res[1][0][:, l]):.3e} {rel(res[0][1][:, l], res[1][1][:, l]):.3e}", flush=True)

vw2, bw2 = weights.make_vit_weights(), weights.make_bert_weights()
bb = synth.scientific_batch(bs=128, seed=12345, device="cuda", images_on_device=True)
vit = encoders.PackedVit(vw2, weights.VIT_BASE, "cuda")
bert = encoders.PackedBert(bw2, weights.BERT_BASE, "cuda")
sel = [0, 2, 4, 6, 8, 10, 12]
_lib.dev_set("full_blocks", 1)
taps = {}
for rnd in range(3):
    for r32 in (1, 0):
        _lib.dev_set("resid32", r32)
        for enc, x, nm in ((vit, bb.images, "vit"), (bert, bb.text, "bert")):
            t = enc.forward_taps(x, sel); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                t = enc.forward_taps(x, sel)
            torch.cuda.synchronize()
            taps[(r32, nm)] = t
            print(f"round {rnd} resid32={r32} {nm}: {(time.perf_counter() - t0) / 3 * 1e3:.2f} ms", flush=True)
_lib.dev_set("resid32", 0); _lib.dev_set("full_blocks", 0)
for nm in ("vit", "bert"):
    print(nm, "production batch, mixed vs fp32 stream per tap:", " ".join(f"{rel(taps[(0, nm)][:, k], taps[(1, nm)][:, k]):.2e}" for k in range(len(sel))))
