#!/usr/bin/env python3
"""A/B of the ViT tower with its LayerNorms applied in the epilogues of the QKV / FC1 products (dev switch ln_fold(1), round 4: the add
kernels write the fp16 stream + rstd per row, the products read the stream against gamma-folded, centred weights) against the
LayerNorm images of round 4.  Per-layer tap error against the reference's golden taps (4 items; gemm16_h256 forced, which is what
the production batch runs) and forward time of the production batch (1,408 item slots, every block on every token)."""
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
ref_c = torch.from_numpy(z["taps_cv"])
vit = encoders.PackedVit(vw, weights.VIT_BASE, "cuda")
res = {}
_lib.dev_set("gemm16_variant", 4)
for fb in (1, 0):
    _lib.dev_set("full_blocks", fb)
    for fold in (0, 1, 2):
        _lib.dev_set("ln_fold", fold)
        res[(fb, fold)] = vit.forward_taps(b.images.cuda(), list(range(13))).cpu()
_lib.dev_set("gemm16_variant", 0); _lib.dev_set("ln_fold", 2); _lib.dev_set("full_blocks", 0)
print("relative Frobenius error of ViT tap l vs the reference golden  [every block on every token: images (0), LN in the epilogues (1), + adds in the epilogues (2) | CLS-only last block: 0, 1, 2 | 2 vs 0]")
for l in range(13):
    print(f"  tap {l:2d}: " + " ".join(f"{rel(res[(1, f)][:, l], ref_c[:, l]):.3e}" for f in (0, 1, 2)) + " | " + " ".join(f"{rel(res[(0, f)][:, l], ref_c[:, l]):.3e}" for f in (0, 1, 2)) +
          f" | {rel(res[(1, 2)][:, l], res[(1, 0)][:, l]):.3e}", flush=True)

vw2 = weights.make_vit_weights()
bb = synth.scientific_batch(bs=128, seed=12345, device="cuda", images_on_device=True)
vit = encoders.PackedVit(vw2, weights.VIT_BASE, "cuda")
sel = [0, 2, 4, 6, 8, 10, 12]
_lib.dev_set("full_blocks", 1)
taps = {}
for rnd in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    for fold in (0, 1, 2):
        _lib.dev_set("ln_fold", fold)
        t = vit.forward_taps(bb.images, sel); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            t = vit.forward_taps(bb.images, sel)
        torch.cuda.synchronize()
        taps[fold] = t
        print(f"round {rnd} ln_fold={fold} vit: {(time.perf_counter() - t0) / 3 * 1e3:.2f} ms", flush=True)
_lib.dev_set("ln_fold", 2); _lib.dev_set("full_blocks", 0)
for f in (1, 2):
    print(f"production batch, ln_fold={f} vs images per tap:", " ".join(f"{rel(taps[f][:, k], taps[0][:, k]):.2e}" for k in range(len(sel))))
