#!/usr/bin/env python3
"""The parity numbers behind the tolerances of tests/test_gpu_trainable.py, written down: markdown on stdout.

  (A) bs = 128, the headline's shape and DEFAULT routes, against the fp32 CPU oracle fed the same inputs
      (test_default_dispatch_at_bs128_against_the_oracle_directly): training-loss error, per-layer tap error of both towers over all
      1,408 slots, the worst single slot per layer — with the CLS-only last block (library default) and with every block on every token
      (the bench headline);
  (B) bs = 2 (22 slots, 13 prediction rows), seeds x kernel routes (test_production_size_step_meets_the_north_star_tolerance): the
      scatter of the loss error that the 2.5e-3 bound of that test envelopes.

usage (GPU box):  python tools/parity_report.py [seeds...] > gpurun_out/r6_parity.md      -> copy to profiles/r6_parity.md
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers  # noqa: E402
from iisan_amd import _lib, synth, weights  # noqa: E402
from oracle import iisan_oracle as O  # noqa: E402

NEED = [0, 2, 4, 6, 8, 10, 12]


def build(b, args, vw, bw):
    model = helpers.build_model(args, synth.SCI_ITEM_NUM, b.pop_prob, vw, weights.VIT_BASE, bw, weights.BERT_BASE, cached=False)
    P = weights.make_trainable_params(seed=99)
    helpers.load_trainables(model, P)
    model.train()
    return model, P


def hip_step(model, b, ids):
    d = b.to("cuda")
    loss = model(ids.cuda(), d.images, d.text, d.log_mask, 0)
    with torch.no_grad():
        enc = model.mm_encoder
        hc = enc.cv_encoder.forward_taps(d.images, NEED).cpu()
        ht = enc.bert_encoder.forward_taps(d.text, NEED).cpu()
    return loss.item(), hc, ht


def part_a(vw, bw):
    b = synth.scientific_batch(bs=128, seed=12345)
    args = helpers.make_args(drop_rate=0.0)
    model, P = build(b, args, vw, bw)
    ids = b.ids.view(-1)
    M = ids.numel()
    real = torch.nonzero(ids != 0).view(-1)
    pad = torch.nonzero(ids == 0).view(-1)
    rows = torch.cat([real, pad[:1]])
    oc, ot = torch.empty(M, 13, 768), torch.empty(M, 13, 768)
    with torch.no_grad():
        for i in range(0, rows.numel(), 32):
            r = rows[i:i + 32]
            oc[r] = O.vit_cls_taps(b.images[r], vw, weights.VIT_BASE)
            ot[r] = O.bert_cls_taps(b.text[r], bw, weights.BERT_BASE)
        oc[pad] = oc[pad[0]].clone()
        ot[pad] = ot[pad[0]].clone()
        ref, _ = O.model_loss_from_taps(b.ids, oc, ot, b.log_mask, b.pop_prob, P, O.side_layer_list(args.side_adapter_vit_list, False))
    print(f"## (A) bs = 128 ({M} item slots: {real.numel()} real, {pad.numel()} padding), default routes `{_lib.dev_state() or 'library defaults'}`, "
          f"towers {'overlapped on two streams' if model.mm_encoder.overlap_towers else 'back to back'}, vs the fp32 CPU oracle\n")
    print(f"oracle loss {ref.item():.7f}\n")
    print("| encoder blocks | HIP loss | (HIP - oracle) / oracle | bound in the test |")
    print("|---|---|---|---|")
    taps = {}
    for fb in (0, 1):
        with _lib.dev(full_blocks=fb):
            loss, hc, ht = hip_step(model, b, ids)
        taps[fb] = (hc, ht)
        name = "every block on every token (bench headline)" if fb else "CLS-only last block (library default)"
        print(f"| {name} | {loss:.7f} | {(loss - ref.item()) / abs(ref.item()):+.2e} | 1e-3 |")
    for fb in (0, 1):
        hc, ht = taps[fb]
        print(f"\nTap errors, full_blocks = {fb}: relative Frobenius error of the layer over all {M} slots (bound 1.5e-3) and the worst single slot (bound 4e-3)\n")
        print("| hidden state | ViT layer | ViT worst slot (slot) | BERT layer | BERT worst slot (slot) |")
        print("|---|---|---|---|---|")
        for k, l in enumerate(NEED):
            if l == 0:
                e0c = (hc[:, 0] - oc[:, 0]).abs().max().item()
                e0t = (ht[:, 0] - ot[:, 0]).abs().max().item()
                print(f"| 0 (embeddings) | max abs {e0c:.1e} | | max abs {e0t:.1e} | |")
                continue
            ec = ((hc[:, k] - oc[:, l]).norm() / oc[:, l].norm()).item()
            et = ((ht[:, k] - ot[:, l]).norm() / ot[:, l].norm()).item()
            pc = (hc[:, k] - oc[:, l]).norm(dim=1) / oc[:, l].norm(dim=1)
            pt = (ht[:, k] - ot[:, l]).norm(dim=1) / ot[:, l].norm(dim=1)
            print(f"| {l} | {ec:.2e} | {pc.max().item():.2e} ({int(pc.argmax())}) | {et:.2e} | {pt.max().item():.2e} ({int(pt.argmax())}) |")


def part_b(vw, bw, seeds):
    routes = [(0, 0, 2, "product dispatch at 22 slots (128x128 kernels)"), (1, 0, 2, "v1 128x128 kernels forced (golden-pinned)"),
              (3, 0, 2, "staggered 256x256 kernel forced"), (4, 0, 2, "gemm16_h256 + ln_fold 2, CLS-only last block"),
              (4, 1, 2, "gemm16_h256 + ln_fold 2, every block on every token (headline kernel set)"),
              (4, 1, 0, "gemm16_h256, LayerNorm images (ln_fold 0), every block")]
    print("\n## (B) bs = 2 (22 slots, 13 prediction rows): (HIP loss - oracle) / oracle by seed and kernel route\n")
    print("The bound of `test_production_size_step_meets_the_north_star_tolerance` (2.5e-3) envelopes this scatter; the north-star 1e-3 is asserted on (A).\n")
    print("| route (gemm16_variant, full_blocks, ln_fold) | " + " | ".join(f"seed {s}" for s in seeds) + " | worst ViT tap | worst BERT tap |")
    print("|---|" + "---|" * (len(seeds) + 2))
    cells = {r[:3]: [] for r in routes}
    worst = {r[:3]: [0.0, 0.0] for r in routes}
    for seed in seeds:
        b = synth.scientific_batch(bs=2, seed=seed, lengths=[11, 4])
        args = helpers.make_args(drop_rate=0.0)
        model, P = build(b, args, vw, bw)
        ids = b.ids.view(-1)
        with torch.no_grad():
            tc = O.vit_cls_taps(b.images, vw, weights.VIT_BASE)
            tt = O.bert_cls_taps(b.text, bw, weights.BERT_BASE)
            ref, _ = O.model_loss_from_taps(b.ids, tc, tt, b.log_mask, b.pop_prob, P, O.side_layer_list(args.side_adapter_vit_list, False))
        for v, fb, fold, _n in routes:
            with _lib.dev(gemm16_variant=v, full_blocks=fb, ln_fold=fold):
                loss, hc, ht = hip_step(model, b, ids)
            cells[(v, fb, fold)].append((loss - ref.item()) / abs(ref.item()))
            for k, l in enumerate(NEED[1:], 1):
                worst[(v, fb, fold)][0] = max(worst[(v, fb, fold)][0], ((hc[:, k] - tc[:, l]).norm() / tc[:, l].norm()).item())
                worst[(v, fb, fold)][1] = max(worst[(v, fb, fold)][1], ((ht[:, k] - tt[:, l]).norm() / tt[:, l].norm()).item())
    lo, hi = 0.0, 0.0
    for v, fb, fold, name in routes:
        c = cells[(v, fb, fold)]
        lo, hi = min(lo, min(c)), max(hi, max(c))
        print(f"| {name} ({v}, {fb}, {fold}) | " + " | ".join(f"{x:+.2e}" for x in c) + f" | {worst[(v, fb, fold)][0]:.2e} | {worst[(v, fb, fold)][1]:.2e} |")
    print(f"\nrange over all cells: {lo:+.2e} … {hi:+.2e}")


def main():
    seeds = [int(x) for x in sys.argv[1:]] or [2024, 2025, 2026, 2027, 2028]
    torch.manual_seed(0)
    vw, bw = weights.make_vit_weights(), weights.make_bert_weights()
    p = torch.cuda.get_device_properties(0)
    print(f"# Parity numbers, round 6 (`python tools/parity_report.py`; {p.name}, fp16 encoder operands, library {_lib.load().iisan_version().decode()})\n")
    part_a(vw, bw)
    part_b(vw, bw, seeds)


if __name__ == "__main__":
    main()
