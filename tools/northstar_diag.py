#!/usr/bin/env python3
"""Where does the bs = 2 north-star loss sit, route by route?  (tests/test_gpu_trainable.py::test_production_size_step_meets_the_north_star_tolerance)
Prints loss rel vs the oracle, item-embedding rel and the per-tap rel of both towers for every (gemm16_variant, full_blocks, ln_fold)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers
from iisan_amd import _lib, synth, weights
from oracle import iisan_oracle as O

seeds = [int(x) for x in sys.argv[1:]] or [2024]
vw, bw = weights.make_vit_weights(), weights.make_bert_weights()
for seed in seeds:
    b = synth.scientific_batch(bs=2, seed=seed, lengths=[11, 4])
    args = helpers.make_args(drop_rate=0.0)
    model = helpers.build_model(args, synth.SCI_ITEM_NUM, b.pop_prob, vw, weights.VIT_BASE, bw, weights.BERT_BASE, cached=False)
    P = weights.make_trainable_params(seed=99)
    helpers.load_trainables(model, P)
    model.train()
    ids = b.ids.view(-1)
    need = [0, 2, 4, 6, 8, 10, 12]
    with torch.no_grad():
        tc = O.vit_cls_taps(b.images, vw, weights.VIT_BASE)
        tt = O.bert_cls_taps(b.text, bw, weights.BERT_BASE)
        layers = O.side_layer_list(args.side_adapter_vit_list, False)
        ref, aux = O.model_loss_from_taps(b.ids, tc, tt, b.log_mask, b.pop_prob, P, layers)
    for variant, fb, fold in [(0, 0, 2), (1, 0, 2), (3, 0, 2), (4, 0, 0), (4, 0, 1), (4, 0, 2), (4, 1, 0), (4, 1, 1), (4, 1, 2)]:
        with _lib.dev(gemm16_variant=variant, full_blocks=fb, ln_fold=fold):
            loss = model(ids.cuda(), b.images.cuda(), b.text.cuda(), b.log_mask.cuda(), 0)
            with torch.no_grad():
                enc = model.mm_encoder
                hc = enc.cv_encoder.forward_taps(b.images.cuda(), need).cpu()
                ht = enc.bert_encoder.forward_taps(b.text.cuda(), need).cpu()
        rel = (loss.item() - ref.item()) / abs(ref.item())
        ec = [((hc[:, k] - tc[:, l]).norm() / tc[:, l].norm()).item() for k, l in enumerate(need[1:], 1)]
        et = [((ht[:, k] - tt[:, l]).norm() / tt[:, l].norm()).item() for k, l in enumerate(need[1:], 1)]
        print(f"seed {seed} variant {variant} full_blocks {fb} ln_fold {fold}: loss rel {rel:+.2e}  ViT taps " + " ".join(f"{e:.1e}" for e in ec)
              + "  BERT taps " + " ".join(f"{e:.1e}" for e in et), flush=True)
