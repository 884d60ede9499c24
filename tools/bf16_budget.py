#!/usr/bin/env python3
"""can bf16 operands sit inside the north-star tolerance?  The production-size step of
tests/test_gpu_trainable.py::test_production_size_step_meets_the_north_star_tolerance under fp16 and under bf16 encoder operands:
relative error of the loss and of the item embeddings against the fp32 CPU oracle, and the error of every tapped hidden state of
both towers.  The budget arithmetic that follows from the table is in DESIGN.md 3."""
import os, sys, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from iisan_amd import _lib, encoders, factory as helpers, synth, weights
from oracle import iisan_oracle as O
lib = _lib.load()
vw, bw = weights.make_vit_weights(), weights.make_bert_weights()
b = synth.scientific_batch(bs=2, seed=2024, lengths=[11, 4])
args = helpers.make_args(drop_rate=0.0)
P = weights.make_trainable_params(seed=99)
ids = b.ids.view(-1)
need = list(range(13))
with torch.no_grad():
    tc = O.vit_cls_taps(b.images, vw, weights.VIT_BASE)
    tt = O.bert_cls_taps(b.text, bw, weights.BERT_BASE)
    layers = O.side_layer_list(args.side_adapter_vit_list, False)
    ref, aux = O.model_loss_from_taps(b.ids, tc, tt, b.log_mask, b.pop_prob, P, layers)
rows = {}
for name in ("fp16", "bf16"):
    model = helpers.build_model(args, synth.SCI_ITEM_NUM, b.pop_prob, vw, weights.VIT_BASE, bw, weights.BERT_BASE, cached=False)
    helpers.load_trainables(model, P)
    enc = model.mm_encoder
    for m in (enc.cv_encoder, enc.bert_encoder.text_encoders["title"]):
        m.dtype16 = encoders.DTYPE_NAMES[name]; m._packed = None
    model.train()
    loss = model(ids.cuda(), b.images.cuda(), b.text.cuda(), b.log_mask.cuda(), 0)
    with torch.no_grad():
        score = model.score_embs(b.images.cuda(), b.text.cuda(), ids.cuda()).cpu()
        hc = enc.cv_encoder.forward_taps(b.images.cuda(), need).cpu()
        ht = enc.bert_encoder.forward_taps(b.text.cuda(), need).cpu()
    real = ids != 0
    rows[name] = dict(loss=abs(loss.item() - ref.item()) / abs(ref.item()),
                      emb=((score[real] - aux["score"][real]).norm() / aux["score"][real].norm()).item(),
                      vit=[((hc[:, l] - tc[:, l]).norm() / tc[:, l].norm()).item() for l in need],
                      bert=[((ht[:, l] - tt[:, l]).norm() / tt[:, l].norm()).item() for l in need])
print("| quantity (relative error vs the fp32 CPU oracle, production-size step, 22 item slots) | fp16 operands | bf16 operands | bf16 / fp16 |")
print("|---|---|---|---|")
print(f"| loss (north-star tolerance 1e-3) | {rows['fp16']['loss']:.2e} | {rows['bf16']['loss']:.2e} | {rows['bf16']['loss'] / max(rows['fp16']['loss'], 1e-12):.1f} |")
print(f"| item embeddings of real slots | {rows['fp16']['emb']:.2e} | {rows['bf16']['emb']:.2e} | {rows['bf16']['emb'] / rows['fp16']['emb']:.1f} |")
for l in range(1, 13):
    print(f"| ViT hidden state {l} (CLS rows) | {rows['fp16']['vit'][l]:.2e} | {rows['bf16']['vit'][l]:.2e} | {rows['bf16']['vit'][l] / rows['fp16']['vit'][l]:.1f} |")
for l in range(1, 13):
    print(f"| BERT hidden state {l} (CLS rows) | {rows['fp16']['bert'][l]:.2e} | {rows['bf16']['bert'][l]:.2e} | {rows['bf16']['bert'][l] / rows['fp16']['bert'][l]:.1f} |")
