#!/usr/bin/env python3
"""Gate-gradient conditioning of the e2e_inter fixture under both residual-stream formats (diagnostic for
tests/test_gpu_trainable.py::test_modality_inter_end_to_end_matches_reference)."""
import os, sys, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import golden_io as gio
from iisan_amd import _lib, factory as helpers
from oracle import iisan_oracle as O
lib = _lib.load()
for r32 in (1, 0):
    _lib.dev_set("resid32", r32)
    z, vw, bw, b, P = gio.e2e_small_inputs("e2e_inter", modality="inter")
    args = helpers.make_args(side_adapter_vit_list="0,1", side_adapter_bert_list="0,1", num_words_title=8, modality="inter")
    model = helpers.build_model(args, 40, b.pop_prob, vw, gio.E2E_VIT, bw, gio.E2E_BERT, cached=False)
    helpers.load_trainables(model, P); model.eval()
    ids, lm = b.ids.cuda().view(-1), b.log_mask.cuda()
    img, txt = b.images.cuda(), b.text.cuda()
    loss = model(ids, img, txt, lm, 0); loss.backward()
    tc = model.mm_encoder.cv_encoder.forward_taps(img, [0, 1, 2]).cpu()
    tt = model.mm_encoder.bert_encoder.forward_taps(txt, [0, 1, 2]).cpu()
    for dtype in (torch.float32, torch.float64):
        Pg = {k: v.clone().to(dtype).requires_grad_(True) for k, v in P.items()}
        lo, _ = O.model_loss_from_taps(b.ids, tc.to(dtype), tt.to(dtype), b.log_mask.to(dtype), b.pop_prob.to(dtype), Pg, O.side_layer_list("0,1", False), modality="inter")
        lo.backward()
        for n, p in model.named_parameters():
            if p.requires_grad and Pg[n].grad is not None and (dtype == torch.float64):
                g, go = p.grad.cpu().double(), Pg[n].grad.double()
                print(f"resid32={r32} oracle {str(dtype)[6:]}: {n}: |g| {go.norm().item():.3e} abs err {(g-go).norm().item():.3e} rel {((g-go).norm()/go.norm()).item():.2e}")
        if dtype == torch.float64: print("loss hip", loss.item(), "oracle", lo.item())
_lib.dev_set("resid32", 0)
