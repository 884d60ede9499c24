"""Pins the CPU oracle (`oracle/iisan_oracle.py`) against vectors produced by the real reference
(`tests/golden/make_golden.py`).  CPU only."""
import json
import os

import numpy as np
import pytest
import torch

import golden_io as gio
from oracle import iisan_oracle as O


def _close(a, b, rtol, atol, what):
    a = torch.as_tensor(a, dtype=torch.float64)
    b = torch.as_tensor(b, dtype=torch.float64)
    err = (a - b).abs().max().item()
    ref = b.abs().max().item()
    assert err <= atol + rtol * ref, f"{what}: max|err| {err:.3e} vs scale {ref:.3e}"


def test_encoders_full_taps_match_reference():
    z, vw, bw, b = gio.encoders_full_inputs()
    torch.set_num_threads(8)
    with torch.no_grad():
        hs = O.vit_hidden_states(b.images, vw, gio.weights.VIT_BASE)
        taps_cv = torch.stack([h[:, 0] for h in hs], 1)
        ht = O.bert_hidden_states(b.text, bw, gio.weights.BERT_BASE)
        taps_tx = torch.stack([h[:, 0] for h in ht], 1)
    _close(taps_cv, z["taps_cv"], 2e-5, 2e-5, "ViT CLS taps")
    _close(taps_tx, z["taps_text"], 2e-5, 2e-5, "BERT CLS taps")
    _close(hs[1][1], z["vit_h1_img1"], 2e-5, 2e-5, "ViT layer-1 tokens")
    _close(ht[1][1], z["bert_h1_item1"], 2e-5, 2e-5, "BERT layer-1 tokens")


@pytest.mark.parametrize("variant", ["default", "gelu", "rmfirst"])
def test_sidenet_sasrec_loss_and_grads_match_reference(variant):
    z, b, taps_cv, taps_tx, P, kw = gio.sidenet_full_inputs(variant)
    P = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    layers = O.side_layer_list("1,3,5,7,9,11", kw.get("remove_first", False))
    loss, aux = O.model_loss_from_taps(b.ids, taps_cv, taps_tx, b.log_mask, b.pop_prob, P, layers,
                                       cv_head="mm_encoder.cv_pre_fc.", text_head="mm_encoder.bert_pre_fc.", **kw)
    pre = variant + "/"
    for k in ("cv", "text", "mm", "score", "prec"):
        _close(aux[k].detach(), z[pre + k], 1e-5, 1e-5, f"{variant}:{k}")
    _close(loss.detach(), z[pre + "loss"], 1e-5, 0, f"{variant}:loss")
    loss.backward()
    for n, p in P.items():
        g = gio.sample_like_golden(p.grad)
        _close(g, z[pre + "g/" + n], 2e-4, 1e-7, f"{variant}:grad {n}")
        gn = z[pre + "gn/" + n]
        assert abs(float(p.grad.double().norm()) - gn[0]) <= 2e-4 * gn[0] + 1e-7, n


def test_adam_step_matches_reference():
    z, b, taps_cv, taps_tx, P, kw = gio.sidenet_full_inputs("default")
    P = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    layers = O.side_layer_list("1,3,5,7,9,11", False)
    loss, _ = O.model_loss_from_taps(b.ids, taps_cv, taps_tx, b.log_mask, b.pop_prob, P, layers,
                                     cv_head="mm_encoder.cv_pre_fc.", text_head="mm_encoder.bert_pre_fc.")
    loss.backward()
    lrs = dict(recsys=2e-4, adapter_cv=1e-4, adapter_text=1e-4, image_net=1e-4, text_encoder=5e-5)
    for n, p in P.items():
        new, _, _ = O.adam_step(p.detach(), p.grad, torch.zeros_like(p), torch.zeros_like(p), 1, lrs[O.adam_group_of(n)])
        _close(gio.sample_like_golden(new - p.detach()), z["default/adam/" + n], 1e-3, 1e-9, f"adam delta {n}")


def test_adam_groups_match_reference_rule():
    groups = json.load(open(os.path.join(gio.GOLDEN, "adam_groups.json")))
    assert len(groups) == 146
    for n, g in groups.items():
        assert O.adam_group_of(n) == g, n
    sizes = {}
    for g in groups.values():
        sizes[g] = sizes.get(g, 0) + 1
    # SURVEY.md §5.6 [probe]: 51 / 56 / 28 / 9 / 2
    assert sizes == dict(recsys=51, adapter_cv=56, adapter_text=28, image_net=9, text_encoder=2)


@pytest.mark.parametrize("case", ["e2e_small", "e2e_bs8"])
def test_e2e_small_matches_reference(case):
    z, vw, bw, b, P = gio.e2e_small_inputs() if case == "e2e_small" else gio.e2e_small_inputs("e2e_bs8", gio.E2E_BS8_LENGTHS)
    P = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    with torch.no_grad():
        taps_cv = O.vit_cls_taps(b.images, vw, gio.E2E_VIT)
        taps_tx = O.bert_cls_taps(b.text, bw, gio.E2E_BERT)
    layers = O.side_layer_list("0,1", False)
    loss, aux = O.model_loss_from_taps(b.ids, taps_cv, taps_tx, b.log_mask, b.pop_prob, P, layers)
    _close(aux["cv"].detach(), z["cv"], 2e-5, 2e-5, "cv")
    _close(aux["text"].detach(), z["text_emb"], 2e-5, 2e-5, "text")
    _close(aux["mm"].detach(), z["mm"], 2e-5, 2e-5, "mm")
    _close(loss.detach(), z["loss"], 2e-5, 0, "loss")
    loss.backward()
    for n, p in P.items():
        _close(gio.sample_like_golden(p.grad), z["g/" + n], 5e-4, 1e-7, f"grad {n}")


def test_modality_inter_matches_reference():
    """`--modality inter` (Code_Uncached/model/model.py:38-39,70-72,182-205): only the inter-modal tower exists and
    `com_dense` is 64 -> 64 on its output.  Fixture `e2e_inter.npz` = the reference run with that flag; tensors outside
    the loss (the two encoder heads) have no gradient there and none here."""
    z, vw, bw, b, P = gio.e2e_small_inputs("e2e_inter", modality="inter")
    assert P["com_dense.weight"].shape == (64, 64) and not any("cv_adapter" in k or "bert_adapter" in k or "fc_cv" in k for k in P)
    P = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    with torch.no_grad():
        taps_cv = O.vit_cls_taps(b.images, vw, gio.E2E_VIT)
        taps_tx = O.bert_cls_taps(b.text, bw, gio.E2E_BERT)
    loss, aux = O.model_loss_from_taps(b.ids, taps_cv, taps_tx, b.log_mask, b.pop_prob, P, O.side_layer_list("0,1", False),
                                       modality="inter")
    assert aux["cv"] is None and aux["text"] is None
    _close(aux["mm"].detach(), z["mm"], 2e-5, 2e-5, "mm")
    _close(loss.detach(), z["loss"], 2e-5, 0, "loss")
    loss.backward()
    n_checked = 0
    for n, p in P.items():
        if "g/" + n in z.files:
            _close(gio.sample_like_golden(p.grad), z["g/" + n], 5e-4, 1e-7, f"grad {n}")
            n_checked += 1
        else:
            assert p.grad is None and ("classifier" in n or "title.fc" in n), n
    assert n_checked == len(P) - 4
    with pytest.raises(NotImplementedError):
        O.model_loss_from_taps(b.ids, taps_cv, taps_tx, b.log_mask, b.pop_prob, P, O.side_layer_list("0,1", False), modality="intra")


def test_eval_ranks_match_reference():
    z, seqs, tables, P = gio.eval_inputs()
    item_emb = torch.nn.functional.linear(torch.cat(tables, 1), P["com_dense.weight"], P["com_dense.bias"])
    S = 10
    rows, masks, hist, tgt = [], [], [], []
    for seq in seqs:
        tokens = seq[:-1]
        pad = S - len(tokens)
        rows.append(item_emb[[0] * pad + tokens])
        masks.append([0.0] * pad + [1.0] * len(tokens))
        hist.append(torch.tensor(tokens))
        tgt.append(seq[-1])
    x = torch.stack(rows)
    lm = torch.tensor(masks)
    prec = O.sasrec(x, lm, P, 2, 2)[:, -1]
    ranks = O.eval_ranks(prec, item_emb, hist, torch.tensor(tgt))
    ref = torch.from_numpy(z["ranks"])
    in_hist = torch.tensor([t in s[:-1] for s, t in zip(seqs, tgt)])
    # targets scored -inf tie with the rest of the history: only their tie ORDER is unspecified in the reference
    assert torch.equal(ranks[~in_hist], ref[~in_hist])
    n_items = int(z["item_num"])
    for r_o, r_r, s in zip(ranks[in_hist], ref[in_hist], [s for s, f in zip(seqs, in_hist) if f]):
        lo = n_items - len(set(s[:-1])) + 1
        assert lo <= int(r_o) <= n_items and lo <= int(r_r) <= n_items
    hit, ndcg = O.hit_ndcg(ranks)
    assert abs(float(hit.mean()) - float(z["hit10"])) < 1e-6
    assert abs(float(ndcg.mean()) - float(z["ndcg10"])) < 1e-6


def test_eval_topk_matches_the_reference_argsort():
    """The oracle's recommendation list (`O.eval_topk`: stable descending argsort of the history-masked row, metrics.py:59-67,
    198-207) against the first ten positions of the reference's OWN `torch.argsort` inside `metrics_topK`, captured while
    `eval_model` ran unmodified (tests/golden/make_golden.py: gen_eval).  300 items, 50 users: no score ties, so the
    reference's unspecified tie order does not enter."""
    z, seqs, tables, P = gio.eval_inputs()
    item_emb = torch.nn.functional.linear(torch.cat(tables, 1), P["com_dense.weight"], P["com_dense.bias"])
    S = 10
    rows, masks, hist = [], [], []
    for seq in seqs:
        tokens = seq[:-1]
        pad = S - len(tokens)
        rows.append(item_emb[[0] * pad + tokens])
        masks.append([0.0] * pad + [1.0] * len(tokens))
        hist.append(torch.tensor(tokens))
    prec = O.sasrec(torch.stack(rows), torch.tensor(masks), P, 2, 2)[:, -1]
    top = O.eval_topk(prec, item_emb, hist, 10)
    assert torch.equal(top, torch.from_numpy(z["top10"]).long())
    # consistent with the rank oracle: a target ranked r <= 10 sits at position r - 1 of the list
    tgt = torch.tensor([s[-1] for s in seqs])
    ranks = O.eval_ranks(prec, item_emb, hist, tgt)
    for u in range(len(seqs)):
        if int(ranks[u]) <= 10:
            assert int(top[u, int(ranks[u]) - 1]) == int(tgt[u])
    # history items never appear; fewer than k candidates: the excluded items follow in ascending id order
    for u, h in enumerate(hist):
        assert not set(top[u].tolist()) & set(h.tolist())
    tiny = O.eval_topk(prec[:1], item_emb[:6], [torch.tensor([2, 4])], 5)
    assert sorted(tiny[0, :3].tolist()) == [1, 3, 5] and tiny[0, 3:].tolist() == [2, 4]


@pytest.mark.parametrize("variant", ["text_wide_long", "image_wide_long", "equal_rmfirst"])
def test_versa_side_network_matches_reference(variant):
    z, b, taps_cv, taps_tx, args, model, P = gio.versa_inputs(variant)
    P = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    lc = O.side_layer_list(args.side_adapter_vit_list, args.remove_first == "TRUE")
    lt = O.side_layer_list(args.side_adapter_bert_list, args.remove_first == "TRUE")
    cv, text, mm = O.versa_side_network(taps_cv, taps_tx, P, lc, lt, activation=args.adapter_activation,
                                        remove_first=args.remove_first == "TRUE")
    pre = variant + "/"
    _close(cv.detach(), z[pre + "cv"], 2e-5, 2e-5, "cv")
    _close(text.detach(), z[pre + "text"], 2e-5, 2e-5, "text")
    _close(mm.detach(), z[pre + "mm"], 2e-5, 2e-5, "mm")
    bs, S = b.log_mask.shape
    score = torch.nn.functional.linear(torch.cat([cv, text, mm], 1), P["com_dense.weight"], P["com_dense.bias"])
    prec = O.sasrec(score.view(bs, S + 1, 64)[:, :-1], b.log_mask, P, 2, 2).reshape(-1, 64)
    loss = O.inbatch_ce(b.ids, score, prec, b.log_mask, b.pop_prob)
    _close(loss.detach(), z[pre + "loss"], 2e-5, 0, "loss")
    loss.backward()
    for n, p in P.items():
        _close(gio.sample_like_golden(p.grad), z[pre + "g/" + n], 5e-4, 1e-7, f"grad {n}")
