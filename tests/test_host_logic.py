"""CPU tests of the host-side logic: module tree / state-dict keys, freeze + Adam-group rules, parameter tables of the
C ABI, flat trainer storage, synthetic batches, HF <-> canonical weight conversion."""
import json
import os

import numpy as np
import torch

import golden_io as gio
import helpers
from iisan_amd import ops, synth, trainer, weights


def _tiny_cfgs():
    return (weights.VitConfig(hidden=768, layers=1, heads=12, mlp=128, image=32, patch=16),
            weights.BertConfig(hidden=768, layers=1, heads=12, mlp=128, vocab=64, max_pos=16))


def test_uncached_module_tree_has_the_reference_trainable_names():
    args = helpers.make_args()
    model = helpers.build_model(args, 100, torch.ones(101), cached=False, device="cpu")
    names = {n for n, p in model.named_parameters() if p.requires_grad}
    want = set(weights.trainable_shapes())
    assert names == want, names ^ want
    shapes = weights.trainable_shapes()
    for n, p in model.named_parameters():
        if p.requires_grad:
            assert tuple(p.shape) == shapes[n], n
    assert sum(p.numel() for p in model.parameters() if p.requires_grad) == 4113877


def test_cached_module_tree_names():
    args = helpers.make_args()
    model = helpers.build_model(args, 100, torch.ones(101), cached=True, device="cpu")
    names = {n for n, p in model.named_parameters() if p.requires_grad}
    assert names == set(weights.trainable_shapes(cached=True))


def test_adam_groups_follow_reference_rule():
    groups = json.load(open(os.path.join(gio.GOLDEN, "adam_groups.json")))
    for n, g in groups.items():
        assert trainer.adam_group_of(n) == g, n
    args = helpers.make_args()
    model = helpers.build_model(args, 100, torch.ones(101), cached=False, device="cpu")
    pg = trainer.build_param_groups(model, args)
    assert [len(g["params"]) for g in pg] == [2, 9, 51, 56, 28]          # text_encoder, image_net, recsys, adapter_cv, adapter_text
    assert [g["lr"] for g in pg] == [5e-5, 1e-4, 2e-4, 1e-4, 1e-4]


def test_abi_parameter_tables_resolve():
    args = helpers.make_args()
    for cached in (False, True):
        model = helpers.build_model(args, 100, torch.ones(101), cached=cached, device="cpu")
        side = dict(model.mm_encoder.named_parameters())
        order = ops.side_param_order(7, cached)
        assert len(order) == 15 * 7 + 12 and all(k in side for k in order)
    te = dict(model.user_encoder.transformer_encoder.named_parameters())
    order = ops.sasrec_param_order(2)
    assert len(order) == 27 and set(order) == set(te)


def test_flat_trainer_storage_is_contiguous_by_group():
    args = helpers.make_args()
    model = helpers.build_model(args, 100, torch.ones(101), cached=True, device="cpu")
    before = {n: p.detach().clone() for n, p in model.named_parameters() if p.requires_grad}
    tr = trainer.FlatTrainer(model, args)
    assert tr.n_params == 4113877 and tr.seg_end[-1] == tr.flat.numel() and len(tr.seg_end) == 5
    assert all(o % 16 == 0 for o in tr.offsets) and tr.flat.numel() - tr.n_params < 16 * len(tr.names)
    assert tr.seg_lr == [5e-5, 1e-4, 2e-4, 1e-4, 1e-4]
    base = tr.flat.data_ptr()
    for n, p in model.named_parameters():
        if p.requires_grad:
            assert torch.equal(p.detach(), before[n])
            off = (p.data_ptr() - base) // 4
            assert 0 <= off < tr.flat.numel() and p.grad.data_ptr() - tr.grad.data_ptr() == p.data_ptr() - base
    g = sorted((trainer.GROUP_ORDER.index(trainer.adam_group_of(n)), (dict(model.named_parameters())[n].data_ptr() - base) // 4)
               for n in tr.names)
    assert [o for _, o in g] == sorted(o for _, o in g)            # group index is monotone in the flat offset


def test_synthetic_batch_contract():
    b = synth.scientific_batch(bs=16, seed=1, res=32)
    assert b.ids.shape == (16, 11) and b.log_mask.shape == (16, 10) and b.images.shape == (176, 3, 32, 32)
    assert b.text.shape == (176, 60) and b.pop_prob[0] == 1 and (b.pop_prob > 0).all()
    ids = b.ids.numpy()
    for r in ids:                                                   # left padded, >= 2 real items, unique within a row
        nz = r[r != 0]
        assert len(nz) >= 2 and (r[:11 - len(nz)] == 0).all() and len(set(nz)) == len(nz)
    assert torch.equal(b.log_mask, (b.ids[:, :-1] != 0).float())
    pad = (b.ids.view(-1) == 0)
    assert (b.images[pad] == 0).all() and (b.text[pad] == 0).all()
    assert (b.text[~pad][:, 0] == 101).all() and (b.text[~pad][:, 30] == 1).all()
    assert b.images.abs().max() <= 1.0


def test_hf_weight_roundtrip():
    vcfg, bcfg = _tiny_cfgs()
    vw, bw = weights.make_vit_weights(vcfg, 1), weights.make_bert_weights(bcfg, 2)
    back = weights.vit_from_hf(weights.vit_to_hf5(vw, vcfg))
    assert set(back) == set(vw) and all(torch.equal(back[k], vw[k]) for k in vw)
    back = weights.bert_from_hf(weights.bert_to_hf(bw, bcfg))
    assert set(back) == set(bw) and all(torch.equal(back[k], bw[k]) for k in bw)
    # 4.x layout of the same ViT weights
    sd5 = weights.vit_to_hf5(vw, vcfg)
    sd4 = {}
    for k, v in sd5.items():
        k = k.replace("vit.layers.", "vit.encoder.layer.").replace("attention.q_proj", "attention.attention.query") \
             .replace("attention.k_proj", "attention.attention.key").replace("attention.v_proj", "attention.attention.value") \
             .replace("attention.o_proj", "attention.output.dense").replace("mlp.fc1", "intermediate.dense").replace("mlp.fc2", "output.dense")
        sd4[k] = v
    back4 = weights.vit_from_hf(sd4)
    assert all(torch.equal(back4[k], vw[k]) for k in vw)


def test_oracle_invariants_padding_has_zero_influence():
    """padding slots contribute exactly nothing to the loss."""
    from oracle import iisan_oracle as O
    z, b, taps_cv, taps_tx, P, kw = gio.sidenet_full_inputs("default")
    layers = O.side_layer_list("1,3,5,7,9,11", False)
    kwargs = dict(cv_head="mm_encoder.cv_pre_fc.", text_head="mm_encoder.bert_pre_fc.")
    l0, _ = O.model_loss_from_taps(b.ids, taps_cv, taps_tx, b.log_mask, b.pop_prob, P, layers, **kwargs)
    pad = (b.ids.view(-1) == 0)
    tc, tt = taps_cv.clone(), taps_tx.clone()
    tc[pad] = torch.randn_like(tc[pad]) * 3
    tt[pad] = torch.randn_like(tt[pad]) * 3
    l1, _ = O.model_loss_from_taps(b.ids, tc, tt, b.log_mask, b.pop_prob, P, layers, **kwargs)
    assert torch.equal(l0, l1)


def test_checkpoint_format_matches_torch_adam_and_round_trips(tmp_path):
    """checkpoints carry the reference's keys (utils.py:104-110) and the optimizer entry is a genuine
    torch.optim.Adam state dict over run.py's five groups — loadable by a stock Adam — and round-trips."""
    import helpers
    from iisan_amd import synth, trainer
    args = helpers.make_args()
    model = helpers.build_model(args, 30, synth.make_pop_prob(30), cached=True, device="cpu")     # wired + frozen like run.py
    tr = trainer.FlatTrainer(model, args)
    g = torch.Generator().manual_seed(3)
    real = torch.zeros(tr.flat.numel(), dtype=torch.bool)        # the alignment padding between tensors holds no state
    for (_, o, k, _) in trainer._segments(tr):
        real[o:o + k] = True
    tr.m.copy_(torch.rand(tr.m.shape, generator=g) * real)
    tr.v.copy_(torch.rand(tr.v.shape, generator=g) * real)
    tr.step_no = 7
    before = tr.flat.clone()
    sd = trainer.optimizer_state_dict(tr)
    ref_opt = torch.optim.Adam(trainer.build_param_groups(model, args))
    ref_sd = ref_opt.state_dict()
    assert [len(g_["params"]) for g_ in sd["param_groups"]] == [len(g_["params"]) for g_ in ref_sd["param_groups"]] == [2, 9, 51, 56, 28]
    assert [g_["lr"] for g_ in sd["param_groups"]] == [g_["lr"] for g_ in ref_sd["param_groups"]]
    ref_opt.load_state_dict(sd)                                  # a stock optimiser accepts it ...
    flat_params = [p for g_ in ref_opt.param_groups for p in g_["params"]]
    for i, p in enumerate(flat_params):                          # ... and lands every moment on the right tensor
        st = ref_opt.state[p]
        assert st["exp_avg"].shape == p.shape and float(st["step"]) == 7.0
    path = str(tmp_path / "epoch-3.pt")
    trainer.save_checkpoint(path, model, tr)
    ck = torch.load(path, map_location="cpu", weights_only=False)
    assert set(ck) == {"model_state_dict", "optimizer", "rng_state", "cuda_rng_state"}
    m0, v0 = tr.m.clone(), tr.v.clone()
    tr.m.zero_(); tr.v.zero_(); tr.step_no = 0
    with torch.no_grad():
        for p in model.parameters():                             # (the alignment padding of the flat buffer stays zero)
            if p.requires_grad:
                p.add_(1.0)
    trainer.load_checkpoint(path, model, tr)
    assert torch.equal(tr.flat, before) and torch.equal(tr.m, m0) and torch.equal(tr.v, v0) and tr.step_no == 7
    for n, p in model.named_parameters():                        # still views of the flat buffer
        if p.requires_grad:
            assert p.data.untyped_storage().data_ptr() == tr.flat.untyped_storage().data_ptr(), n
