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


def test_real_huggingface_modules_are_accepted_at_the_boundary():
    """The drop-in boundary as `Code_Uncached/run.py:50-100,161` uses it: a `ViTForImageClassification` (classifier
    re-created as Linear(768, 64), run.py:56-61) and a `BertModel(output_hidden_states=True)` built from the values of the
    reference's `pretrained_models/*/config.json` go straight into the product `ModelMM`.  The weight extraction the HIP
    encoders pack from (`model/encoders.py:_vit_canonical/_bert_canonical`) must return exactly the weights that were
    loaded — for the installed transformers' key layout (5.x: `vit.layers.N.attention.q_proj`) and for the reference's
    pinned 4.20.1 layout (`vit.encoder.layer.N.attention.attention.query`)."""
    import pytest
    tf = pytest.importorskip("transformers")
    from torch import nn
    from iisan_amd.model import ModelMM
    from iisan_amd.model.encoders import _bert_canonical, _vit_canonical
    vcfg, bcfg = weights.VIT_BASE, weights.BERT_BASE
    vw, bw = weights.make_vit_weights(), weights.make_bert_weights()
    hf_v = tf.ViTConfig(hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072, image_size=224,
                        patch_size=16, num_channels=3, qkv_bias=True, layer_norm_eps=1e-12, hidden_act="gelu",
                        hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    hf_b = tf.BertConfig(hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
                         vocab_size=30522, max_position_embeddings=512, type_vocab_size=2, layer_norm_eps=1e-12,
                         hidden_act="gelu", output_hidden_states=True)
    vit = tf.ViTForImageClassification(hf_v)
    missing, unexpected = vit.load_state_dict(weights.vit_to_hf5(vw, vcfg), strict=False)
    assert not unexpected and all(k.startswith("classifier") for k in missing), (missing, unexpected)
    vit.classifier = nn.Linear(vit.classifier.in_features, 64)                       # run.py:59-60
    bert = tf.BertModel(hf_b)
    missing, unexpected = bert.load_state_dict(weights.bert_to_hf(bw, bcfg), strict=False)
    assert not unexpected and all("pooler" in k or "position_ids" in k for k in missing), (missing, unexpected)

    args = helpers.make_args()
    model = ModelMM(args, 100, True, vit, bert, torch.ones(101))
    trainer.apply_iisan_freeze_rules(model, args, cached=False)
    # same trainable set and sizes as with the light containers — and as the reference (199,394,773
    # parameters of which 4,113,877 trainable; the pooler is there and frozen, run.py:83-100)
    names = {n for n, p in model.named_parameters() if p.requires_grad}
    assert names == set(weights.trainable_shapes())
    assert sum(p.numel() for p in model.parameters() if p.requires_grad) == 4113877
    assert sum(p.numel() for p in model.parameters()) == 199394773
    assert "mm_encoder.cv_encoder.image_net.vit.embeddings.cls_token" in model.state_dict()
    assert "mm_encoder.bert_encoder.text_encoders.title.bert_model.embeddings.word_embeddings.weight" in model.state_dict()

    gv, gcfg = _vit_canonical(model.mm_encoder.cv_encoder.image_net)
    assert gcfg == vcfg and set(gv) == set(vw) and all(torch.equal(gv[k], vw[k]) for k in vw)
    gb, gbcfg = _bert_canonical(model.mm_encoder.bert_encoder.text_encoders["title"].bert_model)
    assert gbcfg == bcfg and set(gb) == set(bw) and all(torch.equal(gb[k], bw[k]) for k in bw)

    class Vit4x(nn.Module):                       # the same module as transformers 4.20.1 names its tensors
        def __init__(self, inner):
            super().__init__()
            self.inner, self.config = inner, inner.config

        def state_dict(self, *a, **k):
            ren = (("vit.layers.", "vit.encoder.layer."), ("attention.q_proj", "attention.attention.query"),
                   ("attention.k_proj", "attention.attention.key"), ("attention.v_proj", "attention.attention.value"),
                   ("attention.o_proj", "attention.output.dense"), ("mlp.fc1", "intermediate.dense"), ("mlp.fc2", "output.dense"))
            out = {}
            for key, v in self.inner.state_dict().items():
                for a_, b_ in ren:
                    key = key.replace(a_, b_)
                out[key] = v
            return out

    is5 = any(k.startswith("vit.layers.") for k in vit.state_dict())
    g4, _ = _vit_canonical(Vit4x(vit) if is5 else vit)
    assert any(".encoder.layer." in k for k in (Vit4x(vit) if is5 else vit).state_dict())
    assert all(torch.equal(g4[k], vw[k]) for k in vw)


def test_bench_starts_its_own_ranks_and_describes_the_host():
    """`python bench.py --gpus N` outside a rank environment must launch N fresh ranks itself (it used to
    exit non-zero), with the rendezvous on 127.0.0.1 and its own arguments passed through."""
    import importlib.util
    import sys
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(gio.GOLDEN.rstrip("/")), "..", "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    a = bench.parse(["--gpus", "8", "--steps", "4", "--warmup", "1"])
    cmd = bench.launch_command(a, ["--gpus", "8", "--steps", "4", "--warmup", "1"], 29123)
    assert cmd[0] == sys.executable and cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nproc-per-node=8" in cmd and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("--master-port") + 1] == "29123"
    i = cmd.index(os.path.abspath(bench.__file__))
    assert cmd[i + 1:] == ["--gpus", "8", "--steps", "4", "--warmup", "1"]
    assert bench.parse([]).bs == 128 and bench.parse(["--cached", "fp32"]).bs == 1024 and bench.parse(["--cached", "fp16", "--versa"]).bs == 128
    cores, model = bench.host_cpu()
    assert 1 <= cores <= (os.cpu_count() or 1) and isinstance(model, str) and model


def test_hit_ndcg_never_counts_the_invalid_target_signal_as_a_hit():
    """`iisan_score_rank` answers -1 for a target outside 1..item_num (the reference raises IndexError, metrics.py:206);
    `hit_ndcg` must not read that as rank <= 10."""
    from iisan_amd import evaluate
    hit, ndcg = evaluate.hit_ndcg(torch.tensor([1, 3, 11, -1]))
    assert abs(hit - 2 / 4) < 1e-12
    assert abs(ndcg - (1.0 + 0.5) / 4) < 1e-12


def test_bench_refuses_an_rccl_line_with_two_ranks_on_one_device():
    """the first 8-GPU SCALE run must verify itself — a multi-rank line under backend nccl whose ranks do
    not each own a device is refused (gloo with both ranks on cuda:0 is the single-GPU test transport and passes)."""
    import importlib.util
    import pytest
    spec = importlib.util.spec_from_file_location("bench_mod2", os.path.join(os.path.dirname(gio.GOLDEN.rstrip("/")), "..", "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    r = lambda dev, uuid, host="n0": {"rank": dev, "host": host, "device": dev, "uuid": uuid}
    bench.check_one_device_per_rank("nccl", [r(0, "GPU-a"), r(1, "GPU-b")])
    bench.check_one_device_per_rank("nccl", [r(0, ""), r(1, "")])                      # no uuid reported: device indices differ
    bench.check_one_device_per_rank("nccl", [r(0, "00000000"), r(1, "00000000")])      # a runtime with degenerate uuids
    bench.check_one_device_per_rank("nccl", [r(0, "GPU-a", "n0"), r(0, "GPU-a", "n1")])  # same index on two hosts
    bench.check_one_device_per_rank("gloo", [r(0, "GPU-a"), r(0, "GPU-a")])
    with pytest.raises(SystemExit):
        bench.check_one_device_per_rank("nccl", [r(0, "GPU-a"), r(0, "GPU-a")])
    with pytest.raises(SystemExit):
        bench.check_one_device_per_rank("nccl", [r(0, ""), r(0, "")])


def test_h256_head_major_addressing_bound_is_independent_of_which0():
    """`gemm16_h256_kernel` stores head-major QKV through 32-bit byte offsets into [item][head][3][S][64]; the
    element-row index reaches rows * 3 * heads whatever `qkv_which0` is, so the applicability bound must too (it used 3 - which0:
    row counts in [932k, 1.40M) passed with which0 = 1 and the offsets wrapped).  Pure host logic: no device is touched."""
    from iisan_amd import _lib
    lib = _lib.load()
    QKV = 4                                       # EPI_QKVH16
    heads, S = 12, 197
    limit = (1 << 32) // (3 * heads * 128)        # rows beyond this cannot be addressed with 32 bits
    for which0, N in ((0, 2304), (1, 1536)):
        assert lib.iisan_gemm16_h256_applicable(QKV, 277376, N, 768, S, heads, which0) == 1          # the production shape
        assert lib.iisan_gemm16_h256_applicable(QKV, 256 * (limit // 256 - 1), N, 768, S, heads, which0) == 1
        for rows in (940_000, 1_200_000, 1_390_000, 256 * (limit // 256 + 1)):
            assert lib.iisan_gemm16_h256_applicable(QKV, rows, N, 768, S, heads, which0) == 0, (which0, rows)
    # the 16-bit row-major epilogues keep their own (leading-dimension) bound
    assert lib.iisan_gemm16_h256_applicable(0, 1_390_000, 768, 768, 0, 0, 0) == 1
    assert lib.iisan_gemm16_h256_applicable(0, 277376, 768, 64, 0, 0, 0) == 0                       # K / 64 < 2


def test_tools_compile_and_name_only_declared_entry_points():
    """a development tool that still iterated a table the ABI clean-up had removed died silently in every child process.  Every
    script under tools/ must byte-compile, and every `iisan_*` entry point it names must be one the loader binds (`_lib.SIGNATURES` =
    the symbols include/iisan_hip.h declares); attributes of `_lib` it touches must exist."""
    import glob
    import py_compile
    import re
    from iisan_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    retired = {"iisan_set_x3"}                       # named in prose (error messages / docstrings) only
    for path in sorted(glob.glob(os.path.join(root, "tools", "*.py"))):
        py_compile.compile(path, doraise=True)
        src = open(path).read()
        for name in set(re.findall(r"\biisan_[a-z0-9_]+\b", src)):
            if name in ("iisan_amd", "iisan_hip", "iisan_oracle") or name in retired:
                continue
            assert name in _lib.SIGNATURES, f"{os.path.basename(path)} names {name}, which the library does not export"
        for attr in set(re.findall(r"\b_lib\.([A-Za-z_]+)\b", src)):
            assert hasattr(_lib, attr), f"{os.path.basename(path)} uses _lib.{attr}, which does not exist"
