#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REAL reference.

Runs only in the build container (needs /root/reference and `transformers`); the GPU box never runs it and never
sees the reference.  It imports the reference's own `model` packages unmodified (Code_Uncached, Code_Cached) and
`data_utils.metrics.eval_model`, loads seeded weights produced by `iisan_amd.weights` into them, pushes seeded
inputs from `iisan_amd.synth` through, and stores INPUT CHECKSUMS + EXPECTED OUTPUTS only (no reference source).

  encoders_full.npz   HF ViT-B / BERT-B (repo config.json shapes, eager attention) -> CLS taps [4,13,768] x2
  sidenet_full.npz    Cached IISANAdaptedMModel + ModelMM at full width on synthetic taps: cv/text/mm, score,
                      prec, loss, gradients (small tensors whole, large ones strided), one Adam step; variants
  e2e_small.npz       Uncached ModelMM end to end with 2-layer ViT/BERT (hidden 768): loss + gradients
  e2e_bs8.npz         the same on 8 sequences (88 item slots)
  e2e_inter.npz       the same as e2e_small with --modality inter (mm tower only)
  eval.npz            data_utils.metrics.eval_model: Hit@10 / nDCG@10, per-user ranks and the first ten item ids of metrics_topK's own argsort
  adam_groups.json    name -> Adam group of the 146 trainable tensors under the rule of run.py:296-321

Usage:  python tests/golden/make_golden.py [--only NAME]
"""
from __future__ import annotations

import argparse
import hashlib
import importlib.util
import json
import os
import sys
import types
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"

from iisan_amd import synth, weights  # noqa: E402

GRAD_STRIDE = 97      # large gradient tensors are stored as flat[::GRAD_STRIDE]
GRAD_FULL_MAX = 5000  # tensors up to this many elements are stored whole


def sha(t: torch.Tensor) -> str:
    return hashlib.sha256(t.detach().cpu().contiguous().numpy().tobytes()).hexdigest()[:16]


def load_ref_pkg(variant: str, pkg: str):
    """Import /root/reference/<variant>/<pkg> as a uniquely named package (the three variants share names)."""
    name = f"ref_{variant}_{pkg}"
    path = os.path.join(REF, variant, pkg)
    spec = importlib.util.spec_from_file_location(name, os.path.join(path, "__init__.py"),
                                                  submodule_search_locations=[path])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def ref_args(**kw):
    a = dict(max_seq_len=10, l2_weight=0.1, embedding_dim=64, num_attention_heads=2, drop_rate=0.1,
             transformer_block=2, modality="intra_inter", CV_model_load="vit-base-mae", bert_model_load="bert_base_uncased",
             word_embedding_dim=768, num_words_title=30, num_words_abstract=0, num_words_body=0,
             news_attributes=["title"], remove_first="None", side_adapter_vit_list="1,3,5,7,9,11",
             side_adapter_bert_list="1,3,5,7,9,11", cv_adapter_down_size=64, bert_adapter_down_size=64,
             adapter_dropout_rate=0.1, adapter_activation="RELU", fusion_method="gated")
    a.update(kw)
    return SimpleNamespace(**a)


def hf_models(vcfg: weights.VitConfig, bcfg: weights.BertConfig, vw, bw):
    from transformers import BertConfig, BertModel, ViTConfig, ViTForImageClassification
    vc = ViTConfig.from_pretrained(os.path.join(REF, "pretrained_models/vit-base-patch16-224"))
    vc.hidden_size, vc.num_hidden_layers, vc.num_attention_heads = vcfg.hidden, vcfg.layers, vcfg.heads
    vc.intermediate_size, vc.image_size, vc.patch_size = vcfg.mlp, vcfg.image, vcfg.patch
    vc._attn_implementation = "eager"
    bc = BertConfig.from_pretrained(os.path.join(REF, "pretrained_models/bert/bert_base_uncased"),
                                    output_hidden_states=True)
    bc.hidden_size, bc.num_hidden_layers, bc.num_attention_heads = bcfg.hidden, bcfg.layers, bcfg.heads
    bc.intermediate_size, bc.vocab_size, bc.max_position_embeddings = bcfg.mlp, bcfg.vocab, bcfg.max_pos
    bc._attn_implementation = "eager"
    vit = ViTForImageClassification(vc)
    bert = BertModel(bc)
    vit.classifier = torch.nn.Linear(vcfg.hidden, 64)          # run.py:56-57
    missing, unexpected = vit.load_state_dict(weights.vit_to_hf5(vw, vcfg), strict=False)
    assert not unexpected and all("classifier" in m for m in missing), (missing, unexpected)
    missing, unexpected = bert.load_state_dict(weights.bert_to_hf(bw, bcfg), strict=False)
    assert not unexpected and all("pooler" in m for m in missing), (missing, unexpected)
    return vit.eval(), bert.eval()


def pack_grads(named_grads):
    out = {}
    for n, g in named_grads.items():
        g = g.detach().reshape(-1)
        out["g/" + n] = (g if g.numel() <= GRAD_FULL_MAX else g[::GRAD_STRIDE]).numpy().copy()
        out["gn/" + n] = np.array([float(g.double().norm()), float(g.double().sum())])
    return out


# ---------------------------------------------------------------------------------------------------------------
def gen_encoders_full():
    vcfg, bcfg = weights.VIT_BASE, weights.BERT_BASE
    vw, bw = weights.make_vit_weights(vcfg, seed=1234), weights.make_bert_weights(bcfg, seed=4321)
    vit, bert = hf_models(vcfg, bcfg, vw, bw)
    b = synth.scientific_batch(bs=1, seed=2024, seq_len=3, lengths=[3])       # 4 slots, first one padding
    ref = load_ref_pkg("Code_Uncached", "model")
    args = ref_args()
    cv_enc = ref.Vit_Encoder(vit)
    bert_enc = ref.Bert_Encoder(args, bert)
    with torch.no_grad():
        _, hs_cv = cv_enc(b.images)
        _, hs_tx = bert_enc(b.text)
    taps_cv = torch.stack([h[:, 0] for h in hs_cv], 1)
    taps_tx = torch.stack([h[:, 0] for h in hs_tx], 1)
    assert taps_cv.shape == (4, 13, 768) and taps_tx.shape == (4, 13, 768)
    np.savez_compressed(os.path.join(HERE, "encoders_full.npz"),
                        vit_seed=1234, bert_seed=4321, batch_seed=2024,
                        images_sha=sha(b.images), text=b.text.numpy(), ids=b.ids.numpy(),
                        w_sha=np.array([sha(vw["L11.fc2_w"]), sha(bw["word_emb"])]),
                        taps_cv=taps_cv.numpy(), taps_text=taps_tx.numpy(),
                        # a few full-token rows of intermediate hidden states for kernel-level debugging
                        vit_h1_img1=hs_cv[1][1].numpy(), bert_h1_item1=hs_tx[1][1].numpy())
    print("encoders_full: taps", taps_cv.abs().mean().item(), taps_tx.abs().mean().item())


# ---------------------------------------------------------------------------------------------------------------
class _FakeNet(torch.nn.Module):
    """Stand-in exposing only the attributes the Cached wrapper dereferences (`Code_Cached/model/model.py:261-262`)."""
    def __init__(self):
        super().__init__()
        self.classifier = torch.nn.Linear(768, 64)


def build_cached_ref(P, args, item_num, pop):
    ref = load_ref_pkg("Code_Cached", "model")
    from transformers import BertConfig, BertModel
    bc = BertConfig(hidden_size=32, num_hidden_layers=1, num_attention_heads=2, intermediate_size=32, vocab_size=16)
    model = ref.ModelMM(args, item_num, True, _FakeNet(), BertModel(bc), pop.numpy())
    model.mm_encoder = ref.IISANAdaptedMModel(model.mm_encoder, args)
    sd = {k: v for k, v in P.items()}
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert not [m for m in missing if m in P], missing
    return model.eval()     # eval: SASRec dropout off (deterministic parity)


def run_ref_loss(model, ids, img_in, txt_in, log_mask, names):
    for n, p in model.named_parameters():
        p.requires_grad_(n in names)
        p.grad = None
    loss = model(ids.view(-1), img_in, txt_in, log_mask, "cpu")
    loss.backward()
    # (modality "inter": the cv / text towers are outside the loss and keep grad None — they are left out of the fixture)
    grads = {n: p.grad.clone() for n, p in model.named_parameters() if n in names and p.grad is not None}
    return loss.detach(), grads


def gen_sidenet_full():
    out = {}
    bs, S = 3, 10
    b = synth.scientific_batch(bs=bs, seed=77, lengths=[4, 11, 7], dup_items=True, res=16, item_num=50)
    taps_cv = synth.cached_taps(b.ids, 12, 768, seed=5)
    taps_tx = synth.cached_taps(b.ids, 12, 768, seed=6)
    out.update(ids=b.ids.numpy(), log_mask=b.log_mask.numpy(), pop=b.pop_prob.numpy(),
               taps_sha=np.array([sha(taps_cv), sha(taps_tx)]))
    variants = {
        "default": dict(),
        "gelu": dict(adapter_activation="GELU"),
        "rmfirst": dict(remove_first="TRUE"),
    }
    for vname, kw in variants.items():
        # remove_first drops the layer-0 SANB: 6 blocks per tower (model.py:172-174)
        P = weights.make_trainable_params(seed=99, cached=True, n_side=6 if vname == "rmfirst" else 7)
        names = set(P)
        args = ref_args(**kw)
        model = build_cached_ref(P, args, 50, b.pop_prob)
        # intermediate outputs of the wrapper (eval form [M,13,768], `model.py:301-302` dim()==3 branch)
        with torch.no_grad():
            cv, (text, mm) = model.mm_encoder(taps_cv, taps_tx)
            score = model.com_dense(torch.cat([cv, text, mm], 1))
            prec = model.user_encoder(score.view(bs, S + 1, 64)[:, :-1], b.log_mask, "cpu").reshape(-1, 64)
        used = names
        loss, grads = run_ref_loss(model, b.ids, taps_cv.view(bs, S + 1, 13, 768), taps_tx.view(bs, S + 1, 13, 768),
                                   b.log_mask, used)
        pre = vname + "/"
        out.update({pre + "cv": cv.numpy(), pre + "text": text.numpy(), pre + "mm": mm.numpy(),
                    pre + "score": score.numpy(), pre + "prec": prec.numpy(), pre + "loss": loss.numpy()})
        out.update({pre + k: v for k, v in pack_grads(grads).items()})
        print(f"sidenet_full[{vname}]: loss {loss.item():.6f}")
        if vname == "default":
            # one Adam step with the five groups of run.py:323-336 (lr values of scripts/run_IISAN.py)
            lrs = dict(recsys=2e-4, adapter_cv=1e-4, adapter_text=1e-4, image_net=1e-4, text_encoder=5e-5)
            groups = {g: [] for g in lrs}
            nm = dict(model.named_parameters())
            rule = group_rule()
            for n in sorted(names):
                groups[rule(n)].append(nm[n])
            opt = torch.optim.Adam([{"params": ps, "lr": lrs[g]} for g, ps in groups.items()])
            before = {n: nm[n].detach().clone() for n in names}
            opt.step()
            out.update({pre + "adam/" + n: ((nm[n].detach() - before[n]).reshape(-1)[::GRAD_STRIDE]
                                             if nm[n].numel() > GRAD_FULL_MAX else (nm[n].detach() - before[n]).reshape(-1)).numpy()
                        for n in names})
    np.savez_compressed(os.path.join(HERE, "sidenet_full.npz"), **out)


def group_rule():
    """The grouping rule of `Code_Uncached/run.py:296-321`, executed here as the reference writes it."""
    def rule(name):
        if 'cv' in name:
            if ('fc' in name and "fc_" not in name) or 'classifier' in name or 'decoder_pred' in name:
                return "recsys"
            if "adapter" not in name and "lora" not in name:
                return "image_net"
            return "adapter_cv"
        elif "bert" in name:
            if 'fc' in name and "fc_" not in name:
                return "recsys"
            if "adapter" not in name and "lora" not in name:
                return "text_encoder"
            return "adapter_text"
        elif "mm_adapter" in name:
            return "adapter_cv"
        return "recsys"
    return rule


# ---------------------------------------------------------------------------------------------------------------
def gen_e2e_small(name="e2e_small", lengths=(3, 11, 6), write_groups=True, modality="intra_inter"):
    vcfg = weights.VitConfig(hidden=768, layers=2, heads=12, mlp=512, image=32, patch=16)
    bcfg = weights.BertConfig(hidden=768, layers=2, heads=12, mlp=512, vocab=512, max_pos=64)
    vw, bw = weights.make_vit_weights(vcfg, seed=11), weights.make_bert_weights(bcfg, seed=12)
    vit, bert = hf_models(vcfg, bcfg, vw, bw)
    bs, S, words = len(lengths), 10, 8
    b = synth.scientific_batch(bs=bs, seed=31, lengths=list(lengths), dup_items=True, res=32, words=words,
                               vocab=512, item_num=40)
    ref = load_ref_pkg("Code_Uncached", "model")
    args = ref_args(side_adapter_vit_list="0,1", side_adapter_bert_list="0,1", num_words_title=words, modality=modality)
    model = ref.ModelMM(args, 40, True, vit, bert, b.pop_prob.numpy())
    for p in model.parameters():
        p.requires_grad_(False)                                               # run.py:177-183
    model.mm_encoder = ref.IISANAdaptedMModel(model.mm_encoder, args)        # run.py:214-216
    P = weights.make_trainable_params(seed=101, n_side=3, modality=modality)
    missing, unexpected = model.load_state_dict(P, strict=False)
    assert not unexpected, unexpected
    model.eval()
    names = set(P)
    loss, grads = run_ref_loss(model, b.ids, b.images, b.text, b.log_mask, names)
    with torch.no_grad():
        cv, (text, mm) = model.mm_encoder(b.images, b.text)
    out = dict(ids=b.ids.numpy(), log_mask=b.log_mask.numpy(), text=b.text.numpy(), pop=b.pop_prob.numpy(),
               images_sha=sha(b.images), loss=loss.numpy(), mm=mm.numpy())
    if "intra" in modality:      # modality "inter": the wrapper returns the untouched 768-wide zero states for cv / text
        out.update(cv=cv.numpy(), text_emb=text.numpy())
    out.update(pack_grads(grads))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(f"{name}: loss {loss.item():.6f}")
    if not write_groups:
        return
    # Adam grouping of the real module tree (Uncached names)
    rule = group_rule()
    full = sorted(weights.trainable_shapes())        # the 146 Uncached names (7 SANBs per tower)
    json.dump({n: rule(n) for n in full}, open(os.path.join(HERE, "adam_groups.json"), "w"), indent=0)


# ---------------------------------------------------------------------------------------------------------------
def gen_eval():
    import transformers  # noqa: F401  (must be imported BEFORE the stubs below)
    for name in ("torchvision", "torchvision.transforms", "torchvision.models", "lmdb"):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__spec__ = importlib.machinery.ModuleSpec(name, None)
            sys.modules[name] = m
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    sys.modules["torchvision"].models = sys.modules["torchvision.models"]
    du = load_ref_pkg("Code_Uncached", "data_utils")
    import torch.distributed as dist
    if not dist.is_initialized():
        dist.init_process_group("gloo", init_method="tcp://127.0.0.1:29581", rank=0, world_size=1)
    item_num, n_users, E = 300, 50, 64
    rs = np.random.RandomState(5)
    P = weights.make_trainable_params(seed=99, cached=True)
    tables = [torch.from_numpy(rs.standard_normal((item_num + 1, E)).astype(np.float32)) for _ in range(3)]
    for t in tables:
        t[0] = 0
    seqs, hist = {}, {}
    for u in range(n_users):
        l = int(rs.randint(2, 12))
        seq = [int(x) for x in rs.choice(np.arange(1, item_num + 1), size=l, replace=False)]
        if u % 7 == 0 and l > 2:
            seq[-1] = seq[0]                       # repeat purchase: target inside its own history
        seqs[u] = seq
        hist[u] = torch.LongTensor(seq[:-1])
    ref = load_ref_pkg("Code_Cached", "model")
    args = ref_args(num_workers=0)
    ue = ref.User_Encoder(item_num, 10, E, 2, 0.1, 2)
    ue.load_state_dict({k[len("user_encoder."):]: v for k, v in P.items() if k.startswith("user_encoder.")})
    com = torch.nn.Linear(3 * E, E)
    com.load_state_dict({"weight": P["com_dense.weight"], "bias": P["com_dense.bias"]})
    holder = SimpleNamespace(module=SimpleNamespace(com_dense=com, user_encoder=ue.eval()), eval=lambda: None)

    class _Log:
        def info(self, *a, **k):
            pass

    captured, top10 = [], []
    orig = du.metrics.metrics_topK

    def spy(y_score, y_true, item_rank, topK, local_rank):
        # what metrics_topK computes at metrics.py:60 — `y_score` is the history-masked score row without column 0
        # (metrics.py:204-206), so position p of `order` is item id order[p] + 1: the user's recommendation list
        order = torch.argsort(y_score, descending=True)
        captured.append(int(torch.sum(torch.take(y_true, order) * item_rank).item()))
        top10.append((order[:10] + 1).numpy().astype(np.int32))
        return orig(y_score, y_true, item_rank, topK, local_rank)

    du.metrics.metrics_topK = spy
    hit = du.metrics.eval_model(holder, hist, seqs, tables[0], [tables[1], tables[2]], 16, args, item_num,
                                _Log(), "valid", "cpu")
    ranks = np.array(captured[:n_users], dtype=np.int64)
    hit10 = float((ranks <= 10).mean())
    ndcg10 = float(np.where(ranks <= 10, 1.0 / np.log2(ranks + 1.0), 0.0).mean())
    assert abs(hit - hit10) < 1e-6, (hit, hit10)
    np.savez_compressed(os.path.join(HERE, "eval.npz"), item_num=item_num, tables_seed=5,
                        table_cv=tables[0].numpy(), table_text=tables[1].numpy(), table_mm=tables[2].numpy(),
                        seq_flat=np.concatenate([np.array(seqs[u]) for u in range(n_users)]),
                        seq_len=np.array([len(seqs[u]) for u in range(n_users)]),
                        ranks=ranks, hit10=hit10, ndcg10=ndcg10, top10=np.stack(top10[:n_users]))
    print(f"eval: Hit@10 {hit10:.4f} nDCG@10 {ndcg10:.4f} (ranks min {ranks.min()} max {ranks.max()})")


VERSA_VARIANTS = {
    # name: (Di, Dt, image taps, text taps, vit list, bert list, extra args)
    "text_wide_long": (256, 512, 6, 10, "0,2,4", "1,3,5,7,8", {}),
    "image_wide_long": (512, 256, 7, 5, "0,1,2,3,5", "1,3", dict(adapter_activation="GELU")),
    "equal_rmfirst": (256, 256, 5, 5, "0,2,3", "1,2,3", dict(remove_first="TRUE")),
}


def gen_versa():
    """IISAN-Versa (Code_Cached_Asym): asymmetric towers, group layer-drop, dim-align."""
    ref = load_ref_pkg("Code_Cached_Asym", "model")
    from transformers import BertConfig, BertModel
    out = {}
    bs, S = 3, 10
    b = synth.scientific_batch(bs=bs, seed=78, lengths=[5, 11, 3], dup_items=True, res=16, item_num=50)
    out.update(ids=b.ids.numpy(), log_mask=b.log_mask.numpy(), pop=b.pop_prob.numpy())
    for vname, (Di, Dt, Lc, Lt, vlist, blist, extra) in VERSA_VARIANTS.items():
        args = ref_args(text_embedding_dim=Dt, image_embedding_dim=Di, side_adapter_vit_list=vlist, side_adapter_bert_list=blist,
                        image_layers=Lc - 1, text_layers=Lt - 1, **extra)
        bc = BertConfig(hidden_size=32, num_hidden_layers=1, num_attention_heads=2, intermediate_size=32, vocab_size=16)
        model = ref.ModelMM(args, 50, True, _FakeNet(), BertModel(bc), b.pop_prob.numpy())
        for p_ in model.parameters():
            p_.requires_grad_(False)
        model.mm_encoder = ref.IISANAdaptedMModel(model.mm_encoder, args)
        train_names = {n: tuple(p_.shape) for n, p_ in model.named_parameters()
                       if n.startswith("mm_encoder.") or n.startswith("user_encoder.") or n.startswith("com_dense.")}
        P = weights.fill_params_seeded(train_names, seed=555)
        missing, unexpected = model.load_state_dict(P, strict=False)
        assert not unexpected, unexpected
        model.eval()
        taps_cv = synth.cached_taps(b.ids, Lc - 1, Di, seed=15)
        taps_tx = synth.cached_taps(b.ids, Lt - 1, Dt, seed=16)
        for n, p_ in model.named_parameters():
            p_.requires_grad_(n in P)
            p_.grad = None
        with torch.no_grad():
            cv, (text, mm) = model.mm_encoder(taps_cv, taps_tx)
        loss = model(b.ids.view(-1), taps_cv.view(bs, S + 1, Lc, Di), taps_tx.view(bs, S + 1, Lt, Dt), b.log_mask, "cpu")
        loss.backward()
        grads = {n: (p_.grad.clone() if p_.grad is not None else None) for n, p_ in model.named_parameters() if n in P}
        pre = vname + "/"
        out.update({pre + "cv": cv.numpy(), pre + "text": text.numpy(), pre + "mm": mm.numpy(), pre + "loss": loss.detach().numpy(),
                    pre + "taps_sha": np.array([sha(taps_cv), sha(taps_tx)]),
                    pre + "names": np.array(sorted(P)), pre + "unused": np.array(sorted(n for n, g in grads.items() if g is None) or [""])})
        out.update({pre + k: v for k, v in pack_grads({n: g for n, g in grads.items() if g is not None}).items()})
        print(f"versa[{vname}]: loss {loss.item():.6f}, {len(P)} tensors, {sum(g is None for g in grads.values())} unused")
    np.savez_compressed(os.path.join(HERE, "versa.npz"), **out)


E2E_BS8_LENGTHS = (3, 11, 6, 4, 9, 5, 7, 11)


def gen_e2e_bs8():
    """The same end-to-end case on 8 sequences (88 item slots): more terms per gradient sum, so the reference's gradients
    can be held tighter than on the 3-sequence batch."""
    gen_e2e_small("e2e_bs8", E2E_BS8_LENGTHS, write_groups=False)


def gen_e2e_inter():
    """modality "inter" (only the inter-modal tower; model.py:38-39,70-72,182-205).  Runs in Code_Uncached only: the
    Cached wrapper's forward walks `bert_adapter_list`, which that modality does not create (Code_Cached/model/model.py:318)."""
    gen_e2e_small("e2e_inter", write_groups=False, modality="inter")


GENS = dict(versa=gen_versa, encoders_full=gen_encoders_full, sidenet_full=gen_sidenet_full, e2e_small=gen_e2e_small,
            e2e_bs8=gen_e2e_bs8, e2e_inter=gen_e2e_inter, eval=gen_eval)

if __name__ == "__main__":
    import importlib.machinery  # noqa: F401
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    a = ap.parse_args()
    torch.manual_seed(0)
    torch.set_num_threads(8)
    for k, fn in GENS.items():
        if a.only in (None, k):
            fn()
