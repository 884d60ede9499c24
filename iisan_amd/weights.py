"""Frozen-encoder weight containers for the IISAN hot path.

The reference never owns ViT/BERT weights itself: it loads HuggingFace checkpoints
(`Code_Uncached/run.py:50-100`) and only calls the two models' forward
(`Code_Uncached/model/encoders.py:30,86`).  The checkpoints are not shipped and there is no
network, so this module provides

* a *canonical* flat naming of the encoder tensors (what the HIP path packs into HBM),
* converters from both HuggingFace state-dict layouts (4.x `vit.encoder.layer.N.attention.attention.query`
  and 5.x `vit.layers.N.attention.q_proj`) into the canonical naming and back (5.x),
* a seeded generator (`numpy.random.RandomState`, whose stream is frozen by NumPy's compatibility
  policy) used by bench.py, the tests and the golden-fixture script so that the GPU box can rebuild the
  exact weights a fixture was generated with, without `transformers` and without shipping 800 MB.

Canonical names
---------------
ViT  : cls_token[D] pos_emb[T,D] patch_w[D,3*P*P] patch_b[D] lnf_w[D] lnf_b[D]
       L{l}.ln1_w L{l}.ln1_b L{l}.qkv_w[3D,D] L{l}.qkv_b[3D] L{l}.o_w[D,D] L{l}.o_b[D]
       L{l}.ln2_w L{l}.ln2_b L{l}.fc1_w[F,D] L{l}.fc1_b[F] L{l}.fc2_w[D,F] L{l}.fc2_b[D]
BERT : word_emb[V,D] pos_emb[P,D] type_emb[2,D] emb_ln_w emb_ln_b
       L{l}.qkv_w L{l}.qkv_b L{l}.o_w L{l}.o_b L{l}.ln1_w L{l}.ln1_b   (post-attention LayerNorm)
       L{l}.fc1_w L{l}.fc1_b L{l}.fc2_w L{l}.fc2_b L{l}.ln2_w L{l}.ln2_b (post-FFN LayerNorm)
"""
from __future__ import annotations

from dataclasses import dataclass, asdict
from typing import Dict

import math

import numpy as np
import torch


@dataclass(frozen=True)
class VitConfig:
    """Shape of the image tower (`pretrained_models/vit-base-patch16-224/config.json`)."""
    hidden: int = 768
    layers: int = 12
    heads: int = 12
    mlp: int = 3072
    image: int = 224
    patch: int = 16
    channels: int = 3
    eps: float = 1e-12

    @property
    def tokens(self) -> int:
        return (self.image // self.patch) ** 2 + 1

    @property
    def patch_dim(self) -> int:
        return self.channels * self.patch * self.patch


@dataclass(frozen=True)
class BertConfig:
    """Shape of the text tower (`pretrained_models/bert/bert_base_uncased/config.json`)."""
    hidden: int = 768
    layers: int = 12
    heads: int = 12
    mlp: int = 3072
    vocab: int = 30522
    max_pos: int = 512
    eps: float = 1e-12


VIT_BASE = VitConfig()
BERT_BASE = BertConfig()


def _normal(rs: np.random.RandomState, shape, std: float) -> torch.Tensor:
    return torch.from_numpy((rs.standard_normal(size=shape) * std).astype(np.float32))


def make_vit_weights(cfg: VitConfig = VIT_BASE, seed: int = 1234, std: float = 0.02,
                     ln_jitter: float = 0.05) -> Dict[str, torch.Tensor]:
    """Seeded random ViT weights (fp32, CPU).  LayerNorm gains/biases and Linear biases are jittered away
    from their (1, 0, 0) defaults so that a kernel dropping one of them cannot pass a parity test."""
    rs = np.random.RandomState(seed)
    D, F = cfg.hidden, cfg.mlp
    w = {
        "cls_token": _normal(rs, (D,), std),
        "pos_emb": _normal(rs, (cfg.tokens, D), std),
        "patch_w": _normal(rs, (D, cfg.patch_dim), std),
        "patch_b": _normal(rs, (D,), std),
        "lnf_w": 1.0 + _normal(rs, (D,), ln_jitter),
        "lnf_b": _normal(rs, (D,), ln_jitter),
    }
    for l in range(cfg.layers):
        p = f"L{l}."
        w[p + "ln1_w"] = 1.0 + _normal(rs, (D,), ln_jitter)
        w[p + "ln1_b"] = _normal(rs, (D,), ln_jitter)
        w[p + "qkv_w"] = _normal(rs, (3 * D, D), std)
        w[p + "qkv_b"] = _normal(rs, (3 * D,), std)
        w[p + "o_w"] = _normal(rs, (D, D), std)
        w[p + "o_b"] = _normal(rs, (D,), std)
        w[p + "ln2_w"] = 1.0 + _normal(rs, (D,), ln_jitter)
        w[p + "ln2_b"] = _normal(rs, (D,), ln_jitter)
        w[p + "fc1_w"] = _normal(rs, (F, D), std)
        w[p + "fc1_b"] = _normal(rs, (F,), std)
        w[p + "fc2_w"] = _normal(rs, (D, F), std)
        w[p + "fc2_b"] = _normal(rs, (D,), std)
    return w


def make_bert_weights(cfg: BertConfig = BERT_BASE, seed: int = 4321, std: float = 0.02,
                      ln_jitter: float = 0.05) -> Dict[str, torch.Tensor]:
    rs = np.random.RandomState(seed)
    D, F = cfg.hidden, cfg.mlp
    w = {
        "word_emb": _normal(rs, (cfg.vocab, D), std),
        "pos_emb": _normal(rs, (cfg.max_pos, D), std),
        "type_emb": _normal(rs, (2, D), std),
        "emb_ln_w": 1.0 + _normal(rs, (D,), ln_jitter),
        "emb_ln_b": _normal(rs, (D,), ln_jitter),
    }
    for l in range(cfg.layers):
        p = f"L{l}."
        w[p + "qkv_w"] = _normal(rs, (3 * D, D), std)
        w[p + "qkv_b"] = _normal(rs, (3 * D,), std)
        w[p + "o_w"] = _normal(rs, (D, D), std)
        w[p + "o_b"] = _normal(rs, (D,), std)
        w[p + "ln1_w"] = 1.0 + _normal(rs, (D,), ln_jitter)
        w[p + "ln1_b"] = _normal(rs, (D,), ln_jitter)
        w[p + "fc1_w"] = _normal(rs, (F, D), std)
        w[p + "fc1_b"] = _normal(rs, (F,), std)
        w[p + "fc2_w"] = _normal(rs, (D, F), std)
        w[p + "fc2_b"] = _normal(rs, (D,), std)
        w[p + "ln2_w"] = 1.0 + _normal(rs, (D,), ln_jitter)
        w[p + "ln2_b"] = _normal(rs, (D,), ln_jitter)
    return w


# ----------------------------------------------------------------------------------------------------------------
# HuggingFace state-dict <-> canonical
# ----------------------------------------------------------------------------------------------------------------

def _strip_prefix(sd: Dict[str, torch.Tensor], prefixes) -> Dict[str, torch.Tensor]:
    out = {}
    for k, v in sd.items():
        for p in prefixes:
            if k.startswith(p):
                k = k[len(p):]
                break
        out[k] = v
    return out


def vit_from_hf(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Canonical ViT weights from a `ViTForImageClassification`/`ViTModel` state dict (either HF layout).
    The classifier head is NOT part of the frozen encoder (it is the trainable `classifier` the reference
    re-creates at `Code_Uncached/run.py:57-61`) and is ignored here."""
    sd = _strip_prefix(sd, ("vit.",))
    w = {
        "cls_token": sd["embeddings.cls_token"].reshape(-1),
        "pos_emb": sd["embeddings.position_embeddings"].reshape(-1, sd["embeddings.position_embeddings"].shape[-1]),
        "patch_w": sd["embeddings.patch_embeddings.projection.weight"].reshape(
            sd["embeddings.patch_embeddings.projection.weight"].shape[0], -1),
        "patch_b": sd["embeddings.patch_embeddings.projection.bias"],
        "lnf_w": sd["layernorm.weight"],
        "lnf_b": sd["layernorm.bias"],
    }
    new = any(k.startswith("layers.") for k in sd)
    l = 0
    while True:
        if new:
            b = f"layers.{l}."
            names = dict(q=b + "attention.q_proj", k=b + "attention.k_proj", v=b + "attention.v_proj",
                         o=b + "attention.o_proj", fc1=b + "mlp.fc1", fc2=b + "mlp.fc2")
        else:
            b = f"encoder.layer.{l}."
            names = dict(q=b + "attention.attention.query", k=b + "attention.attention.key",
                         v=b + "attention.attention.value", o=b + "attention.output.dense",
                         fc1=b + "intermediate.dense", fc2=b + "output.dense")
        if names["q"] + ".weight" not in sd:
            break
        p = f"L{l}."
        w[p + "ln1_w"] = sd[b + "layernorm_before.weight"]
        w[p + "ln1_b"] = sd[b + "layernorm_before.bias"]
        w[p + "ln2_w"] = sd[b + "layernorm_after.weight"]
        w[p + "ln2_b"] = sd[b + "layernorm_after.bias"]
        w[p + "qkv_w"] = torch.cat([sd[names[x] + ".weight"] for x in "qkv"], 0)
        w[p + "qkv_b"] = torch.cat([sd[names[x] + ".bias"] for x in "qkv"], 0)
        for x in ("o", "fc1", "fc2"):
            w[p + x + "_w"] = sd[names[x] + ".weight"]
            w[p + x + "_b"] = sd[names[x] + ".bias"]
        l += 1
    return {k: v.detach().float().contiguous() for k, v in w.items()}


def bert_from_hf(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Canonical BERT weights from a `BertModel` state dict (`bert.` prefix tolerated; pooler ignored —
    the reference freezes it and IISAN never reads it, `Code_Uncached/run.py:83-100`)."""
    sd = _strip_prefix(sd, ("bert.",))
    w = {
        "word_emb": sd["embeddings.word_embeddings.weight"],
        "pos_emb": sd["embeddings.position_embeddings.weight"],
        "type_emb": sd["embeddings.token_type_embeddings.weight"],
        "emb_ln_w": sd["embeddings.LayerNorm.weight"],
        "emb_ln_b": sd["embeddings.LayerNorm.bias"],
    }
    l = 0
    while f"encoder.layer.{l}.attention.self.query.weight" in sd:
        b = f"encoder.layer.{l}."
        p = f"L{l}."
        w[p + "qkv_w"] = torch.cat([sd[b + f"attention.self.{x}.weight"] for x in ("query", "key", "value")], 0)
        w[p + "qkv_b"] = torch.cat([sd[b + f"attention.self.{x}.bias"] for x in ("query", "key", "value")], 0)
        w[p + "o_w"] = sd[b + "attention.output.dense.weight"]
        w[p + "o_b"] = sd[b + "attention.output.dense.bias"]
        w[p + "ln1_w"] = sd[b + "attention.output.LayerNorm.weight"]
        w[p + "ln1_b"] = sd[b + "attention.output.LayerNorm.bias"]
        w[p + "fc1_w"] = sd[b + "intermediate.dense.weight"]
        w[p + "fc1_b"] = sd[b + "intermediate.dense.bias"]
        w[p + "fc2_w"] = sd[b + "output.dense.weight"]
        w[p + "fc2_b"] = sd[b + "output.dense.bias"]
        w[p + "ln2_w"] = sd[b + "output.LayerNorm.weight"]
        w[p + "ln2_b"] = sd[b + "output.LayerNorm.bias"]
        l += 1
    return {k: v.detach().float().contiguous() for k, v in w.items()}


def vit_to_hf5(w: Dict[str, torch.Tensor], cfg: VitConfig) -> Dict[str, torch.Tensor]:
    """Canonical -> `ViTForImageClassification` state dict in the transformers-5.x layout (classifier excluded).
    Used only by the golden-fixture script to load generated weights into the imported reference."""
    D = cfg.hidden
    sd = {
        "vit.embeddings.cls_token": w["cls_token"].reshape(1, 1, D),
        "vit.embeddings.position_embeddings": w["pos_emb"].reshape(1, cfg.tokens, D),
        "vit.embeddings.patch_embeddings.projection.weight": w["patch_w"].reshape(D, cfg.channels, cfg.patch, cfg.patch),
        "vit.embeddings.patch_embeddings.projection.bias": w["patch_b"],
        "vit.layernorm.weight": w["lnf_w"],
        "vit.layernorm.bias": w["lnf_b"],
    }
    for l in range(cfg.layers):
        p, b = f"L{l}.", f"vit.layers.{l}."
        q, k, v = w[p + "qkv_w"].split(D, 0)
        qb, kb, vb = w[p + "qkv_b"].split(D, 0)
        sd[b + "attention.q_proj.weight"], sd[b + "attention.q_proj.bias"] = q, qb
        sd[b + "attention.k_proj.weight"], sd[b + "attention.k_proj.bias"] = k, kb
        sd[b + "attention.v_proj.weight"], sd[b + "attention.v_proj.bias"] = v, vb
        sd[b + "attention.o_proj.weight"], sd[b + "attention.o_proj.bias"] = w[p + "o_w"], w[p + "o_b"]
        sd[b + "layernorm_before.weight"], sd[b + "layernorm_before.bias"] = w[p + "ln1_w"], w[p + "ln1_b"]
        sd[b + "layernorm_after.weight"], sd[b + "layernorm_after.bias"] = w[p + "ln2_w"], w[p + "ln2_b"]
        sd[b + "mlp.fc1.weight"], sd[b + "mlp.fc1.bias"] = w[p + "fc1_w"], w[p + "fc1_b"]
        sd[b + "mlp.fc2.weight"], sd[b + "mlp.fc2.bias"] = w[p + "fc2_w"], w[p + "fc2_b"]
    return sd


def bert_to_hf(w: Dict[str, torch.Tensor], cfg: BertConfig) -> Dict[str, torch.Tensor]:
    """Canonical -> `BertModel` state dict (pooler excluded)."""
    D = cfg.hidden
    sd = {
        "embeddings.word_embeddings.weight": w["word_emb"],
        "embeddings.position_embeddings.weight": w["pos_emb"],
        "embeddings.token_type_embeddings.weight": w["type_emb"],
        "embeddings.LayerNorm.weight": w["emb_ln_w"],
        "embeddings.LayerNorm.bias": w["emb_ln_b"],
    }
    for l in range(cfg.layers):
        p, b = f"L{l}.", f"encoder.layer.{l}."
        q, k, v = w[p + "qkv_w"].split(D, 0)
        qb, kb, vb = w[p + "qkv_b"].split(D, 0)
        for n, (ww, bb) in dict(query=(q, qb), key=(k, kb), value=(v, vb)).items():
            sd[b + f"attention.self.{n}.weight"], sd[b + f"attention.self.{n}.bias"] = ww, bb
        sd[b + "attention.output.dense.weight"], sd[b + "attention.output.dense.bias"] = w[p + "o_w"], w[p + "o_b"]
        sd[b + "attention.output.LayerNorm.weight"], sd[b + "attention.output.LayerNorm.bias"] = w[p + "ln1_w"], w[p + "ln1_b"]
        sd[b + "intermediate.dense.weight"], sd[b + "intermediate.dense.bias"] = w[p + "fc1_w"], w[p + "fc1_b"]
        sd[b + "output.dense.weight"], sd[b + "output.dense.bias"] = w[p + "fc2_w"], w[p + "fc2_b"]
        sd[b + "output.LayerNorm.weight"], sd[b + "output.LayerNorm.bias"] = w[p + "ln2_w"], w[p + "ln2_b"]
    return sd


def config_dict(cfg) -> dict:
    return asdict(cfg)


# ----------------------------------------------------------------------------------------------------------------
# seeded trainable tensors (side network + SASRec + heads), keyed by the reference's state-dict names
# ----------------------------------------------------------------------------------------------------------------

def trainable_shapes(n_side: int = 7, dim_cv: int = 768, dim_text: int = 768, down: int = 64, emb: int = 64,
                     seq_len: int = 10, n_blocks: int = 2, cached: bool = False, modality: str = "intra_inter") -> Dict[str, tuple]:
    """Shapes of the 146 trainable tensors of IISAN mode in `named_parameters()` order-independent form
    (`Code_Uncached/model/model.py:166-205`, `modules.py:35-116`, `model.py:36-37`)."""
    ue = "user_encoder.transformer_encoder."
    s = {ue + "position_embedding.weight": (seq_len, emb),
         ue + "layer_norm.weight": (emb,), ue + "layer_norm.bias": (emb,)}
    for b in range(n_blocks):
        a = ue + f"transformer_blocks.{b}.multi_head_attention."
        f = ue + f"transformer_blocks.{b}.feed_forward."
        for n in ("w_Q", "w_K", "w_V", "fc"):
            s[a + n + ".weight"] = (emb, emb)
        s[a + "layer_norm.weight"] = (emb,)
        s[a + "layer_norm.bias"] = (emb,)
        s[f + "w_1.weight"] = (4 * emb, emb)
        s[f + "w_1.bias"] = (4 * emb,)
        s[f + "w_2.weight"] = (emb, 4 * emb)
        s[f + "w_2.bias"] = (emb,)
        s[f + "layer_norm.weight"] = (emb,)
        s[f + "layer_norm.bias"] = (emb,)
    s["com_dense.weight"] = (emb, 3 * emb if "intra_inter" in modality else emb)       # model.py:36-41 ("inter": the mm tower only)
    s["com_dense.bias"] = (emb,)
    m = "mm_encoder."
    cvh = m + ("cv_pre_fc." if cached else "cv_encoder.image_net.classifier.")
    txh = m + ("bert_pre_fc." if cached else "bert_encoder.text_encoders.title.fc.")
    s[cvh + "weight"], s[cvh + "bias"] = (emb, dim_cv), (emb,)
    s[txh + "weight"], s[txh + "bias"] = (emb, dim_text), (emb,)
    intra = "intra" in modality                     # model.py:178-205: the cv / text towers exist only with "intra"
    for tower, d in (("cv", dim_cv), ("bert", dim_text), ("mm", dim_text)):
        if tower != "mm" and not intra:
            continue
        for k in range(n_side):
            p = m + f"{tower}_adapter_list.{k}."
            s[p + "fc_down.weight"], s[p + "fc_down.bias"] = (down, d), (down,)
            s[p + "fc_up.weight"], s[p + "fc_up.bias"] = (d, down), (d,)
    if intra:
        s[m + "fc_bert.weight"], s[m + "fc_bert.bias"] = (dim_text, dim_text), (dim_text,)
        s[m + "fc_cv.weight"], s[m + "fc_cv.bias"] = (dim_cv, dim_cv), (dim_cv,)
    s[m + "fc_mm.weight"], s[m + "fc_mm.bias"] = (dim_text, dim_text), (dim_text,)
    s[m + "fc_mm_down.weight"], s[m + "fc_mm_down.bias"] = (emb, dim_text), (emb,)
    for g in ("text", "cv", "mm"):
        if g != "mm" and not intra:
            continue
        for k in range(n_side):
            s[m + f"side_gate_params_{g}.{k}"] = (1,)
    return s


def make_trainable_params(seed: int = 99, **shape_kw) -> Dict[str, torch.Tensor]:
    """Seeded, deliberately NON-default values for every trainable tensor (biases and gates away from zero,
    LayerNorm gains away from one, adapter weights at std 0.05 so the bottleneck path is not negligible):
    parity tests on these catch a dropped bias/gate that the reference initialisers (zeros) would hide.
    Product modules initialise like the reference (`modules.py:101-110`, `encoders.py:52-58`); this generator is
    for tests, fixtures and the benchmark only."""
    rs = np.random.RandomState(seed)
    out = {}
    for name, shape in trainable_shapes(**shape_kw).items():
        if "side_gate_params" in name:
            t = _normal(rs, shape, 0.08)
        elif name.endswith("layer_norm.weight"):
            t = 1.0 + _normal(rs, shape, 0.1)
        elif name.endswith("bias"):
            t = _normal(rs, shape, 0.05)
        elif "adapter_list" in name:
            t = _normal(rs, shape, 0.05)
        elif "position_embedding" in name:
            t = _normal(rs, shape, 0.2)
        else:
            fan_out, fan_in = shape[0], shape[-1]
            t = _normal(rs, shape, math.sqrt(2.0 / (fan_in + fan_out)))
        out[name] = t
    return out


def fill_params_seeded(named_shapes: Dict[str, tuple], seed: int) -> Dict[str, torch.Tensor]:
    """Seeded non-default values for an arbitrary set of named tensors (sorted by name), with the same value rules as
    `make_trainable_params`.  Used for the Versa fixtures, whose tensor set depends on the depth/width configuration."""
    rs = np.random.RandomState(seed)
    out = {}
    for name in sorted(named_shapes):
        shape = tuple(named_shapes[name])
        if "side_gate_params" in name:
            t = _normal(rs, shape, 0.08)
        elif name.endswith("layer_norm.weight"):
            t = 1.0 + _normal(rs, shape, 0.1)
        elif name.endswith("bias"):
            t = _normal(rs, shape, 0.05)
        elif "adapter_list" in name:
            t = _normal(rs, shape, 0.05)
        elif "position_embedding" in name:
            t = _normal(rs, shape, 0.2)
        else:
            t = _normal(rs, shape, math.sqrt(2.0 / (shape[0] + shape[-1])))
        out[name] = t
    return out
