"""CPU restatement of the IISAN hot path — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this module; the product
(`iisan_amd/`) never does and fails loudly when its HIP extension is missing.

Plain PyTorch-CPU fp32 (the path is floating point, so a torch fp32 statement is the right oracle; the integer
parts — masks, labels, ranks — are plain index arithmetic).  Every function restates one row of SURVEY.md §8(a)
and cites the reference lines it follows.  It is written from the maths in SURVEY.md Appendix A, functional
(weights passed as dicts), and is differentiable with autograd so the same code checks the HIP backward.

Parity pin: `tests/golden/make_golden.py` imports the real reference (`/root/reference/Code_*/model`, HuggingFace
ViT/BERT built from the repo's config.json) in the build container and stores its outputs on seeded inputs under
`tests/golden/*.npz`; `tests/test_oracle_vs_golden.py` checks this file against those vectors.  The reference has
no tests of its own for this path, so those generated vectors are the pin.

Weights naming: encoders use the canonical names of `iisan_amd/weights.py`; trainable tensors use the reference's
state-dict keys relative to `ModelMM` (e.g. `mm_encoder.cv_adapter_list.0.fc_down.weight`).
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# ---------------------------------------------------------------------------------------------------------------
# U1  frozen ViT with per-layer hidden states      reference: Code_Uncached/model/encoders.py:29-31 -> HF ViT
# ---------------------------------------------------------------------------------------------------------------

def _mha(x: Tensor, qkv_w: Tensor, qkv_b: Tensor, heads: int, key_bias: Tensor | None) -> Tensor:
    """softmax(q k^T / sqrt(d) + key_bias) v for [B,S,D] tokens; returns the heads re-concatenated [B,S,D]."""
    B, S, D = x.shape
    d = D // heads
    qkv = F.linear(x, qkv_w, qkv_b).view(B, S, 3, heads, d)
    q, k, v = (qkv[:, :, i].transpose(1, 2) for i in range(3))          # [B,H,S,d]
    s = torch.matmul(q, k.transpose(-1, -2)) * (d ** -0.5)
    if key_bias is not None:
        s = s + key_bias[:, None, None, :]
    p = torch.softmax(s, dim=-1)
    return torch.matmul(p, v).transpose(1, 2).reshape(B, S, D)


def vit_patches(images: Tensor, patch: int) -> Tensor:
    """[B,C,H,W] -> [B, (H/p)*(W/p), C*p*p]; inner order (c, py, px) = the flattened Conv2d kernel."""
    B, C, H, W = images.shape
    x = images.view(B, C, H // patch, patch, W // patch, patch)
    return x.permute(0, 2, 4, 1, 3, 5).reshape(B, (H // patch) * (W // patch), C * patch * patch)


def vit_hidden_states(images: Tensor, w: Dict[str, Tensor], cfg) -> List[Tensor]:
    """All `layers+1` hidden states [B,T,D] (embeddings, then each pre-LN layer's output; the last one is BEFORE
    the final LayerNorm) — what `output_hidden_states=True` returns."""
    B = images.shape[0]
    x = F.linear(vit_patches(images, cfg.patch), w["patch_w"], w["patch_b"])
    x = torch.cat([w["cls_token"].view(1, 1, -1).expand(B, -1, -1), x], dim=1) + w["pos_emb"][None]
    hs = [x]
    D = cfg.hidden
    for l in range(cfg.layers):
        p = f"L{l}."
        h = F.layer_norm(x, (D,), w[p + "ln1_w"], w[p + "ln1_b"], cfg.eps)
        x = x + F.linear(_mha(h, w[p + "qkv_w"], w[p + "qkv_b"], cfg.heads, None), w[p + "o_w"], w[p + "o_b"])
        h = F.layer_norm(x, (D,), w[p + "ln2_w"], w[p + "ln2_b"], cfg.eps)
        x = x + F.linear(F.gelu(F.linear(h, w[p + "fc1_w"], w[p + "fc1_b"])), w[p + "fc2_w"], w[p + "fc2_b"])
        hs.append(x)
    return hs


def vit_cls_taps(images: Tensor, w: Dict[str, Tensor], cfg) -> Tensor:
    """[B, layers+1, D]: CLS row of every hidden state (`Code_Uncached/model/model.py:212`; the cached-path file
    format of `Code_Cached/preprocess_vectors.py:100-107`)."""
    return torch.stack([h[:, 0] for h in vit_hidden_states(images, w, cfg)], dim=1)


# ---------------------------------------------------------------------------------------------------------------
# U2  frozen BERT with per-layer hidden states     reference: Code_Uncached/model/encoders.py:81-91,148-159
# ---------------------------------------------------------------------------------------------------------------

def bert_hidden_states(text: Tensor, w: Dict[str, Tensor], cfg) -> List[Tensor]:
    """`text` is the reference's packed row: first half token ids, second half attention mask
    (`encoders.py:82-85`).  Post-LN layers; masked keys get the additive fp32-min bias of HF's eager path, so an
    all-masked (padding-slot) row attends uniformly."""
    n = text.shape[1] // 2
    ids, mask = text[:, :n].long(), text[:, n:]
    B = ids.shape[0]
    D = cfg.hidden
    x = w["word_emb"][ids] + w["pos_emb"][:n][None] + w["type_emb"][0][None, None]
    x = F.layer_norm(x, (D,), w["emb_ln_w"], w["emb_ln_b"], cfg.eps)
    key_bias = (1.0 - mask.to(torch.float32)) * torch.finfo(torch.float32).min
    hs = [x]
    for l in range(cfg.layers):
        p = f"L{l}."
        a = F.linear(_mha(x, w[p + "qkv_w"], w[p + "qkv_b"], cfg.heads, key_bias), w[p + "o_w"], w[p + "o_b"])
        a = F.layer_norm(a + x, (D,), w[p + "ln1_w"], w[p + "ln1_b"], cfg.eps)
        f = F.linear(F.gelu(F.linear(a, w[p + "fc1_w"], w[p + "fc1_b"])), w[p + "fc2_w"], w[p + "fc2_b"])
        x = F.layer_norm(f + a, (D,), w[p + "ln2_w"], w[p + "ln2_b"], cfg.eps)
        hs.append(x)
    return hs


def bert_cls_taps(text: Tensor, w: Dict[str, Tensor], cfg) -> Tensor:
    return torch.stack([h[:, 0] for h in bert_hidden_states(text, w, cfg)], dim=1)


# ---------------------------------------------------------------------------------------------------------------
# U4  SANB adapter block                           reference: Code_*/model/modules.py:98-116
# ---------------------------------------------------------------------------------------------------------------

def adapter_block(x: Tensor, P: Dict[str, Tensor], prefix: str, activation: str = "RELU") -> Tensor:
    h = F.linear(x, P[prefix + "fc_down.weight"], P[prefix + "fc_down.bias"])
    h = F.gelu(h) if activation == "GELU" else F.relu(h)
    return F.linear(h, P[prefix + "fc_up.weight"], P[prefix + "fc_up.bias"]) + x


# ---------------------------------------------------------------------------------------------------------------
# U3 / C1  intra+inter side network                reference: Code_Uncached/model/model.py:209-271,
#                                                             Code_Cached/model/model.py:300-349
# ---------------------------------------------------------------------------------------------------------------

def side_layer_list(side_adapter_list: str, remove_first: bool) -> List[int]:
    """`model.py:172-177`: hidden-state indices tapped by the side network."""
    body = [int(i) + 1 for i in side_adapter_list.split(",")]
    return body if remove_first else [0] + body


def side_network(taps_cv: Tensor, taps_text: Tensor, P: Dict[str, Tensor], layers: Sequence[int],
                 fusion: str = "gated", activation: str = "RELU", remove_first: bool = False,
                 cv_head: str = "mm_encoder.cv_encoder.image_net.classifier.",
                 text_head: str = "mm_encoder.bert_encoder.text_encoders.title.fc.",
                 pre: str = "mm_encoder.", modality: str = "intra_inter") -> Tuple[Tensor, Tensor, Tensor]:
    """taps_* : [M, L+1, D] CLS taps (all hidden states; `layers` selects).  Returns (cv, text, mm), each [M,E].
    `cv_head`/`text_head` name the 768->64 projections (Uncached keys by default; Cached uses
    `mm_encoder.cv_pre_fc.` / `mm_encoder.bert_pre_fc.`, `Code_Cached/model/model.py:261-262`).
    `modality` "inter" (no "intra"): the cv / text towers do not exist (model.py:178-205,228,253,260) and (None, None, mm)
    is returned — the reference hands back their untouched initial states, which its ModelMM ignores (model.py:70-72)."""
    zeros = torch.zeros_like(taps_cv[:, 0])
    intra = "intra" in modality
    if remove_first:                                   # model.py:215-218
        cv, text, mm = taps_cv[:, 0], taps_text[:, 0], zeros
    else:
        cv, text, mm = zeros, zeros, zeros
    for k, l in enumerate(layers):
        tv, tt = taps_cv[:, l], taps_text[:, l]
        if intra:
            if fusion == "gated":                       # model.py:229-236
                g = torch.sigmoid(P[pre + f"side_gate_params_cv.{k}"] / 0.1)
                f_cv = g * tv + (1 - g) * cv
                g = torch.sigmoid(P[pre + f"side_gate_params_text.{k}"] / 0.1)
                f_text = g * tt + (1 - g) * text
            else:                                       # model.py:237-239
                f_cv, f_text = tv + cv, tt + text
            text = adapter_block(f_text, P, pre + f"bert_adapter_list.{k}.", activation)
            cv = adapter_block(f_cv, P, pre + f"cv_adapter_list.{k}.", activation)
        if fusion == "gated":                           # model.py:245-250
            g = torch.sigmoid(P[pre + f"side_gate_params_mm.{k}"] / 0.1)
            mm = mm + g * tv + (1 - g) * tt
        else:                                           # model.py:251-252
            mm = mm + tv + tt
        mm = adapter_block(mm, P, pre + f"mm_adapter_list.{k}.", activation)
    mm = F.linear(mm, P[pre + "fc_mm.weight"], P[pre + "fc_mm.bias"])               # model.py:256-257
    mm = F.linear(mm, P[pre + "fc_mm_down.weight"], P[pre + "fc_mm_down.bias"])     # model.py:268-269
    if not intra:
        return None, None, mm
    text = F.linear(text, P[pre + "fc_bert.weight"], P[pre + "fc_bert.bias"])      # model.py:253-255
    cv = F.linear(cv, P[pre + "fc_cv.weight"], P[pre + "fc_cv.bias"])
    cv = F.linear(cv, P[cv_head + "weight"], P[cv_head + "bias"])                    # model.py:260-267
    text = F.linear(text, P[text_head + "weight"], P[text_head + "bias"])
    return cv, text, mm


# ---------------------------------------------------------------------------------------------------------------
# V1  IISAN-Versa side network                     reference: Code_Cached_Asym/model/model.py:322-429
# ---------------------------------------------------------------------------------------------------------------

def versa_side_network(taps_cv: Tensor, taps_text: Tensor, P: Dict[str, Tensor], layers_cv: Sequence[int],
                       layers_text: Sequence[int], fusion: str = "gated", activation: str = "RELU",
                       remove_first: bool = False, pre: str = "mm_encoder.") -> Tuple[Tensor, Tensor, Tensor]:
    """Asymmetric towers: taps_cv [M, Lc+1, Di], taps_text [M, Lt+1, Dt].  The longer tower first runs its surplus
    leading SANBs alone (group layer-drop, model.py:353-378); in the aligned loop the wider modality's tap goes through
    down_project_list[i] (dim-align, model.py:400-411) before the inter-modal gated sum."""
    Di, Dt = taps_cv.shape[-1], taps_text.shape[-1]
    M = taps_cv.shape[0]
    n_cv, n_t = len(layers_cv), len(layers_text)
    zc, zt, zm = taps_cv.new_zeros(M, Di), taps_text.new_zeros(M, Dt), taps_cv.new_zeros(M, min(Di, Dt))
    cv, text = (taps_cv[:, 0], taps_text[:, 0]) if remove_first else (zc, zt)
    mm = zm
    gated = fusion == "gated"
    diff_t, diff_cv = max(n_t - n_cv, 0), max(n_cv - n_t, 0)

    def intra(state, tap, gate_name):
        if gated:
            g = torch.sigmoid(P[pre + gate_name] / 0.1)
            return g * tap + (1 - g) * state
        return tap + state

    for k in range(diff_t):
        text = adapter_block(intra(text, taps_text[:, layers_text[k]], f"side_gate_params_text.{k}"), P, pre + f"bert_adapter_list.{k}.", activation)
    for k in range(diff_cv):
        cv = adapter_block(intra(cv, taps_cv[:, layers_cv[k]], f"side_gate_params_cv.{k}"), P, pre + f"cv_adapter_list.{k}.", activation)
    for i in range(min(n_cv, n_t)):
        kc, kt = i + diff_cv, i + diff_t
        tv, tt = taps_cv[:, layers_cv[kc]], taps_text[:, layers_text[kt]]
        text = adapter_block(intra(text, tt, f"side_gate_params_text.{kt}"), P, pre + f"bert_adapter_list.{kt}.", activation)
        cv = adapter_block(intra(cv, tv, f"side_gate_params_cv.{kc}"), P, pre + f"cv_adapter_list.{kc}.", activation)
        av, at = tv, tt
        if Dt > Di:
            at = F.linear(tt, P[pre + f"down_project_list.{i}.weight"], P[pre + f"down_project_list.{i}.bias"])
        elif Di > Dt:
            av = F.linear(tv, P[pre + f"down_project_list.{i}.weight"], P[pre + f"down_project_list.{i}.bias"])
        if gated:
            g = torch.sigmoid(P[pre + f"side_gate_params_mm.{i}"] / 0.1)
            mm = mm + g * av + (1 - g) * at
        else:
            mm = mm + av + at       # (the reference always gates here, model.py:413-415; plain sum kept for symmetry)
        mm = adapter_block(mm, P, pre + f"mm_adapter_list.{i}.", activation)
    text = F.linear(F.linear(text, P[pre + "fc_bert.weight"], P[pre + "fc_bert.bias"]), P[pre + "bert_pre_fc.weight"], P[pre + "bert_pre_fc.bias"])
    cv = F.linear(F.linear(cv, P[pre + "fc_cv.weight"], P[pre + "fc_cv.bias"]), P[pre + "cv_pre_fc.weight"], P[pre + "cv_pre_fc.bias"])
    mm = F.linear(F.linear(mm, P[pre + "fc_mm.weight"], P[pre + "fc_mm.bias"]), P[pre + "fc_mm_down.weight"], P[pre + "fc_mm_down.bias"])
    return cv, text, mm


# ---------------------------------------------------------------------------------------------------------------
# U5  SASRec user encoder                          reference: Code_Uncached/model/encoders.py:60-65, modules.py:6-96
# ---------------------------------------------------------------------------------------------------------------

def sasrec_mask(log_mask: Tensor) -> Tensor:
    """[B,1,S,S] additive mask: 0 where key<=query and key is a real position, else -1e9 (`encoders.py:60-64`)."""
    S = log_mask.shape[-1]
    m = (log_mask != 0)[:, None, None, :].expand(-1, -1, S, -1)
    return torch.where(torch.tril(m), 0.0, -1e9)


def sasrec(x: Tensor, log_mask: Tensor, P: Dict[str, Tensor], heads: int, n_layers: int,
           pre: str = "user_encoder.transformer_encoder.", drop: Dict[int, Tensor] | None = None) -> Tensor:
    """x [B,S,E] -> [B,S,E].  `drop` (optional) maps a dropout site to its keep-factor tensor (0 or 1/(1-p)) so a test
    can inject the exact masks the HIP path generates; sites follow the reference's four nn.Dropout calls
    (modules.py:17,31,62,94): 0 = after the embedding LayerNorm [B,S,E]; 1+3l = attention probabilities of block l
    [B,H,S,S]; 2+3l = fc output [B,S,E]; 3+3l = FFN output [B,S,E].  None = identity (eval)."""
    B, S, E = x.shape
    d = E // heads
    mask = sasrec_mask(log_mask)
    dr = (lambda site, t: t * drop[site]) if drop is not None else (lambda site, t: t)
    x = F.layer_norm(x + P[pre + "position_embedding.weight"][:S][None], (E,),
                     P[pre + "layer_norm.weight"], P[pre + "layer_norm.bias"], 1e-6)       # modules.py:89-93
    x = dr(0, x)
    for l in range(n_layers):
        a = pre + f"transformer_blocks.{l}.multi_head_attention."
        f = pre + f"transformer_blocks.{l}.feed_forward."
        q = F.linear(x, P[a + "w_Q.weight"]).view(B, S, heads, d).transpose(1, 2)          # modules.py:54-57
        k = F.linear(x, P[a + "w_K.weight"]).view(B, S, heads, d).transpose(1, 2)
        v = F.linear(x, P[a + "w_V.weight"]).view(B, S, heads, d).transpose(1, 2)
        s = torch.matmul(q, k.transpose(-1, -2)) / (d ** 0.5) + mask                       # modules.py:28-30
        c = torch.matmul(dr(1 + 3 * l, torch.softmax(s, -1)), v).transpose(1, 2).reshape(B, S, E)
        x = F.layer_norm(x + dr(2 + 3 * l, F.linear(c, P[a + "fc.weight"])), (E,),
                         P[a + "layer_norm.weight"], P[a + "layer_norm.bias"], 1e-6)       # modules.py:60-64
        h = F.linear(F.relu(F.linear(x, P[f + "w_1.weight"], P[f + "w_1.bias"])),
                     P[f + "w_2.weight"], P[f + "w_2.bias"])
        x = F.layer_norm(x + dr(3 + 3 * l, h), (E,), P[f + "layer_norm.weight"], P[f + "layer_norm.bias"], 1e-6)  # modules.py:15-18
    return x


# ---------------------------------------------------------------------------------------------------------------
# U6  in-batch debiased cross-entropy              reference: Code_Uncached/model/model.py:61-105
# ---------------------------------------------------------------------------------------------------------------

def inbatch_logits(ids: Tensor, score: Tensor, prec: Tensor, log_mask: Tensor, pop_prob: Tensor) -> Tuple[Tensor, Tensor]:
    """Masked logits [T, M] and labels [T] (T = bs*S, M = bs*(S+1)), written without the per-sequence Python
    loop of `model.py:92-100` but with the same evaluation order: debias, column-padding fill, then the
    false-negative fill that spares only the positive."""
    bs, S = log_mask.shape
    M = bs * (S + 1)
    ids = ids.view(-1)
    debias = torch.log(pop_prob[ids])                                                       # model.py:63-64
    z = prec @ score.t() - debias[None, :]                                                  # model.py:86-87
    col_pad = torch.cat([log_mask, torch.ones(bs, 1, dtype=log_mask.dtype)], 1).view(-1) == 0
    z = torch.where(col_pad[None, :], torch.full_like(z, -1e4), z)                          # model.py:88-89
    seq_ids = ids.view(bs, S + 1)
    same = (ids[None, None, :] == seq_ids[:, :, None]).any(1)                               # [bs, M]
    label = (torch.arange(bs)[:, None] * (S + 1) + torch.arange(1, S + 1)[None, :]).view(-1)   # model.py:83-85
    reject = same[:, None, :].expand(bs, S, M).reshape(bs * S, M).clone()
    reject[torch.arange(bs * S), label] = False                                             # model.py:98-99
    z = torch.where(reject, torch.full_like(z, -1e4), z)                                    # model.py:100
    return z, label


def inbatch_ce(ids: Tensor, score: Tensor, prec: Tensor, log_mask: Tensor, pop_prob: Tensor) -> Tensor:
    z, label = inbatch_logits(ids, score, prec, log_mask, pop_prob)
    keep = log_mask.reshape(-1) != 0                                                        # model.py:102
    return F.cross_entropy(z[keep], label[keep])                                            # model.py:104


# ---------------------------------------------------------------------------------------------------------------
# ModelMM.forward on taps (everything after the encoders)    reference: Code_Uncached/model/model.py:61-105
# ---------------------------------------------------------------------------------------------------------------

def model_loss_from_taps(ids: Tensor, taps_cv: Tensor, taps_text: Tensor, log_mask: Tensor, pop_prob: Tensor,
                         P: Dict[str, Tensor], layers: Sequence[int], heads: int = 2, n_layers: int = 2,
                         modality: str = "intra_inter", **side_kw) -> Tuple[Tensor, Dict[str, Tensor]]:
    bs, S = log_mask.shape
    if "inter" not in modality:   # model.py:73-74 concatenates cv with the LIST [text, mm] the IISAN wrapper returns: not runnable in the reference
        raise NotImplementedError("modality 'intra' does not run with the IISAN wrapper in the reference (model.py:73-74)")
    cv, text, mm = side_network(taps_cv, taps_text, P, layers, modality=modality, **side_kw)
    if "intra_inter" in modality:
        score = F.linear(torch.cat([cv, text, mm], 1), P["com_dense.weight"], P["com_dense.bias"])  # model.py:67-69
    else:
        score = F.linear(mm, P["com_dense.weight"], P["com_dense.bias"])                            # model.py:70-72
    E = score.shape[1]
    prec = sasrec(score.view(bs, S + 1, E)[:, :-1], log_mask, P, heads, n_layers).reshape(-1, E)    # model.py:74-77
    loss = inbatch_ce(ids, score, prec, log_mask, pop_prob)
    return loss, dict(cv=cv, text=text, mm=mm, score=score, prec=prec)


# ---------------------------------------------------------------------------------------------------------------
# U7  eval scoring                                  reference: Code_Uncached/data_utils/metrics.py:59-67,157-246
# ---------------------------------------------------------------------------------------------------------------

def eval_ranks(prec_last: Tensor, item_emb: Tensor, histories: Sequence[Tensor], targets: Tensor) -> Tensor:
    """rank (1-based, int64) of each user's target among items 1..item_num, history scored -inf
    (`metrics.py:202-207`).  rank = 1 + #{c>=1 : score[c] > score[target]} + #{c>=1, c<target : score[c]==score[target]}
    — ties broken towards the lower item id, the deterministic tie rule the HIP kernel pins (the reference's
    `torch.argsort` has unspecified tie order).  If the target itself is in the history its score is -inf too
    (repeat purchase; reproduced, not fixed)."""
    scores = prec_last @ item_emb.t()
    out = []
    for u in range(scores.shape[0]):
        s = scores[u].clone()
        s[histories[u]] = -math.inf
        t = int(targets[u])
        st = s[t]
        c = s[1:]
        idx = torch.arange(1, s.shape[0])
        out.append(1 + int((c > st).sum()) + int(((c == st) & (idx < t)).sum()))
    return torch.tensor(out, dtype=torch.int64)


def eval_topk(prec_last: Tensor, item_emb: Tensor, histories: Sequence[Tensor], k: int = 10) -> Tensor:
    """The first k item ids (int64 [U, k]) of each user's recommendation list: `order = argsort(score, descending)` over the
    history-masked score row without column 0 (`metrics.py:60` on the row built at `metrics.py:202-206`), item id = position
    in that row + 1.  Ties towards the lower item id (a STABLE descending argsort; the reference's `torch.argsort` leaves tie
    order unspecified) — the rule `eval_ranks` pins, so `eval_topk(...)[u, r-1] == target` whenever `eval_ranks` says r <= k.
    History items score -inf and therefore come last, in ascending id order, only when fewer than k other items exist."""
    scores = prec_last @ item_emb.t()
    out = []
    for u in range(scores.shape[0]):
        s = scores[u].clone()
        s[histories[u]] = -math.inf
        out.append(torch.argsort(s[1:], descending=True, stable=True)[:k] + 1)
    return torch.stack(out)


def hit_ndcg(ranks: Tensor, topk: int = 10) -> Tuple[Tensor, Tensor]:
    """`metrics_topK` (`metrics.py:59-67`): Hit@k = [rank<=k], nDCG@k = 1/log2(rank+1) inside the top k."""
    hit = (ranks <= topk).to(torch.float32)
    ndcg = torch.where(ranks <= topk, 1.0 / torch.log2(ranks.to(torch.float64) + 1.0), torch.zeros((), dtype=torch.float64))
    return hit, ndcg.to(torch.float32)


# ---------------------------------------------------------------------------------------------------------------
# trainer rules                                     reference: Code_Uncached/run.py:177-224 (freeze), :296-336 (groups)
# ---------------------------------------------------------------------------------------------------------------

def adam_group_of(name: str) -> str:
    """The 5-way parameter grouping of `run.py:296-321`, restated on the state-dict key."""
    if "cv" in name:
        if ("fc" in name and "fc_" not in name) or "classifier" in name or "decoder_pred" in name:
            return "recsys"
        return "adapter_cv" if ("adapter" in name or "lora" in name) else "image_net"
    if "bert" in name:
        if "fc" in name and "fc_" not in name:
            return "recsys"
        return "adapter_text" if ("adapter" in name or "lora" in name) else "text_encoder"
    if "mm_adapter" in name:
        return "adapter_cv"
    return "recsys"


def adam_step(p: Tensor, g: Tensor, m: Tensor, v: Tensor, step: int, lr: float,
              b1: float = 0.9, b2: float = 0.999, eps: float = 1e-8) -> Tuple[Tensor, Tensor, Tensor]:
    """torch.optim.Adam defaults (no weight decay, no amsgrad) — what `run.py:323-336` constructs."""
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    mh = m / (1 - b1 ** step)
    vh = v / (1 - b2 ** step)
    return p - lr * mh / (vh.sqrt() + eps), m, v
