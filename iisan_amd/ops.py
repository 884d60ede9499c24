"""Thin host wrappers (ctypes + torch.autograd.Function) around the trainable-path entry points of
libiisan_hip.so.  Tensors, the stream and autograd bookkeeping come from PyTorch; every FLOP happens in the HIP
library.  There is no fallback: a missing library raises at first use (`_lib.load`).
"""
from __future__ import annotations

import ctypes as C
from typing import List, Sequence

import torch

from . import _lib


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _f32c(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.float32 or not t.is_contiguous():
        t = t.float().contiguous()
    return t


def _ptr_table(ts: Sequence[torch.Tensor]):
    return (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.IisanHipError("IISAN HIP ops need CUDA/ROCm tensors (got a CPU tensor); there is no CPU path")


# When True (set by trainer.FlatTrainer around its backward), the backward kernels accumulate parameter gradients
# STRAIGHT into `param.grad` (views of the trainer's flat gradient buffer) and the autograd Functions return None for
# the parameters: no temporary gradient buffers and no per-parameter AccumulateGrad add kernels (146 tiny launches per
# step).  Must stay False under torch DDP, whose reducer is driven by the AccumulateGrad hooks this bypasses.
DIRECT_PARAM_GRADS = False


def _grad_targets(params: Sequence[torch.Tensor]):
    """(views to hand to the kernel, what backward returns for the parameters)."""
    if DIRECT_PARAM_GRADS:
        views = []
        for p in params:
            if not p.requires_grad:
                views.append(torch.zeros_like(p))     # frozen tensor or placeholder (e.g. the gates of a non-gated fusion): discarded
                continue
            g = p.grad
            if g is None or not g.is_contiguous() or g.dtype != torch.float32:
                views = None
                break
            views.append(g)
        if views is not None:
            return views, (None,) * len(params)
    _, views = _flat_grads(params)
    return views, tuple(views)


def _flat_grads(params: Sequence[torch.Tensor]) -> (torch.Tensor, List[torch.Tensor]):
    """One zeroed flat buffer with a view per parameter (the kernels accumulate with +=)."""
    total = sum(p.numel() for p in params)
    flat = torch.zeros(total, dtype=torch.float32, device=params[0].device)
    views, o = [], 0
    for p in params:
        views.append(flat[o:o + p.numel()].view(p.shape))
        o += p.numel()
    return flat, views


# ---------------------------------------------------------------------------------------------------------------
# side network
# ---------------------------------------------------------------------------------------------------------------

def side_param_order(n_side: int, cached: bool = False) -> List[str]:
    """State-dict keys (relative to the IISAN wrapper module) in the order the C ABI expects
    (include/iisan_hip.h, iisan_side_net_fwd)."""
    names = []
    for tower in ("cv", "bert", "mm"):
        for k in range(n_side):
            p = f"{tower}_adapter_list.{k}."
            names += [p + "fc_down.weight", p + "fc_down.bias", p + "fc_up.weight", p + "fc_up.bias"]
    for g in ("cv", "text", "mm"):
        names += [f"side_gate_params_{g}.{k}" for k in range(n_side)]
    names += ["fc_cv.weight", "fc_cv.bias", "fc_bert.weight", "fc_bert.bias", "fc_mm.weight", "fc_mm.bias"]
    if cached:
        names += ["cv_pre_fc.weight", "cv_pre_fc.bias", "bert_pre_fc.weight", "bert_pre_fc.bias"]
    else:
        names += ["cv_encoder.image_net.classifier.weight", "cv_encoder.image_net.classifier.bias",
                  "bert_encoder.text_encoders.title.fc.weight", "bert_encoder.text_encoders.title.fc.bias"]
    names += ["fc_mm_down.weight", "fc_mm_down.bias"]
    return names


def make_side_cfg(n_side: int, dim: int, down: int, emb: int, gated: bool, gelu: bool, remove_first: bool,
                  tap_stride_cv: int, tap_stride_text: int, tap_index: Sequence[int], first_index: int = 0):
    cfg = _lib.SideCfg()
    cfg.n_side, cfg.dim_cv, cfg.dim_text, cfg.down, cfg.emb = n_side, dim, dim, down, emb
    cfg.gated, cfg.gelu, cfg.remove_first = int(gated), int(gelu), int(remove_first)
    cfg.tap_stride_cv, cfg.tap_stride_text, cfg.first_index = tap_stride_cv, tap_stride_text, first_index
    for k, i in enumerate(tap_index):
        cfg.tap_index[k] = i
    return cfg


def make_versa_cfg(dim_cv: int, dim_text: int, down: int, emb: int, gated: bool, gelu: bool, remove_first: bool,
                   tap_stride_cv: int, tap_stride_text: int, tap_index_cv: Sequence[int], tap_index_text: Sequence[int],
                   taps_exact16: bool = False):
    """Asymmetric towers (Code_Cached_Asym): separate depth/width/tap lists per modality.  `taps_exact16`: the taps came from
    fp16 storage (every value exact in fp16) — lets the dim-align products skip the tap's amax pass."""
    cfg = _lib.SideCfg()
    cfg.versa = 1
    cfg.taps_exact16 = int(bool(taps_exact16))
    cfg.n_side, cfg.n_side_text = len(tap_index_cv), len(tap_index_text)
    cfg.dim_cv, cfg.dim_text, cfg.down, cfg.emb = dim_cv, dim_text, down, emb
    cfg.gated, cfg.gelu, cfg.remove_first = int(gated), int(gelu), int(remove_first)
    cfg.tap_stride_cv, cfg.tap_stride_text = tap_stride_cv, tap_stride_text
    for k, i in enumerate(tap_index_cv):
        cfg.tap_index[k] = i
    for k, i in enumerate(tap_index_text):
        cfg.tap_index_text[k] = i
    return cfg


def versa_param_order(n_cv: int, n_text: int, align: bool) -> List[str]:
    """State-dict keys of the Versa wrapper in ABI order (include/iisan_hip.h).  Only the first min(n_cv, n_text)
    mm adapters / gates / down-projects are ever used by the reference forward (model.py:381-417)."""
    n_mm = min(n_cv, n_text)
    names = []
    for tower, n in (("cv", n_cv), ("bert", n_text), ("mm", n_mm)):
        for k in range(n):
            p = f"{tower}_adapter_list.{k}."
            names += [p + "fc_down.weight", p + "fc_down.bias", p + "fc_up.weight", p + "fc_up.bias"]
    names += [f"side_gate_params_cv.{k}" for k in range(n_cv)]
    names += [f"side_gate_params_text.{k}" for k in range(n_text)]
    names += [f"side_gate_params_mm.{k}" for k in range(n_mm)]
    if align:
        for i in range(n_mm):
            names += [f"down_project_list.{i}.weight", f"down_project_list.{i}.bias"]
    names += ["fc_cv.weight", "fc_cv.bias", "fc_bert.weight", "fc_bert.bias", "fc_mm.weight", "fc_mm.bias",
              "cv_pre_fc.weight", "cv_pre_fc.bias", "bert_pre_fc.weight", "bert_pre_fc.bias",
              "fc_mm_down.weight", "fc_mm_down.bias"]
    return names


class SideNetFn(torch.autograd.Function):
    """(taps_cv [M,Lc,D], taps_text [M,Lt,D], *params) -> item3 [M, 3*emb] = cat[cv, text, mm]."""

    @staticmethod
    def forward(ctx, cfg, taps_cv, taps_text, *params):
        lib = _lib.load()
        _need_cuda(taps_cv, taps_text, *params)
        taps_cv, taps_text = _f32c(taps_cv), _f32c(taps_text)
        ctx.orig = list(params)
        params = [_f32c(p.detach()) for p in params]
        M = taps_cv.shape[0]
        need = lib.iisan_side_net_num_params(C.byref(cfg))
        if need != len(params):
            raise _lib.IisanHipError(f"side network expects {need} parameter tensors, got {len(params)}: "
                                     f"{lib.iisan_last_error().decode()}")
        item3 = torch.empty((M, 3 * cfg.emb), dtype=torch.float32, device=taps_cv.device)
        ws = torch.empty(lib.iisan_side_net_ws_bytes(C.byref(cfg), M), dtype=torch.uint8, device=taps_cv.device)
        tab = _ptr_table(params)
        token = C.c_uint64(0)          # which kernel routes this forward took: carried to the backward call with the workspace
        _lib.check(lib.iisan_side_net_fwd(C.byref(cfg), taps_cv.data_ptr(), taps_text.data_ptr(), M, tab,
                                          item3.data_ptr(), ws.data_ptr(), ws.numel(), C.byref(token), _stream()), "iisan_side_net_fwd")
        ctx.cfg, ctx.ws, ctx.params, ctx.token = cfg, ws, params, token.value
        ctx.save_for_backward(taps_cv, taps_text)
        return item3

    @staticmethod
    def backward(ctx, d_item3):
        lib = _lib.load()
        taps_cv, taps_text = ctx.saved_tensors
        cfg, params = ctx.cfg, ctx.params
        d_item3 = _f32c(d_item3)
        views, ret = _grad_targets(ctx.orig)
        _lib.check(lib.iisan_side_net_bwd(C.byref(cfg), taps_cv.data_ptr(), taps_text.data_ptr(), taps_cv.shape[0],
                                          _ptr_table(params), d_item3.data_ptr(), _ptr_table(views), ctx.ws.data_ptr(),
                                          ctx.ws.numel(), ctx.token, _stream()), "iisan_side_net_bwd")
        ctx.ws = None
        return (None, None, None) + ret


# ---------------------------------------------------------------------------------------------------------------
# Linear (com_dense)
# ---------------------------------------------------------------------------------------------------------------

class LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b):
        lib = _lib.load()
        _need_cuda(x, w, b)
        x2 = _f32c(x).reshape(-1, x.shape[-1])
        ctx.orig = [w, b]
        w, b = _f32c(w.detach()), _f32c(b.detach())
        y = torch.empty((x2.shape[0], w.shape[0]), dtype=torch.float32, device=x.device)
        _lib.check(lib.iisan_linear_fwd(x2.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), x2.shape[0],
                                        w.shape[1], w.shape[0], _stream()), "iisan_linear_fwd")
        ctx.save_for_backward(x2, w)
        ctx.in_shape = x.shape
        ctx.need_dx = x.requires_grad
        return y.view(*x.shape[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x2, w = ctx.saved_tensors
        dy2 = _f32c(dy).reshape(-1, w.shape[0])
        dx = torch.empty_like(x2) if ctx.need_dx else None
        (dw, db), ret = _grad_targets(ctx.orig)
        _lib.check(lib.iisan_linear_bwd(x2.data_ptr(), w.data_ptr(), dy2.data_ptr(), dx.data_ptr() if dx is not None else None,
                                        dw.data_ptr(), db.data_ptr(), x2.shape[0], w.shape[1], w.shape[0], _stream()),
                   "iisan_linear_bwd")
        return ((dx.view(ctx.in_shape) if dx is not None else None),) + ret


# ---------------------------------------------------------------------------------------------------------------
# SASRec
# ---------------------------------------------------------------------------------------------------------------

def sasrec_param_order(n_blocks: int) -> List[str]:
    """Keys relative to `User_Encoder.transformer_encoder` in ABI order (include/iisan_hip.h, iisan_sasrec_fwd)."""
    names = ["position_embedding.weight", "layer_norm.weight", "layer_norm.bias"]
    for l in range(n_blocks):
        a, f = f"transformer_blocks.{l}.multi_head_attention.", f"transformer_blocks.{l}.feed_forward."
        names += [a + "w_Q.weight", a + "w_K.weight", a + "w_V.weight", a + "fc.weight", a + "layer_norm.weight",
                  a + "layer_norm.bias", f + "w_1.weight", f + "w_1.bias", f + "w_2.weight", f + "w_2.bias",
                  f + "layer_norm.weight", f + "layer_norm.bias"]
    return names


def make_sasrec_cfg(seq: int, emb: int, heads: int, blocks: int, dropout: float = 0.0, seed: int = 0):
    cfg = _lib.SasrecCfg()
    cfg.seq, cfg.emb, cfg.heads, cfg.blocks, cfg.dropout, cfg.seed = seq, emb, heads, blocks, dropout, seed
    return cfg


class SasrecFn(torch.autograd.Function):
    """(x [B,S,E], log_mask [B,S], *params) -> [B,S,E]."""

    @staticmethod
    def forward(ctx, cfg, x, log_mask, *params):
        lib = _lib.load()
        _need_cuda(x, log_mask, *params)
        x, log_mask = _f32c(x), _f32c(log_mask)
        ctx.orig = list(params)
        params = [_f32c(p.detach()) for p in params]
        B = x.shape[0]
        assert x.shape[1] == cfg.seq and x.shape[2] == cfg.emb, (x.shape, cfg.seq, cfg.emb)
        y = torch.empty_like(x)
        ws = torch.empty(lib.iisan_sasrec_ws_bytes(C.byref(cfg), B), dtype=torch.uint8, device=x.device)
        _lib.check(lib.iisan_sasrec_fwd(C.byref(cfg), x.data_ptr(), log_mask.data_ptr(), B, _ptr_table(params),
                                        y.data_ptr(), ws.data_ptr(), ws.numel(), _stream()), "iisan_sasrec_fwd")
        ctx.cfg, ctx.ws, ctx.params = cfg, ws, params
        ctx.save_for_backward(x, log_mask)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, log_mask = ctx.saved_tensors
        cfg, params = ctx.cfg, ctx.params
        dy = _f32c(dy)
        dx = torch.empty_like(x)
        views, ret = _grad_targets(ctx.orig)
        _lib.check(lib.iisan_sasrec_bwd(C.byref(cfg), x.data_ptr(), log_mask.data_ptr(), x.shape[0], _ptr_table(params),
                                        dy.data_ptr(), dx.data_ptr(), _ptr_table(views), ctx.ws.data_ptr(), ctx.ws.numel(),
                                        _stream()), "iisan_sasrec_bwd")
        ctx.ws = None
        return (None, dx, None) + ret


# ---------------------------------------------------------------------------------------------------------------
# in-batch debiased CE
# ---------------------------------------------------------------------------------------------------------------

class InbatchCeFn(torch.autograd.Function):
    """(ids [bs*(S+1)] i64, score [bs*(S+1),E], prec [bs*S,E], log_mask [bs,S], pop_prob [n]) -> scalar loss."""

    @staticmethod
    def forward(ctx, ids, score, prec, log_mask, pop_prob):
        lib = _lib.load()
        _need_cuda(ids, score, prec, log_mask, pop_prob)
        ids = ids.reshape(-1).contiguous().to(torch.int64)
        score, prec, log_mask, pop_prob = _f32c(score), _f32c(prec), _f32c(log_mask), _f32c(pop_prob)
        bs, S = log_mask.shape
        E = score.shape[1]
        assert ids.numel() == bs * (S + 1) and score.shape[0] == bs * (S + 1) and prec.shape == (bs * S, E)
        loss = torch.empty((), dtype=torch.float32, device=score.device)
        ws = torch.empty(lib.iisan_inbatch_ce_ws_bytes(bs, S), dtype=torch.uint8, device=score.device)
        token = C.c_uint64(0)
        _lib.check(lib.iisan_inbatch_ce_fwd(ids.data_ptr(), score.data_ptr(), prec.data_ptr(), log_mask.data_ptr(),
                                            pop_prob.data_ptr(), pop_prob.numel(), bs, S, E, loss.data_ptr(), ws.data_ptr(), ws.numel(),
                                            C.byref(token), _stream()), "iisan_inbatch_ce_fwd")
        ctx.ws, ctx.token = ws, token.value
        ctx.save_for_backward(ids, score, prec, log_mask, pop_prob)
        return loss

    @staticmethod
    def backward(ctx, d_loss):
        lib = _lib.load()
        ids, score, prec, log_mask, pop_prob = ctx.saved_tensors
        bs, S = log_mask.shape
        d_score, d_prec = torch.empty_like(score), torch.empty_like(prec)
        _lib.check(lib.iisan_inbatch_ce_bwd(ids.data_ptr(), score.data_ptr(), prec.data_ptr(), log_mask.data_ptr(),
                                            pop_prob.data_ptr(), bs, S, score.shape[1], 1.0, d_score.data_ptr(),
                                            d_prec.data_ptr(), ctx.ws.data_ptr(), ctx.ws.numel(), ctx.token, _stream()),
                   "iisan_inbatch_ce_bwd")
        ctx.ws = None
        # the upstream scalar stays on the device (no host sync): scale the two small gradient tensors
        return None, d_score * d_loss, d_prec * d_loss, None, None


# ---------------------------------------------------------------------------------------------------------------
# eval scoring, Adam
# ---------------------------------------------------------------------------------------------------------------

def score_rank(prec_last: torch.Tensor, item_emb: torch.Tensor, history: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """prec_last [U,E], item_emb [N+1,E], history int32 [U,Hs] (0-padded), target int32 [U] -> int32 ranks [U]."""
    lib = _lib.load()
    _need_cuda(prec_last, item_emb, history, target)
    prec_last, item_emb = _f32c(prec_last), _f32c(item_emb)
    history, target = history.to(torch.int32).contiguous(), target.to(torch.int32).contiguous()
    ranks = torch.empty(prec_last.shape[0], dtype=torch.int32, device=prec_last.device)
    _lib.check(lib.iisan_score_rank(prec_last.data_ptr(), item_emb.data_ptr(), prec_last.shape[0], item_emb.shape[0],
                                    item_emb.shape[1], history.data_ptr(), history.shape[1], target.data_ptr(),
                                    ranks.data_ptr(), _stream()), "iisan_score_rank")
    return ranks


def score_topk(prec_last: torch.Tensor, item_emb: torch.Tensor, history: torch.Tensor, k: int = 10):
    """prec_last [U,E], item_emb [N+1,E], history int32 [U,Hs] (0-padded) -> (int32 ids [U,k], fp32 scores [U,k]): each user's k
    best items outside its history, best first, ties towards the lower id (`iisan_score_topk`).  Slots beyond the number of such
    items hold id 0 / score -inf."""
    lib = _lib.load()
    _need_cuda(prec_last, item_emb, history)
    prec_last, item_emb = _f32c(prec_last), _f32c(item_emb)
    history = history.to(torch.int32).contiguous()
    U, dev = prec_last.shape[0], prec_last.device
    ids = torch.empty((U, k), dtype=torch.int32, device=dev)
    scores = torch.empty((U, k), dtype=torch.float32, device=dev)
    nb = lib.iisan_score_topk_ws_bytes(U, item_emb.shape[0], k)
    ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=dev)
    _lib.check(lib.iisan_score_topk(prec_last.data_ptr(), item_emb.data_ptr(), U, item_emb.shape[0], item_emb.shape[1],
                                    history.data_ptr() if history.numel() else None, history.shape[1], k, ids.data_ptr(),
                                    scores.data_ptr(), ws.data_ptr(), nb, _stream()), "iisan_score_topk")
    return ids, scores


def adam_step(p, g, m, v, seg_end: Sequence[int], seg_lr: Sequence[float], step: int, grad_scale: float = 1.0,
              beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8):
    lib = _lib.load()
    n = len(seg_end)
    se = (C.c_int64 * n)(*seg_end)
    sl = (C.c_float * n)(*seg_lr)
    _lib.check(lib.iisan_adam_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), se, sl, n, step,
                                   beta1, beta2, eps, grad_scale, _stream()), "iisan_adam_step")
