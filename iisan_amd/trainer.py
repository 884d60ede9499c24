"""Trainer hooks of the hot path: the pieces of `Code_Uncached/run.py` that decide WHAT is trained and HOW it is
stepped, restated for the HIP path.

* `add_interIISAN_adapter_to_model`      run.py:38-39
* `apply_iisan_freeze_rules`             run.py:177-183 (freeze everything) + :214-224 (wrap, re-enable by name)
* `adam_group_of` / `build_param_groups` run.py:296-336 (five Adam groups)
* `FlatTrainer`                          run.py:408-414 (zero_grad / forward / backward / step) + DDP (run.py:287):
  all trainable tensors live in ONE flat fp32 buffer ordered by Adam group, their gradients in a second flat buffer:
  the data-parallel exchange is a single RCCL all-reduce of 16.5 MB per step and the optimiser is one fused HIP
  launch.  No GradScaler: fp32 trainables + fp16/bf16 frozen encoders need no loss scaling (reference uses fp16
  autocast + GradScaler, run.py:385,409-414).
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch
import torch.distributed as dist

from . import ops
from .model import CachedIISANAdaptedMModel, IISANAdaptedMModel, VersaIISANAdaptedMModel

GROUP_ORDER = ("text_encoder", "image_net", "recsys", "adapter_cv", "adapter_text")     # optimizer order, run.py:330-336


def add_interIISAN_adapter_to_model(mm_model, args, cached: bool = False):
    """`cached`: False = Code_Uncached, True = Code_Cached, "versa" = Code_Cached_Asym."""
    if cached == "versa":
        return VersaIISANAdaptedMModel(mm_model, args)
    return (CachedIISANAdaptedMModel if cached else IISANAdaptedMModel)(mm_model, args)


def apply_iisan_freeze_rules(model, args, cached: bool = False):
    """`fine_tune_to None` + `adapter_type IISAN`: freeze all existing parameters, wrap the encoder pair with the side
    network (new parameters default to trainable), then re-enable by NAME exactly as run.py:218-224 does."""
    for _, p in model.named_parameters():
        p.requires_grad = False
    model.mm_encoder = add_interIISAN_adapter_to_model(model.mm_encoder, args, cached)
    for name, p in model.named_parameters():
        if cached:      # Code_Cached/run.py:186
            hit = any(["user" in name, "classifier" in name, "title.fc" in name, "cv_pre_fc" in name, "bert_pre_fc" in name])
        else:           # Code_Uncached/run.py:220
            hit = any(["user" in name, "cv_proj" in name, "classifier" in name, "title.fc" in name, "lm_head" in name])
        if hit or all(["user" not in name, "encoder" not in name]):
            p.requires_grad = True
    return model


def adam_group_of(name: str) -> str:
    """run.py:296-321 restated on the parameter name."""
    if "cv" in name:
        if ("fc" in name and "fc_" not in name) or "classifier" in name or "decoder_pred" in name:
            return "recsys"
        return "image_net" if ("adapter" not in name and "lora" not in name) else "adapter_cv"
    if "bert" in name:
        if "fc" in name and "fc_" not in name:
            return "recsys"
        return "text_encoder" if ("adapter" not in name and "lora" not in name) else "adapter_text"
    if "mm_adapter" in name:
        return "adapter_cv"
    return "recsys"


def group_lrs(args) -> Dict[str, float]:
    return dict(text_encoder=args.fine_tune_lr_text, image_net=args.fine_tune_lr_image, recsys=args.lr,
                adapter_cv=args.adapter_cv_lr, adapter_text=args.adapter_bert_lr)


def build_param_groups(model, args) -> List[dict]:
    """Parameter groups for `torch.optim.Adam`, identical to run.py:330-336 (drop-in use with a stock optimiser)."""
    groups = {g: [] for g in GROUP_ORDER}
    for name, p in model.named_parameters():
        if p.requires_grad:
            groups[adam_group_of(name)].append(p)
    lrs = group_lrs(args)
    return [{"params": groups[g], "lr": lrs[g]} for g in GROUP_ORDER]


class FlatTrainer:
    """One training step of the hot path with flat parameter / gradient storage (see module docstring)."""

    def __init__(self, model, args, world_size: int = 1):
        self.model, self.args, self.world = model, args, world_size
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        order = {g: i for i, g in enumerate(GROUP_ORDER)}
        named.sort(key=lambda np_: order[adam_group_of(np_[0])])       # stable: groups contiguous
        self.names = [n for n, _ in named]
        total = sum(p.numel() for _, p in named)
        dev = named[0][1].device
        self.flat = torch.empty(total, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(total, dtype=torch.float32, device=dev)
        self.m = torch.zeros_like(self.flat)
        self.v = torch.zeros_like(self.flat)
        self.seg_end, self.seg_lr, o = [], [], 0
        lrs = group_lrs(args)
        cur = None
        for n, p in named:
            g = adam_group_of(n)
            if cur is not None and g != cur:
                self.seg_end.append(o)
                self.seg_lr.append(lrs[cur])
            cur = g
            k = p.numel()
            self.flat[o:o + k].copy_(p.data.reshape(-1))
            p.data = self.flat[o:o + k].view(p.shape)          # parameters become views of the flat buffer
            p.grad = self.grad[o:o + k].view(p.shape)          # and so do their gradients (autograd accumulates in place)
            o += k
        self.seg_end.append(o)
        self.seg_lr.append(lrs[cur])
        self.step_no = 0

    def broadcast_params(self):
        if self.world > 1:
            dist.broadcast(self.flat, src=0)                    # DDP's initial parameter sync (run.py:287)

    def step(self, ids, images, text, log_mask) -> torch.Tensor:
        self.grad.zero_()                                       # optimizer.zero_grad(), run.py:408
        loss = self.model(ids, images, text, log_mask, None)    # run.py:410
        prev, ops.DIRECT_PARAM_GRADS = ops.DIRECT_PARAM_GRADS, True    # kernels accumulate straight into self.grad
        try:
            loss.backward()                                     # run.py:412
        finally:
            ops.DIRECT_PARAM_GRADS = prev
        if self.world > 1:
            dist.all_reduce(self.grad, op=dist.ReduceOp.SUM)    # the ONE data-path collective (DDP grad average)
        self.step_no += 1
        ops.adam_step(self.flat, self.grad, self.m, self.v, self.seg_end, self.seg_lr, self.step_no,
                      grad_scale=1.0 / self.world)              # run.py:413
        return loss
