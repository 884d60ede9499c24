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

from . import dp, ops
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


FLAT_ALIGN = 16          # floats: 64 bytes


class FlatTrainer:
    """One training step of the hot path with flat parameter / gradient storage (see module docstring)."""

    def __init__(self, model, args, world_size: int = 1):
        self.model, self.args, self.world = model, args, world_size
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        order = {g: i for i, g in enumerate(GROUP_ORDER)}
        named.sort(key=lambda np_: order[adam_group_of(np_[0])])       # stable: groups contiguous
        self.names = [n for n, _ in named]
        # every tensor starts on a 64-byte boundary of the flat buffers: packed back to back, the 21 one-element gates
        # left every later weight matrix misaligned and the fp32 GEMM fell back to 4-byte loads for it.  The padding
        # elements are zero parameters with zero gradients: Adam leaves them at zero, the all-reduce carries them along.
        self.n_params = sum(p.numel() for _, p in named)
        self.offsets, total = [], 0
        for _, p in named:
            self.offsets.append(total)
            total += (p.numel() + FLAT_ALIGN - 1) // FLAT_ALIGN * FLAT_ALIGN
        dev = named[0][1].device
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(total, dtype=torch.float32, device=dev)
        self.m = torch.zeros_like(self.flat)
        self.v = torch.zeros_like(self.flat)
        self.seg_end, self.seg_lr, o = [], [], 0
        self._bound = []
        lrs = group_lrs(args)
        cur = None
        for (n, p), o in zip(named, self.offsets):
            g = adam_group_of(n)
            if cur is not None and g != cur:
                self.seg_end.append(o)
                self.seg_lr.append(lrs[cur])
            cur = g
            k = p.numel()
            self.flat[o:o + k].copy_(p.data.reshape(-1))
            p.data = self.flat[o:o + k].view(p.shape)          # parameters become views of the flat buffer
            g = self.grad[o:o + k].view(p.shape)               # and so do their gradients (autograd accumulates in place)
            p.grad = g
            self._bound.append((p, g))
        self.seg_end.append(total)
        self.seg_lr.append(lrs[cur])
        self.step_no = 0
        # optional: HIP events around the one data-path collective (bench.py: all-reduce time per step, read after a sync)
        self.time_allreduce = False
        self._ar_events = []

    def broadcast_params(self):
        if self.world > 1:
            dp.broadcast_(self.flat, src=0)                     # DDP's initial parameter sync (run.py:287)

    def step(self, ids, images, text, log_mask) -> torch.Tensor:
        for p, g in self._bound:                                # a caller's zero_grad(set_to_none=True) detaches p.grad from
            if p.grad is not g:                                 # the flat buffer: Adam would then step on zeros — rebind
                p.grad = g
        self.grad.zero_()                                       # optimizer.zero_grad(), run.py:408
        loss = self.model(ids, images, text, log_mask, None)    # run.py:410
        prev, ops.DIRECT_PARAM_GRADS = ops.DIRECT_PARAM_GRADS, True    # kernels accumulate straight into self.grad
        try:
            loss.backward()                                     # run.py:412
        finally:
            ops.DIRECT_PARAM_GRADS = prev
        if self.world > 1:
            if self.time_allreduce:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                dp.allreduce_sum_(self.grad)
                e1.record()
                self._ar_events.append((e0, e1))
            else:
                dp.allreduce_sum_(self.grad)                    # the ONE data-path collective (DDP grad average)
        self.step_no += 1
        ops.adam_step(self.flat, self.grad, self.m, self.v, self.seg_end, self.seg_lr, self.step_no,
                      grad_scale=1.0 / self.world)              # run.py:413
        return loss


    def allreduce_ms(self, reset: bool = True):
        """(number of timed all-reduces, their total device time in ms) since the last reset; call after a device sync.
        The events sit on the stream the collective is enqueued on (torch's current stream: RCCL orders its own stream
        against it on both sides), so the span covers the wait for RCCL's stream as well as the transfer."""
        n = len(self._ar_events)
        ms = sum(a.elapsed_time(b) for a, b in self._ar_events)
        if reset:
            self._ar_events = []
        return n, ms


# ---------------------------------------------------------------------------------------------------------------
# Checkpoints in the reference's format (`data_utils/utils.py:104-110`, `run.py:262-277`):
#   {'model_state_dict', 'optimizer', 'rng_state', 'cuda_rng_state'} with `optimizer` a torch.optim.Adam state dict
#   over the five run.py groups.  With real HF encoder modules inside the model (as run.py builds it) the state dict
#   carries the reference's full key set and a checkpoint moves between the two programs in either direction; with the
#   light FrozenVit/FrozenBert containers it holds the 146 trainable tensors only (the reference would have to load it
#   with strict=False on top of its pretrained encoders).
# ---------------------------------------------------------------------------------------------------------------

def _segments(tr: "FlatTrainer"):
    """(name, offset, numel, shape) of every trainable tensor in flat-buffer order (= Adam group order, run.py:330-336)."""
    params = dict(tr.model.named_parameters())
    return [(n, o, params[n].numel(), tuple(params[n].shape)) for n, o in zip(tr.names, tr.offsets)]


def optimizer_state_dict(tr: "FlatTrainer") -> dict:
    """The FlatTrainer's Adam state as `torch.optim.Adam(build_param_groups(model, args)).state_dict()` would hold it."""
    segs = _segments(tr)
    state = {}
    if tr.step_no > 0:
        for i, (_, o, k, shape) in enumerate(segs):
            state[i] = {"step": torch.tensor(float(tr.step_no)),
                        "exp_avg": tr.m[o:o + k].view(shape).detach().cpu().clone(),
                        "exp_avg_sq": tr.v[o:o + k].view(shape).detach().cpu().clone()}
    groups, i = [], 0
    lrs = group_lrs(tr.args)
    for g in GROUP_ORDER:
        n_g = sum(1 for n in tr.names if adam_group_of(n) == g)
        groups.append({"lr": lrs[g], "betas": (0.9, 0.999), "eps": 1e-8, "weight_decay": 0, "amsgrad": False,
                       "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                       "params": list(range(i, i + n_g))})
        i += n_g
    return {"state": state, "param_groups": groups}


def load_optimizer_state_dict(tr: "FlatTrainer", sd: dict) -> None:
    segs = _segments(tr)
    n_params = sum(len(g["params"]) for g in sd["param_groups"])
    if n_params != len(segs):
        raise ValueError(f"optimizer state holds {n_params} tensors, the model has {len(segs)} trainable ones")
    step = 0
    for i, (n, o, k, shape) in enumerate(segs):
        st = sd["state"].get(i)
        if st is None:
            tr.m[o:o + k].zero_()
            tr.v[o:o + k].zero_()
            continue
        if tuple(st["exp_avg"].shape) != shape:
            raise ValueError(f"optimizer state of parameter {i} ({n}) has shape {tuple(st['exp_avg'].shape)}, expected {shape}")
        tr.m[o:o + k].copy_(st["exp_avg"].reshape(-1))
        tr.v[o:o + k].copy_(st["exp_avg_sq"].reshape(-1))
        step = max(step, int(float(st["step"])))
    tr.step_no = step


def save_checkpoint(path: str, model, tr: "FlatTrainer") -> None:
    """`save_model` of data_utils/utils.py:104-110 (model = the bare module, i.e. the reference's `model.module`)."""
    torch.save({"model_state_dict": {k: v.detach().cpu() for k, v in model.state_dict().items()},
                "optimizer": optimizer_state_dict(tr),
                "rng_state": torch.get_rng_state(),
                "cuda_rng_state": torch.cuda.get_rng_state() if torch.cuda.is_available() else None}, path)


def load_checkpoint(path: str, model, tr: "FlatTrainer" = None, strict: bool = True) -> dict:
    """run.py:262-277: state dict, optimizer state and RNG streams.  Parameters stay views of the flat buffer
    (`load_state_dict` copies in place)."""
    ck = torch.load(path, map_location="cpu", weights_only=False)
    model.load_state_dict(ck["model_state_dict"], strict=strict)
    if tr is not None and ck.get("optimizer") is not None:
        load_optimizer_state_dict(tr, ck["optimizer"])
    if ck.get("rng_state") is not None:
        torch.set_rng_state(ck["rng_state"])
    if ck.get("cuda_rng_state") is not None and torch.cuda.is_available():
        torch.cuda.set_rng_state(ck["cuda_rng_state"])
    return ck
