"""Frozen ViT / BERT encoders on the HIP path: weight packing + the two `*_forward_taps` calls.

Host-side counterpart of `Vit_Encoder` / `Bert_Encoder` (`Code_Uncached/model/encoders.py:23-31,116-159`) for the
IISAN mode, where only the per-layer CLS rows of the hidden states are consumed (`model.py:212-213`).  PyTorch is
used for device memory and the stream only; all arithmetic happens in libiisan_hip.so.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Sequence

import torch

from . import _lib
from .weights import BertConfig, VitConfig

_TORCH16 = {_lib.IISAN_F16: torch.float16, _lib.IISAN_BF16: torch.bfloat16}
DTYPE_NAMES = {"fp16": _lib.IISAN_F16, "f16": _lib.IISAN_F16, "bf16": _lib.IISAN_BF16}


def _ptr(t: torch.Tensor) -> int:
    return t.data_ptr()


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


class _Workspace:
    """Grow-only device scratch owned by the caller side of the ABI (the library never allocates)."""

    def __init__(self):
        self.buf = None

    def get(self, nbytes: int, device) -> torch.Tensor:
        if self.buf is None or self.buf.numel() < nbytes or self.buf.device != torch.device(device):
            self.buf = None
            self.buf = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
        return self.buf


class _PackedEncoder:
    def __init__(self, w: Dict[str, torch.Tensor], device, dtype16: int):
        self.device = torch.device(device)
        self.dtype16 = dtype16
        self._keep = []
        self.ws = _Workspace()
        self._w = w

    # dead-work policy of the executor (include/iisan_hip.h): a field of the weights struct, i.e. part of every call
    @property
    def full_blocks(self) -> bool:
        return bool(self.struct.full_blocks)

    @full_blocks.setter
    def full_blocks(self, on: bool):
        self.struct.full_blocks = 1 if on else 0

    def _m16(self, name: str) -> int:      # matrix operand of an MFMA GEMM
        t = self._w[name].to(self.device, dtype=torch.float32).to(_TORCH16[self.dtype16]).contiguous()
        self._keep.append(t)
        return _ptr(t)

    def _v32(self, name: str) -> int:      # fp32 vector / table
        t = self._w[name].to(self.device, dtype=torch.float32).contiguous()
        self._keep.append(t)
        return _ptr(t)

    def _layers(self, struct, n_layers: int, masters: bool = False):
        """masters: also keep the fp32 originals of qkv_w / fc1_w on the device (`iisan_layer_weights.qkv_w32 / fc1_w32`): the ViT
        executor folds its LayerNorms into these two matrices and then rounds to fp16 once, not twice."""
        for l in range(n_layers):
            p, L = f"L{l}.", struct.layer[l]
            if masters:
                L.qkv_w32, L.fc1_w32 = self._v32(p + "qkv_w"), self._v32(p + "fc1_w")
            L.qkv_w, L.qkv_b = self._m16(p + "qkv_w"), self._v32(p + "qkv_b")
            L.o_w, L.o_b = self._m16(p + "o_w"), self._v32(p + "o_b")
            L.fc1_w, L.fc1_b = self._m16(p + "fc1_w"), self._v32(p + "fc1_b")
            L.fc2_w, L.fc2_b = self._m16(p + "fc2_w"), self._v32(p + "fc2_b")
            L.ln1_w, L.ln1_b = self._v32(p + "ln1_w"), self._v32(p + "ln1_b")
            L.ln2_w, L.ln2_b = self._v32(p + "ln2_w"), self._v32(p + "ln2_b")


class PackedVit(_PackedEncoder):
    """ViT weights resident in HBM in kernel layout (canonical names of `iisan_amd.weights`)."""

    def __init__(self, w: Dict[str, torch.Tensor], cfg: VitConfig, device="cuda", dtype16: int = _lib.IISAN_F16,
                 keep_masters: bool = False):
        """keep_masters: keep the fp32 originals of qkv_w / fc1_w resident after the one-time LayerNorm fold (tests that fold per call);
        by default they are released once `folded` is written — the executor only reads `folded` (~200 MB for ViT-B)."""
        super().__init__(w, device, dtype16)
        self.cfg = cfg
        s = _lib.VitWeights()
        s.hidden, s.layers, s.heads, s.mlp = cfg.hidden, cfg.layers, cfg.heads, cfg.mlp
        s.image, s.patch, s.channels, s.dtype16, s.eps = cfg.image, cfg.patch, cfg.channels, dtype16, cfg.eps
        s.patch_w, s.patch_b = self._m16("patch_w"), self._v32("patch_b")
        s.cls_token, s.pos_emb = self._v32("cls_token"), self._v32("pos_emb")
        self._layers(s, cfg.layers, masters=dtype16 == _lib.IISAN_F16)
        self.struct = s
        self._w = None
        # LayerNorm 1 / 2 folded into the QKV / FC1 weights once, here, instead of at the start of every forward call
        lib = _lib.load()
        nbytes = lib.iisan_vit_fold_bytes(C.byref(s))
        if nbytes and self.device.type == "cuda":
            self._folded = torch.empty(int(nbytes), dtype=torch.uint8, device=self.device)
            with torch.cuda.device(self.device):
                _lib.check(lib.iisan_vit_fold_layernorm(C.byref(s), _ptr(self._folded), nbytes, _stream()), "iisan_vit_fold_layernorm")
            s.folded = _ptr(self._folded)
            # the fold ran on the stream current at construction; forwards may be issued on any other stream later: make the
            # folded set visible to all of them once, here (a one-time wait at pack time, never on the step)
            torch.cuda.current_stream(self.device).synchronize()
            if not keep_masters:
                masters = set()
                for l in range(cfg.layers):
                    L = s.layer[l]
                    masters.update((L.qkv_w32, L.fc1_w32))
                    L.qkv_w32 = L.fc1_w32 = None
                self._keep = [t for t in self._keep if _ptr(t) not in masters]

    def forward_taps(self, images: torch.Tensor, tap_layers: Sequence[int], chunk_items: int = 0) -> torch.Tensor:
        """images fp32 [M,C,R,R] (normalised) or uint8 [M,C,R,R] (raw pixels, normalised on the device) -> fp32
        [M, len(tap_layers), D] (`encoders.py:29-31` + `model.py:212`)."""
        lib = _lib.load()
        cfg = self.cfg
        assert images.is_cuda and images.dtype in (torch.float32, torch.uint8) and images.is_contiguous()
        assert images.shape[1:] == (cfg.channels, cfg.image, cfg.image), images.shape
        M = images.shape[0]
        taps = torch.empty((M, len(tap_layers), cfg.hidden), dtype=torch.float32, device=images.device)
        tl = (C.c_int32 * len(tap_layers))(*tap_layers)
        nbytes = lib.iisan_vit_forward_taps_ws_bytes(C.byref(self.struct), M, chunk_items)
        ws = self.ws.get(nbytes, images.device)
        fn = lib.iisan_vit_forward_taps_u8 if images.dtype == torch.uint8 else lib.iisan_vit_forward_taps
        _lib.check(fn(C.byref(self.struct), _ptr(images), M, tl, len(tap_layers), _ptr(taps),
                      chunk_items, _ptr(ws), ws.numel(), _stream()), "iisan_vit_forward_taps")
        return taps


class PackedBert(_PackedEncoder):
    def __init__(self, w: Dict[str, torch.Tensor], cfg: BertConfig, device="cuda", dtype16: int = _lib.IISAN_F16):
        super().__init__(w, device, dtype16)
        self.cfg = cfg
        s = _lib.BertWeights()
        s.hidden, s.layers, s.heads, s.mlp = cfg.hidden, cfg.layers, cfg.heads, cfg.mlp
        s.vocab, s.max_pos, s.dtype16, s.eps = cfg.vocab, cfg.max_pos, dtype16, cfg.eps
        s.word_emb, s.pos_emb, s.type_emb = self._v32("word_emb"), self._v32("pos_emb"), self._v32("type_emb")
        s.emb_ln_w, s.emb_ln_b = self._v32("emb_ln_w"), self._v32("emb_ln_b")
        self._layers(s, cfg.layers)
        self.struct = s
        self._w = None

    def forward_taps(self, text: torch.Tensor, tap_layers: Sequence[int], chunk_items: int = 0) -> torch.Tensor:
        """text int64 [M, 2W] -> fp32 [M, len(tap_layers), D] (`encoders.py:81-91` + `model.py:213`)."""
        lib = _lib.load()
        assert text.is_cuda and text.dtype == torch.int64 and text.is_contiguous() and text.shape[1] % 2 == 0
        M, words = text.shape[0], text.shape[1] // 2
        taps = torch.empty((M, len(tap_layers), self.cfg.hidden), dtype=torch.float32, device=text.device)
        tl = (C.c_int32 * len(tap_layers))(*tap_layers)
        nbytes = lib.iisan_bert_forward_taps_ws_bytes(C.byref(self.struct), M, words, chunk_items)
        ws = self.ws.get(nbytes, text.device)
        _lib.check(lib.iisan_bert_forward_taps(C.byref(self.struct), _ptr(text), M, words, tl, len(tap_layers),
                                               _ptr(taps), chunk_items, _ptr(ws), ws.numel(), _stream()),
                   "iisan_bert_forward_taps")
        return taps
