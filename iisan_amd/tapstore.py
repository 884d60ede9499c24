"""Device-resident packed tap store for the Cached / Versa paths.

The reference's Cached dataset does 22 `torch.load` calls per sample on the host — one `[L+1, 768]` file per item and
modality (`Code_Cached/data_utils/dataset.py:29-34,77-90`) — and ships all 13 layers although 7 are used.  Here the
catalogue's SELECTED taps live in HBM as one `[item_num+1, n_sel, D]` table per modality (Scientific, 7 layers, fp32:
437 MB per modality; bf16: 218 MB) and a training step gathers its `[M, n_sel, D]` rows on the device by item id
(`iisan_gather_taps`, a coalesced 16-byte-lane copy/convert).  Row 0 is the padding item: all zeros, as the dataset
produces for pad slots (`dataset.py:79-84`).
"""
from __future__ import annotations

from typing import Sequence

import torch

from . import _lib

_STORE = {"fp32": (_lib.IISAN_F32, torch.float32), "fp16": (_lib.IISAN_F16, torch.float16),
          "bf16": (_lib.IISAN_BF16, torch.bfloat16)}


class TapStore:
    """`taps` [N, L+1, D] (any float dtype, any device) -> packed [N, n_sel, D] on `device` in `store` precision."""

    def __init__(self, taps: torch.Tensor, layers: Sequence[int], device="cuda", store: str = "fp32",
                 zero_padding_row: bool = True):
        if store not in _STORE:
            raise ValueError(f"TapStore: store must be one of {sorted(_STORE)}")
        self.code, tdt = _STORE[store]
        self.layers = [int(l) for l in layers]
        sel = taps[:, self.layers].to(torch.float32)
        if zero_padding_row:
            sel = sel.clone()
            sel[0] = 0
        self.table = sel.to(device=device, dtype=tdt).contiguous()
        self.rows, self.n_sel, self.dim = self.table.shape
        if (self.n_sel * self.dim) % 8:
            raise ValueError("TapStore: n_sel * D must be a multiple of 8")

    def nbytes(self) -> int:
        return self.table.numel() * self.table.element_size()

    def gather(self, ids: torch.Tensor) -> torch.Tensor:
        """ids int64 [...] on the store's device -> fp32 [ids.numel(), n_sel, D]."""
        lib = _lib.load()
        ids = ids.reshape(-1).to(device=self.table.device, dtype=torch.int64).contiguous()
        if not ids.is_cuda:
            raise _lib.IisanHipError("TapStore.gather needs device tensors; there is no CPU path")
        out = torch.empty((ids.numel(), self.n_sel, self.dim), dtype=torch.float32, device=ids.device)
        _lib.check(lib.iisan_gather_taps(self.code, self.table.data_ptr(), self.rows, ids.data_ptr(), out.data_ptr(),
                                         ids.numel(), self.n_sel * self.dim, torch.cuda.current_stream().cuda_stream),
                   "iisan_gather_taps")
        return out
