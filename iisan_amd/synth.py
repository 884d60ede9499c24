"""Synthetic Amazon-Scientific-shaped inputs for the IISAN hot path.

The real inputs are LMDB images + tokenised titles (`Code_Uncached/data_utils/dataset.py:56-86`); neither the
image LMDB nor the network is available, so bench.py, smoke() and the parity tests draw batches with the same
layout and the same length/padding statistics:

* `ids`      int64 [bs, S+1]  item ids, LEFT padded with 0 (`dataset.py:62-71`)
* `log_mask` fp32  [bs, S]    1 where the history position is real (`dataset.py:69-71`)
* `images`   fp32  [bs*(S+1), 3, R, R]  (x/255-.5)/.5-normalised pixels, zeros on padding slots (`dataset.py:73-84`)
* `text`     int64 [bs*(S+1), 2*W]      W WordPiece ids followed by the W-wide attention mask; all-zero row on
                                        padding slots (`run.py:124-131`, `dataset.py:79-84`)
* `pop_prob` fp32  [item_num+1]         pop_prob[0] = 1 (`data_utils/preprocess.py:76-89`)

All integer-valued pieces come from `numpy.random.RandomState` (frozen stream) so CPU oracle and GPU path see
identical inputs; only the bulk pixel tensor may optionally be drawn directly on the device for benchmarks.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
import torch

# train-sequence length histogram of Dataset/Scientific (lengths 3..11 incl. the target), SURVEY.md §8d [probe]
SCI_LEN_HIST = {3: 4892, 4: 2652, 5: 1465, 6: 895, 7: 609, 8: 356, 9: 300, 10: 208, 11: 699}
SCI_ITEM_NUM = 20314


@dataclass
class Batch:
    ids: torch.Tensor        # [bs, S+1] int64
    log_mask: torch.Tensor   # [bs, S]   fp32
    images: torch.Tensor     # [M, 3, R, R] fp32
    text: torch.Tensor       # [M, 2W] int64
    pop_prob: torch.Tensor   # [item_num+1] fp32

    def to(self, device) -> "Batch":
        return Batch(*(t.to(device) for t in (self.ids, self.log_mask, self.images, self.text, self.pop_prob)))


def make_pop_prob(item_num: int, seed: int = 7) -> torch.Tensor:
    """Zipf-like positive popularity over ids 1..item_num, normalised; index 0 (padding) is 1 so log() = 0."""
    rs = np.random.RandomState(seed)
    counts = rs.zipf(1.6, size=item_num).astype(np.float64).clip(max=5000.0)
    p = counts / counts.sum()
    return torch.from_numpy(np.concatenate([[1.0], p]).astype(np.float32))


def make_ids(bs: int, seq_len: int, item_num: int, rs: np.random.RandomState, lengths=None):
    """Left-padded id rows with Scientific-shaped real lengths (>=2: at least one history item + the target)."""
    S1 = seq_len + 1
    ids = np.zeros((bs, S1), dtype=np.int64)
    if lengths is None:
        ls = np.array([l for l in SCI_LEN_HIST if l <= S1] or [S1])
        pr = np.array([SCI_LEN_HIST.get(int(l), 1) for l in ls], dtype=np.float64)
        lengths = rs.choice(ls, size=bs, p=pr / pr.sum())
    for i, l in enumerate(lengths):
        l = int(min(max(l, 2), S1))
        ids[i, S1 - l:] = rs.choice(np.arange(1, item_num + 1), size=l, replace=False)
    log_mask = (ids[:, :-1] != 0).astype(np.float32)
    return ids, log_mask


def make_text(ids_flat: np.ndarray, words: int, vocab: int, rs: np.random.RandomState) -> np.ndarray:
    """[M, 2*words]: `[CLS] w.. [SEP] 0..` + attention mask; all-zero on padding slots.  A given item id always
    gets the same title (titles are a function of the item), drawn from a per-id RandomState."""
    M = ids_flat.shape[0]
    text = np.zeros((M, 2 * words), dtype=np.int64)
    lo = min(1000, vocab - 2)
    for m in range(M):
        it = int(ids_flat[m])
        if it == 0:
            continue
        r = np.random.RandomState(100003 + it)
        n = int(r.randint(min(5, words), words + 1))
        toks = r.randint(lo, vocab, size=n)
        toks[0] = min(101, vocab - 1)
        toks[n - 1] = min(102, vocab - 1)
        text[m, :n] = toks
        text[m, words:words + n] = 1
    return text


def make_images(ids_flat: np.ndarray, res: int, device="cpu", seed: int = 0, by_item: bool = False) -> torch.Tensor:
    """N(0,1) clipped to [-1,1] pixels, zeros on padding slots.  Drawn with a torch generator on `device`
    (CPU draws are reproducible across machines; device draws are for benchmarks only).
    `by_item`: every occurrence of an item id gets the same picture, as in the real dataset (the image is looked up
    by id, `dataset.py:73-78`); the default draws one per slot (the golden fixtures were generated that way)."""
    M = ids_flat.shape[0]
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    img = torch.randn((M, 3, res, res), generator=g, device=device, dtype=torch.float32).clamp_(-1.0, 1.0)
    real = torch.from_numpy((ids_flat != 0)).to(device)
    img *= real.view(M, 1, 1, 1).to(torch.float32)
    if by_item:
        _, first, inverse = np.unique(ids_flat, return_index=True, return_inverse=True)
        img = img.index_select(0, torch.from_numpy(first[inverse]).to(device))
    return img


def scientific_batch(bs: int, seed: int = 12345, seq_len: int = 10, item_num: int = SCI_ITEM_NUM, res: int = 224,
                     words: int = 30, vocab: int = 30522, device="cpu", images_on_device: bool = False,
                     lengths=None, dup_items: bool = False, images_by_item: bool = False) -> Batch:
    rs = np.random.RandomState(seed)
    ids, log_mask = make_ids(bs, seq_len, item_num, rs, lengths)
    if dup_items and bs >= 2:
        # force the in-batch false-negative rule to fire: copy a real item of sequence 0 into sequence 1
        ids[1, -1] = ids[0, -2]
    flat = ids.reshape(-1)
    text = make_text(flat, words, vocab, rs)
    images = make_images(flat, res, device=device if images_on_device else "cpu", seed=seed, by_item=images_by_item)
    b = Batch(torch.from_numpy(ids), torch.from_numpy(log_mask), images, torch.from_numpy(text),
              make_pop_prob(item_num))
    return b.to(device)


def cached_taps(ids_flat: torch.Tensor, n_layers: int, dim: int, seed: int = 0, device="cpu",
                dtype=torch.float32, scale: float = 0.25) -> torch.Tensor:
    """[M, n_layers+1, dim] synthetic CLS taps for the Cached path, zeros on padding slots
    (`Code_Cached/data_utils/dataset.py:77-90`)."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    M = ids_flat.numel()
    t = torch.randn((M, n_layers + 1, dim), generator=g, dtype=torch.float32) * scale
    t *= (ids_flat.reshape(-1).cpu() != 0).view(M, 1, 1).to(torch.float32)
    return t.to(device=device, dtype=dtype)
