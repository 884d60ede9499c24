"""Batched evaluation and tap caching on the HIP path — the callers on either side of the hot path.

* `recommend_topk`  the recommendation list itself: the first k item ids of `metrics_topK`'s argsort per user
  (`metrics.py:59-60`), selected on the device from the same score tiles as the ranks (`iisan_score_topk`).
* `evaluate_ranks` / `hit_ndcg`  replace the per-user Python loop of `eval_model` + the full argsort of `metrics_topK`
  (`Code_Uncached/data_utils/metrics.py:59-67,157-246`): user vectors from SASRec, scores against the whole item table
  and the target's rank are computed on the device in two launches per user batch; users are sharded contiguously
  across ranks like `SequentialDistributedSampler` and gathered on ALL ranks (the reference's rank-0-only test eval would
  deadlock for world_size > 1).
* `item_table`  replaces `get_MM_item_embeddings` (`metrics.py:69-107`): items are sharded across ranks instead of being
  encoded redundantly by every rank.
* `build_tap_cache`  is what `Code_Cached/preprocess_vectors.py:68-112` does with HuggingFace models: the per-layer CLS
  taps `[N, L+1, 768]` of a catalogue, from the same encoder kernels as the Uncached path.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np
import torch

from . import dp, ops


@torch.no_grad()
def build_tap_cache(model, images: torch.Tensor, text: torch.Tensor, batch: int = 256) -> Tuple[torch.Tensor, torch.Tensor]:
    """CLS taps of every hidden state for a catalogue: ([N, Lc+1, 768], [N, Lt+1, 768]) fp32 on the inputs' device."""
    enc = model.mm_encoder
    cvs, txs = [], []
    for i in range(0, images.shape[0], batch):
        img, txt = images[i:i + batch].contiguous(), text[i:i + batch].contiguous()
        Lc = enc.cv_encoder.packed(img.device).cfg.layers
        Lt = enc.bert_encoder.text_encoders["title"].packed(txt.device).cfg.layers
        cvs.append(enc.cv_encoder.forward_taps(img, range(Lc + 1)))
        txs.append(enc.bert_encoder.forward_taps(txt, range(Lt + 1)))
    return torch.cat(cvs), torch.cat(txs)


@torch.no_grad()
def item_table(model, images_or_taps: torch.Tensor, text_or_taps: torch.Tensor, batch: int = 512, rank: int = 0,
               world: int = 1) -> torch.Tensor:
    """Item embedding table `com_dense(cat(cv, text, mm))` (modality 'inter': `com_dense(mm)`) [N, emb] for items 0..N-1
    (row 0 = the padding item)."""
    N = images_or_taps.shape[0]
    per = (N + world - 1) // world
    lo, hi = min(rank * per, N), min((rank + 1) * per, N)
    rows = []
    for i in range(lo, hi, batch):
        j = min(i + batch, hi)
        item3, _ = model.mm_encoder.forward_item3(images_or_taps[i:j].contiguous(), text_or_taps[i:j].contiguous())
        rows.append(model.fuse_item3(item3))
    emb = model.com_dense.weight.shape[0]
    mine = torch.cat(rows) if rows else torch.empty(0, emb, device=images_or_taps.device)
    if world == 1:
        return mine
    pad = torch.zeros(per, emb, device=mine.device)
    pad[:mine.shape[0]] = mine
    return dp.gather_concat(pad, per * world)[:N]


def _ragged(rows, n):
    """(flat int64 array, lengths, end offsets) of a list of n integer sequences."""
    import itertools
    lens = np.fromiter((len(r) for r in rows), dtype=np.int64, count=n)
    flat = np.fromiter(itertools.chain.from_iterable(rows), dtype=np.int64, count=int(lens.sum()))
    return flat, lens, np.cumsum(lens)


def _pack_users(seqs: Sequence[Sequence[int]], histories: Sequence[Sequence[int]], max_seq_len: int, hist_stride: int):
    """Left-padded history tokens / masks (BuildMMEvalDataset, dataset.py:183-189), 0-padded exclusion lists, targets.
    Vectorised: the per-user Python loop of the first version (two tensor constructions per user) was 95 % of an eval pass
    at Scientific size (12,076 users: 200 ms around 8 ms of device work)."""
    U = len(seqs)
    flat, lens, ends = _ragged(seqs, U)
    if U and int(lens.min()) < 1:
        raise ValueError("evaluate_ranks: every eval sequence needs at least its target item")
    tgt = flat[ends - 1] if U else np.zeros(0, np.int64)
    hl = np.minimum(lens - 1, max_seq_len)                           # history positions that fit the window
    tok = np.zeros((U, max_seq_len), dtype=np.int64)
    lm = np.zeros((U, max_seq_len), dtype=np.float32)
    rows = np.repeat(np.arange(U), hl)
    k = np.arange(int(hl.sum())) - np.repeat(np.cumsum(hl) - hl, hl)  # 0 .. hl[u]-1 within each user
    col = max_seq_len - hl[rows] + k
    tok[rows, col] = flat[(ends - 1 - hl)[rows] + k]
    lm[rows, col] = 1.0
    hflat, hlens, hends = _ragged(histories, U)
    hist = np.zeros((U, hist_stride), dtype=np.int32)
    hrows = np.repeat(np.arange(U), hlens)
    hk = np.arange(int(hlens.sum())) - np.repeat(hends - hlens, hlens)
    hist[hrows, hk] = hflat
    return torch.from_numpy(tok), torch.from_numpy(lm), torch.from_numpy(hist), torch.from_numpy(tgt.astype(np.int32))


HIST_MAX = 256      # exclusion-list entries per user that iisan_score_rank takes in one launch (csrc/score.hip: MAXH)


def _rank_long_history(prec_u: torch.Tensor, item_emb: torch.Tensor, history: Sequence[int], target: int) -> int:
    """Exact rank of ONE user whose exclusion list exceeds HIST_MAX entries, from launches of `iisan_score_rank` alone.
    With R(H) the kernel's rank under exclusion set H and T0 = {target} if the target is itself excluded (its score is then -inf in
    every pass, metrics.py:204-205) else {}:   R(H) = R(T0) + sum_i [R(H_i + T0) - R(T0)]   over disjoint chunks H_i of H - T0 —
    each excluded item changes the count by the same integer whichever pass it sits in, because the target's effective score and
    every item's score bits (csrc/score.hip: one MFMA chain per score) are the same in all passes."""
    dev = prec_u.device
    uniq = list(dict.fromkeys(int(c) for c in history if int(c) != 0))        # order kept, duplicates and padding dropped
    t0 = [target] if target in set(uniq) else []
    rest = [c for c in uniq if c != target]
    step = HIST_MAX - len(t0)
    chunks = [t0] + [rest[i:i + step] + t0 for i in range(0, len(rest), step)]
    hist = torch.zeros((len(chunks), HIST_MAX), dtype=torch.int32)
    for i, ch in enumerate(chunks):
        hist[i, :len(ch)] = torch.tensor(ch, dtype=torch.int32)
    r = ops.score_rank(prec_u.expand(len(chunks), -1).contiguous(), item_emb, hist.to(dev),
                       torch.full((len(chunks),), target, dtype=torch.int32, device=dev)).to(torch.int64).cpu()
    if bool((r < 1).any()):
        return -1                                      # invalid target: reported like the kernel does
    return int(r[0] + (r[1:] - r[0]).sum())


@torch.no_grad()
def evaluate_ranks(model, item_emb: torch.Tensor, eval_seqs: Sequence[Sequence[int]], histories: Sequence[Sequence[int]],
                   max_seq_len: int, batch: int = 1024, rank: int = 0, world: int = 1) -> torch.Tensor:
    """1-based rank of every user's target (int32 [U], identical on all ranks).  `eval_seqs[u]` = history + target
    (`eval_seq`), `histories[u]` = items to exclude (`user_history`, metrics.py:204-205)."""
    U = len(eval_seqs)
    # iisan_score_rank keeps a user's exclusion list in LDS (include/iisan_hip.h: hist_stride <= 256).  Users with longer lists (the
    # reference masks histories of any length, metrics.py:198-207) keep their first 256 entries in the batched pass and are re-ranked
    # exactly by `_rank_long_history` below — a few more launches of the same kernel, no other arithmetic.
    long_users = [u for u in range(U) if len(histories[u]) > HIST_MAX]
    if long_users:
        histories = list(histories)
        full = {u: histories[u] for u in long_users}
        for u in long_users:
            histories[u] = full[u][:HIST_MAX]
    hs = max(1, max(len(h) for h in histories))
    idx = dp.sequential_shard(U, rank, world, batch) if world > 1 else list(range(U))
    tok, lm, hist, tgt = _pack_users([eval_seqs[i] for i in idx], [histories[i] for i in idx], max_seq_len, hs)
    dev = item_emb.device
    was_training = model.training
    model.eval()
    out = []
    for i in range(0, len(idx), batch):
        t, m = tok[i:i + batch].to(dev), lm[i:i + batch].to(dev)
        x = item_emb[t]                                            # == com_dense(cat(tables[tokens])), metrics.py:214
        prec = model.user_encoder(x, m, None)[:, -1].contiguous()   # metrics.py:216
        r = ops.score_rank(prec, item_emb, hist[i:i + batch].to(dev), tgt[i:i + batch].to(dev))
        for k in range(i, min(i + batch, len(idx))):
            if long_users and idx[k] in full:
                r[k - i] = _rank_long_history(prec[k - i:k - i + 1], item_emb, full[idx[k]], int(tgt[k]))
        out.append(r)
    model.train(was_training)
    ranks = torch.cat(out)
    ranks = dp.gather_concat(ranks, U) if world > 1 else ranks
    # `iisan_score_rank` reports -1 for a target outside 1..item_num; the reference indexes the score row with it and raises
    # IndexError (metrics.py:206) — so does this, after the gather, on every rank alike
    if bool((ranks < 1).any()):
        bad = torch.nonzero(ranks < 1).flatten()[:8].tolist()
        raise IndexError(f"evaluate_ranks: target item id outside 1..{item_emb.shape[0] - 1} for users {bad}")
    return ranks


@torch.no_grad()
def recommend_topk(model, item_emb: torch.Tensor, input_seqs: Sequence[Sequence[int]], histories: Sequence[Sequence[int]],
                   max_seq_len: int, k: int = 10, batch: int = 1024, rank: int = 0, world: int = 1) -> Tuple[torch.Tensor, torch.Tensor]:
    """Each user's recommendation list: (int32 ids [U, k], fp32 scores [U, k]), identical on all ranks — the first k entries of
    `order = torch.argsort(y_score, descending=True)` that `metrics_topK` (`metrics.py:59-60`) computes for the row `eval_model`
    builds at `metrics.py:198-206`, as item ids (position + 1).  `input_seqs[u]` = the items fed to the user encoder (the last
    `max_seq_len` are used, `dataset.py:183-189`; for the reference's eval protocol `eval_seq[u][:-1]`), `histories[u]` = the items
    scored -inf.  Ties towards the lower item id (a stable argsort).  Consistent with `evaluate_ranks`: a target it ranks r <= k is
    `ids[u, r-1]`.  The [U, N] score matrix never exists (`iisan_score_topk`)."""
    U = len(input_seqs)
    hs = max(1, max((len(h) for h in histories), default=1))
    idx = dp.sequential_shard(U, rank, world, batch) if world > 1 else list(range(U))
    tok, lm, hist, _ = _pack_users([list(input_seqs[i]) + [0] for i in idx], [histories[i] for i in idx], max_seq_len, hs)
    dev = item_emb.device
    n_items = item_emb.shape[0] - 1
    was_training = model.training
    model.eval()
    ids_out, sc_out = [], []
    for i in range(0, len(idx), batch):
        t, m = tok[i:i + batch].to(dev), lm[i:i + batch].to(dev)
        prec = model.user_encoder(item_emb[t], m, None)[:, -1].contiguous()          # metrics.py:214-216
        ids, sc = ops.score_topk(prec, item_emb, hist[i:i + batch].to(dev), k)
        ids_out.append(ids)
        sc_out.append(sc)
    model.train(was_training)
    ids, sc = torch.cat(ids_out), torch.cat(sc_out)
    if n_items - hs < k:
        # fewer than k items outside some user's history: the reference's argsort lists the -inf (history) items next — for a stable
        # sort in ascending id order — and the kernel leaves those slots 0 (include/iisan_hip.h)
        ids_c = ids.cpu()
        for r in torch.nonzero((ids_c == 0).any(dim=1)).flatten().tolist():
            excl = sorted({int(c) for c in histories[idx[r]] if 1 <= int(c) <= n_items})
            free = int((ids_c[r] != 0).sum())
            fill = excl[:k - free]
            ids_c[r, free:free + len(fill)] = torch.tensor(fill, dtype=torch.int32)
        ids = ids_c.to(dev)
    if world > 1:
        ids, sc = dp.gather_concat(ids, U), dp.gather_concat(sc, U)
    return ids, sc


def hit_ndcg(ranks: torch.Tensor, topk: int = 10) -> Tuple[float, float]:
    """Hit@k and nDCG@k as `metrics_topK` + `eval_concat` report them (metrics.py:50-67)."""
    r = ranks.to(torch.float64)
    ok = (r >= 1) & (r <= topk)              # a rank < 1 is the kernel's "invalid target" signal, never a hit
    hit = ok.to(torch.float64)
    ndcg = torch.where(ok, 1.0 / torch.log2(r.clamp(min=1.0) + 1.0), torch.zeros_like(r))
    return float(hit.mean()), float(ndcg.mean())


def print_metrics(x, Log_file, v_or_t):
    """The reference's result line (`metrics.py:35-36`): percentages with five decimals, tab separated."""
    Log_file.info(v_or_t + "_results   {}".format('\t'.join(["{:0.5f}".format(i * 100) for i in x])))


@torch.no_grad()
def eval_model(model, user_history, eval_seq, item_embeddings_image, item_embeddings_text, test_batch_size, args, item_num,
               Log_file, v_or_t, local_rank):
    """Same signature, log lines and return value (Hit@10) as the reference's `eval_model` (`metrics.py:157-246`),
    for the modalities the IISAN wrapper serves: `item_embeddings_text` is the pair `[text, inter]` that
    `get_MM_item_embeddings` returns; with `modality == "inter"` only `inter` is read (`metrics.py:175-178,192-199`).  `eval_seq` maps user index -> item sequence (last = target), `user_history[u]`
    the items whose scores are set to -inf.  Works with or without an initialised process group; with one, users are
    sharded like `SequentialDistributedSampler` and the ranks gathered on every rank."""
    if "inter" not in args.modality:
        raise NotImplementedError("eval_model: modality 'intra' is not served by the IISAN wrapper")
    m = model.module if hasattr(model, "module") else model
    text, inter = item_embeddings_text
    dev = torch.device("cuda", local_rank) if isinstance(local_rank, int) else torch.device(local_rank)
    topK = 10
    Log_file.info(v_or_t + "_methods   {}".format('\t'.join(['Hit{}'.format(topK), 'nDCG{}'.format(topK)])))
    m.eval()
    if args.modality == "inter":
        item3 = inter.to(dev).float().contiguous()                                      # metrics.py:175-178
    else:
        item3 = torch.cat([item_embeddings_image.to(dev), text.to(dev), inter.to(dev)], dim=1).float().contiguous()
    item_emb = ops.LinearFn.apply(item3, m.com_dense.weight, m.com_dense.bias)          # metrics.py:181
    users = range(len(eval_seq))
    seqs = [list(eval_seq[u]) for u in users]
    hists = [[int(i) for i in user_history[u]] for u in users]
    import torch.distributed as dist
    on = dist.is_available() and dist.is_initialized()
    rank, world = (dist.get_rank(), dist.get_world_size()) if on else (0, 1)
    ranks = evaluate_ranks(m, item_emb, seqs, hists, max_seq_len=args.max_seq_len, batch=test_batch_size, rank=rank, world=world)
    mean_eval = list(hit_ndcg(ranks, topK))
    print_metrics(mean_eval, Log_file, v_or_t)
    return mean_eval[0]
