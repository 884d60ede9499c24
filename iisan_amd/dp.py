"""Data-parallel glue of the hot path (one process per GPU, torch.distributed: "nccl" == RCCL over xGMI on ROCm,
"gloo" in the CPU tests).  The path shards by user sequence and has exactly ONE data-path collective per step — the
SUM all-reduce of the flat gradient buffer (`FlatTrainer.step`); negatives stay rank-local as in the reference
(`Code_Uncached/model/model.py:86`: logits use only the rank's own `score_embs`).

* `shard_indices`        torch DistributedSampler semantics used at `Code_Uncached/run.py:146,395`
* `sequential_shard`     `SequentialDistributedSampler` (`Code_Uncached/data_utils/dataset.py:294-321`)
* `gather_concat`        `distributed_concat` (`Code_Uncached/data_utils/metrics.py:43-47`)
* `allreduce_sum_`       the SUM half of DDP's gradient averaging (`run.py:287`) on one flat buffer; the 1/world is
                         folded into the fused Adam launch (`FlatTrainer.step`)
* `broadcast_`           DDP's initial parameter sync

Backend "gloo" moves device tensors through a host copy (gloo implements only a subset of the collectives for GPU
tensors): that is the transport of the single-GPU multi-rank tests, never of a production run.
"""
from __future__ import annotations

import math
from typing import List

import torch
import torch.distributed as dist


def shard_indices(n: int, rank: int, world: int, epoch: int = 0, seed: int = 0, shuffle: bool = True) -> List[int]:
    """Indices of this rank for one epoch: permutation seeded with seed+epoch, padded by wrap-around to a multiple of
    `world`, then strided `rank::world` (torch.utils.data.distributed.DistributedSampler, drop_last=False)."""
    if shuffle:
        g = torch.Generator()
        g.manual_seed(seed + epoch)
        idx = torch.randperm(n, generator=g).tolist()
    else:
        idx = list(range(n))
    total = math.ceil(n / world) * world
    pad = total - len(idx)
    if pad > 0:
        idx += (idx * math.ceil(pad / len(idx)))[:pad]
    return idx[rank:total:world]


def sequential_shard(n: int, rank: int, world: int, batch_size: int) -> List[int]:
    """Contiguous eval shard, tail padded with the last index (dataset.py:308-318)."""
    num = int(math.ceil(n / batch_size / world)) * batch_size
    idx = list(range(n)) + [n - 1] * (num * world - n)
    return idx[rank * num:(rank + 1) * num]


def _on() -> bool:
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def _staged(t: torch.Tensor) -> bool:
    return t.is_cuda and dist.get_backend() == "gloo"


def gather_concat(t: torch.Tensor, total: int) -> torch.Tensor:
    """all_gather + concat + truncate to the true dataset length (metrics.py:43-47).  Runs on ALL ranks: the
    reference's rank-0-only test eval (`run.py:433-436`) would deadlock with world_size > 1."""
    if not _on():
        return t[:total]
    src = t.cpu() if _staged(t) else t.contiguous()
    outs = [torch.empty_like(src) for _ in range(dist.get_world_size())]
    dist.all_gather(outs, src)
    return torch.cat(outs, 0)[:total].to(t.device)


def allreduce_sum_(flat: torch.Tensor) -> torch.Tensor:
    """In-place SUM over ranks of one flat buffer — the ONE data-path collective of a training step."""
    if _on():
        if _staged(flat):
            h = flat.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM)
            flat.copy_(h)
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    return flat


def broadcast_(flat: torch.Tensor, src: int = 0) -> torch.Tensor:
    if _on():
        if _staged(flat):
            h = flat.cpu()
            dist.broadcast(h, src=src)
            flat.copy_(h)
        else:
            dist.broadcast(flat, src=src)
    return flat
