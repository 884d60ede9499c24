"""world_size-2 CPU (gloo) tests of the data-parallel glue: sharding semantics, the one gradient collective, the eval
gather.  The per-rank 'model' is the CPU oracle (test infrastructure) — what is under test is iisan_amd/dp.py and the
equivalence 'W-rank step == mean of W single-rank gradients' (SURVEY.md §8e)."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from iisan_amd import dp, synth, weights
    from oracle import iisan_oracle as O
    # each rank owns a different shard of sequences; negatives are rank-local
    P = {k: v.clone().requires_grad_(True) for k, v in weights.make_trainable_params(seed=99, cached=True).items()}
    names = sorted(P)
    layers = O.side_layer_list("1,3,5,7,9,11", False)
    kw = dict(cv_head="mm_encoder.cv_pre_fc.", text_head="mm_encoder.bert_pre_fc.")

    def flat_grad(seed):
        b = synth.scientific_batch(bs=2, seed=seed, res=8, item_num=30)
        tc, tt = synth.cached_taps(b.ids, 12, 768, seed=seed), synth.cached_taps(b.ids, 12, 768, seed=seed + 50)
        for p in P.values():
            p.grad = None
        loss, _ = O.model_loss_from_taps(b.ids, tc, tt, b.log_mask, b.pop_prob, P, layers, **kw)
        loss.backward()
        return torch.cat([P[n].grad.reshape(-1) for n in names])

    mine = flat_grad(100 + rank).clone()
    reduced = dp.allreduce_sum_(mine.clone()) / world
    both = torch.stack([flat_grad(100 + r) for r in range(world)]).mean(0)
    ok_grad = torch.allclose(reduced, both, rtol=1e-5, atol=1e-8)
    # eval gather: contiguous shards, padded tail, truncated after the gather
    n, bs = 23, 4
    idx = dp.sequential_shard(n, rank, world, bs)
    vals = torch.tensor(idx, dtype=torch.float32)
    got = dp.gather_concat(vals, n)
    ok_eval = torch.equal(got, torch.arange(n, dtype=torch.float32))
    out[rank] = (ok_grad, ok_eval, len(idx))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_allreduce_and_eval_gather():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    port = 29500 + (os.getpid() % 400)
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    assert len(out) == world
    for r in range(world):
        ok_grad, ok_eval, n_idx = out[r]
        assert ok_grad and ok_eval and n_idx == 12


def test_shard_indices_match_torch_distributed_sampler():
    from torch.utils.data.distributed import DistributedSampler
    sys.path.insert(0, ROOT)
    from iisan_amd import dp
    data = list(range(37))
    for world in (1, 2, 8):
        for epoch in (0, 3):
            seen = []
            for rank in range(world):
                s = DistributedSampler(data, num_replicas=world, rank=rank, shuffle=True, seed=0)
                s.set_epoch(epoch)
                mine = dp.shard_indices(len(data), rank, world, epoch, seed=0)
                assert mine == list(iter(s))
                seen += mine
            assert set(seen) == set(data)
