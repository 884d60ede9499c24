"""One rank of the multi-rank GPU test of the PRODUCT data-parallel step (`tests/test_gpu_dp.py` starts W of these as
fresh processes).  Environment: RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT, DP_BACKEND ("gloo": every rank on cuda:0,
collectives staged through the host; "nccl": rank r on cuda:r over RCCL), DP_OUT (directory for the per-rank dumps).

Each rank builds the Cached IISAN model from a DIFFERENT seed (so that only `broadcast_params` can make them equal),
runs `FlatTrainer(world=W).step` on its own shard of sequences (`Code_Uncached/run.py:146,287,395,408-414`), and
checks the rank-sharded eval helpers (`metrics.py:43-47,69-107,157-246`) against their world=1 result."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

ITEM_NUM, BS = 60, 6


def shard_inputs(rank, dev):
    from iisan_amd import synth
    b = synth.scientific_batch(bs=BS, seed=300 + rank, item_num=ITEM_NUM, res=8, words=4, vocab=64)
    ids = b.ids.view(-1)
    tc = synth.cached_taps(ids, 12, 768, seed=700 + rank).view(BS, 11, 13, 768)
    tt = synth.cached_taps(ids, 12, 768, seed=800 + rank).view(BS, 11, 13, 768)
    return ids.to(dev), tc.to(dev), tt.to(dev), b.log_mask.to(dev), b.pop_prob


def build(seed, dev, pop):
    from iisan_amd import factory, weights
    args = factory.make_args(drop_rate=0.0)
    model = factory.build_model(args, ITEM_NUM, pop, cached=True, device=dev)
    factory.load_trainables(model, weights.make_trainable_params(seed=seed, cached=True))
    model.train()
    return args, model


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    backend = os.environ.get("DP_BACKEND", "gloo")
    dev = torch.device("cuda", rank if backend == "nccl" else 0)
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from iisan_amd import evaluate, synth, trainer

    ids, tc, tt, lm, pop = shard_inputs(rank, dev)
    args, model = build(seed=99 + 7 * rank, dev=dev, pop=pop)
    tr = trainer.FlatTrainer(model, args, world)
    before = tr.flat.clone()
    tr.broadcast_params()
    start = tr.flat.clone()
    losses = []
    losses.append(float(tr.step(ids, tc, tt, lm)))
    out = dict(rank=rank, before=before.cpu(), start=start.cpu(), flat1=tr.flat.cpu().clone(), grad1=tr.grad.cpu().clone(),
               names=tr.names, offsets=tr.offsets, seg_end=tr.seg_end, seg_lr=tr.seg_lr)
    model.zero_grad(set_to_none=True)            # a caller detaching p.grad must not silently stop training
    losses.append(float(tr.step(ids, tc, tt, lm)))
    out.update(flat2=tr.flat.cpu().clone(), losses=losses)

    # rank-sharded eval == single-rank eval (item table sharded by item, users by contiguous block, gathered everywhere)
    n = ITEM_NUM + 1
    cat_ids = torch.arange(n)
    cat_tc = synth.cached_taps(cat_ids, 12, 768, seed=11).to(dev)
    cat_tt = synth.cached_taps(cat_ids, 12, 768, seed=12).to(dev)
    model.eval()
    tab_w = evaluate.item_table(model, cat_tc, cat_tt, batch=16, rank=rank, world=world)
    tab_1 = evaluate.item_table(model, cat_tc, cat_tt, batch=16, rank=0, world=1)
    g = torch.Generator().manual_seed(5)
    seqs, hists = [], []
    for u in range(23):
        L = int(torch.randint(2, 12, (1,), generator=g))
        s = (torch.randperm(ITEM_NUM, generator=g)[:L] + 1).tolist()
        seqs.append(s)
        hists.append(s[:-1])
    r_w = evaluate.evaluate_ranks(model, tab_1, seqs, hists, max_seq_len=10, batch=4, rank=rank, world=world)
    r_1 = evaluate.evaluate_ranks(model, tab_1, seqs, hists, max_seq_len=10, batch=4, rank=0, world=1)
    # ... and so are the recommendation lists (users sharded, gathered everywhere), consistent with the ranks
    t_w, s_w = evaluate.recommend_topk(model, tab_1, [s[:-1] for s in seqs], hists, max_seq_len=10, k=5, batch=4, rank=rank, world=world)
    t_1, s_1 = evaluate.recommend_topk(model, tab_1, [s[:-1] for s in seqs], hists, max_seq_len=10, k=5, batch=4, rank=0, world=1)
    consistent = all(int(t_1[u, int(r_1[u]) - 1]) == seqs[u][-1] for u in range(len(seqs)) if 1 <= int(r_1[u]) <= 5)
    out.update(table_equal=bool(torch.allclose(tab_w, tab_1, rtol=1e-6, atol=1e-7)), ranks_equal=bool(torch.equal(r_w.cpu(), r_1.cpu())),
               n_ranks=int(r_w.numel()),
               topk_equal=bool(torch.equal(t_w.cpu(), t_1.cpu()) and torch.equal(s_w.cpu(), s_1.cpu())) and consistent)
    torch.save(out, os.path.join(os.environ["DP_OUT"], f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
