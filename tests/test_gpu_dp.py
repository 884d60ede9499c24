"""The product's data-parallel step with world > 1 on real GPUs: W fresh processes, one rank each,
`FlatTrainer(world=W).step` through the HIP kernels, gradients exchanged by `torch.distributed`.

* backend "gloo", both ranks on cuda:0 — runs on the 1-GPU box the driver uses;
* backend "nccl" (= RCCL), one GPU per rank — runs wherever >= 2 GPUs are visible.

Reference behaviour being matched: DDP (`Code_Uncached/run.py:287`) = parameters broadcast from rank 0, gradients
averaged over ranks, every rank applies the same Adam step (`run.py:413`); `DistributedSampler` shards (`run.py:146,395`)
stay rank-local as negatives (`model.py:86`)."""
import os
import socket
import subprocess
import sys
import tempfile

import pytest
import torch

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import dp_gpu_worker as W  # noqa: E402
from iisan_amd import ops, trainer  # noqa: E402


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_ranks(world, backend):
    out = tempfile.mkdtemp(prefix="iisan_dp_")
    env = dict(os.environ, WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               DP_BACKEND=backend, DP_OUT=out, HSA_ENABLE_IPC_MODE_LEGACY="0")
    # every rank writes to its own FILE: with pipes drained one after the other, a rank that fills its pipe buffer (RCCL / gloo
    # warnings, a traceback) blocks, its peer waits for it in a collective, and the test hangs to the timeout instead of
    # showing the error
    logs = [open(os.path.join(out, f"rank{r}.log"), "w+") for r in range(world)]
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "dp_gpu_worker.py")], env=dict(env, RANK=str(r)),
                              stdout=logs[r], stderr=subprocess.STDOUT, text=True) for r in range(world)]

    def tail(r):
        logs[r].flush()
        logs[r].seek(0)
        return logs[r].read()[-4000:]

    try:
        for p in procs:
            p.wait(timeout=600)
    except subprocess.TimeoutExpired:
        for q in procs:
            q.kill()
        raise AssertionError("ranks timed out:\n" + "\n".join(f"--- rank {r} ---\n{tail(r)}" for r in range(world)))
    for r, p in enumerate(procs):
        assert p.returncode == 0, "rank %d failed:\n" % r + "\n".join(f"--- rank {q} ---\n{tail(q)}" for q in range(world))
    for f in logs:
        f.close()
    return [torch.load(os.path.join(out, f"rank{r}.pt"), weights_only=False) for r in range(world)]


def _single_rank_grad(rank, start_flat):
    """Gradient of rank `rank`'s shard from a world=1 trainer that starts at the broadcast parameters."""
    dev = torch.device("cuda", 0)
    ids, tc, tt, lm, pop = W.shard_inputs(rank, dev)
    args, model = W.build(seed=99, dev=dev, pop=pop)
    tr = trainer.FlatTrainer(model, args, 1)
    tr.flat.copy_(start_flat.to(dev))
    tr.seg_lr = [0.0] * len(tr.seg_lr)              # forward + backward only: the step must not move the parameters
    tr.step(ids, tc, tt, lm)
    return tr, tr.grad.clone()


def _check(dumps, world):
    d0 = dumps[0]
    # broadcast: ranks were built from different seeds and start from rank 0's parameters
    assert not torch.equal(dumps[1]["before"], d0["before"])
    for d in dumps:
        assert torch.equal(d["start"], d0["before"])
    # every rank holds bit-identical gradients and parameters after each step (DDP's invariant)
    for d in dumps[1:]:
        assert torch.equal(d["grad1"], d0["grad1"]) and torch.equal(d["flat1"], d0["flat1"]) and torch.equal(d["flat2"], d0["flat2"])
    assert not torch.equal(d0["flat1"], d0["start"]) and not torch.equal(d0["flat2"], d0["flat1"])
    # the exchanged gradient is the SUM of the single-rank gradients of the shards ...
    gs = []
    for r in range(world):
        tr, g = _single_rank_grad(r, d0["start"])
        gs.append(g)
    total = torch.stack(gs).sum(0).cpu()
    scale = total.abs().max().item()
    assert scale > 0
    err = (d0["grad1"] - total).abs().max().item() / scale
    assert err < 1e-5, err                           # atomics order only
    # ... and the step is Adam on their MEAN: the fused Adam kernel fed with the exchanged buffer and grad_scale 1/W
    # reproduces the ranks' parameters bit for bit
    dev = torch.device("cuda", 0)
    p = d0["start"].to(dev).clone()
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    ops.adam_step(p, d0["grad1"].to(dev), m, v, d0["seg_end"], d0["seg_lr"], 1, grad_scale=1.0 / world)
    assert torch.equal(p.cpu(), d0["flat1"])
    # and against the textbook formula on the mean gradient (first Adam step: lr * g / (|g| + eps))
    mean = (total / world).double()
    lr = torch.zeros_like(mean)
    lo = 0
    for e, l in zip(d0["seg_end"], d0["seg_lr"]):
        lr[lo:e] = l
        lo = e
    upd = lr * mean / (mean.abs() + 1e-8)
    got = (d0["start"].double() - d0["flat1"].double())
    big = mean.abs() > 1e-6                          # elements whose update does not hinge on summation-order noise
    assert big.sum() > 0.5 * (mean != 0).sum()
    assert (got[big] - upd[big]).abs().max().item() < 1e-3 * max(d0["seg_lr"])
    for d in dumps:
        assert d["table_equal"] and d["ranks_equal"] and d["n_ranks"] == 23 and d["topk_equal"]
        assert all(torch.isfinite(torch.tensor(d["losses"])))


def test_two_ranks_one_gpu_gloo():
    _check(_run_ranks(2, "gloo"), 2)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL refuses two ranks on one device)")
def test_two_ranks_two_gpus_rccl():
    _check(_run_ranks(2, "nccl"), 2)


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` outside a rank environment (it used to exit non-zero): the script starts two
    fresh ranks through torch.distributed.run before touching the GPU, runs warm-up + timed steps of the product DP step
    (barrier + sync on both sides, MAX over ranks, one gradient all-reduce per step) and rank 0 prints one JSON line.  On
    this single-GPU box both ranks share cuda:0 and the collectives go through gloo (env test hook of bench.py)."""
    import json
    env = dict(os.environ, IISAN_BENCH_ONE_GPU="1", IISAN_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    root = os.path.dirname(HERE)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--bs", "16",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["config"]["global_batch"] == 32 and d["config"]["parallelism"] == "dp2"
    # the multi-rank line describes what it ran on: backend, world, every rank's device, and the device time
    # of the one data-path collective per step (HIP events around dp.allreduce_sum_ on every rank)
    di = d["config"]["distributed"]
    assert di["backend"] == "gloo" and di["world"] == 2 and di["devices"] == [0, 0] and len(di["device_uuids"]) == 2
    assert di["allreduces_timed"] == 2 and di["steps_timed"] == 2
    assert len(di["allreduce_ms_per_step"]) == 2 and all(t > 0 for t in di["allreduce_ms_per_step"])
    assert "16." in di["collective"] and "all-reduce" in di["collective"]
    assert d["value"] > 0 and d["ms_per_step"] > 0 and "cpu_baseline" not in d and "secondary" not in d
