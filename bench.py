#!/usr/bin/env python3
"""Headline benchmark of the IISAN hot path on MI355X (contract: see the task statement / DESIGN.md §measurement).

Metric (BASELINE.json): items/s (fwd+bwd+Adam) of Uncached IISAN, ViT-B/16 + BERT-base, Amazon-Scientific-shaped
synthetic batches, bs=128 sequences per GPU (1408 item slots per step per GPU), weak scaling over N GPUs.
One "step" = zero_grad -> frozen ViT+BERT forward with CLS taps (HIP, fp16 MFMA operands / fp32 accumulate) ->
side network -> com_dense -> SASRec -> fused in-batch CE -> backward -> (N>1: one RCCL all-reduce of the flat
gradient buffer) -> fused Adam.  Inputs are resident in HBM before the timed region.  ALL 1408 slots are encoded
(padding slots included, like the reference) and EVERY encoder block runs on EVERY token (as HF does; SURVEY 8d's
clean configuration): no dedup, no pruning.  The two legal work-pruning options are separate `secondary` lines.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python bench.py --gpus 8 --steps 10 --warmup 3          # starts its own 8 ranks (torch.distributed.run) when none exist
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

The default single-GPU run prints ONE JSON line: the headline plus `secondary` — the other BASELINE configurations and
ablations timed in the same process (each a few seconds): the CLS-only last block (`--cls-prune`, work pruning), bf16
encoder operands, Code_Cached at bs=1024 (config 3), IISAN-Versa shapes (config 5) and the eval path (users/s).
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_SLOT = 40.28e9          # ViT 35.126 + BERT 5.129 fwd + side net 3x0.008 (fwd+bwd)
MFMA_PEAK = 2.5e15               # dense bf16/f16 MFMA peak (MI355X_MICROARCH.md)
HBM_PEAK = 8.0e12                # HBM3E spec (MI355X_MICROARCH.md; 6.3e12 achievable)
F32_MFMA_PEAK = 157.3e12         # v_mfma_f32_*_f32 dense peak (cdna_hip_programming.md, gfx950 header line)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--bs", type=int, default=None, help="sequences per GPU (default 128; 1024 with --cached, 128 with --versa)")
    ap.add_argument("--dtype", default="fp16", choices=["fp16", "bf16"], help="MFMA operand type of the frozen encoders")
    ap.add_argument("--chunk", type=int, default=0, help="items per encoder chunk (0 = whole batch)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="headline only (no `secondary` list)")
    ap.add_argument("--no-check", action="store_true",
                    help="skip the kernel-family check before the timed region (profiling runs: its extra forward passes would "
                         "be counted among the traced launches)")
    ap.add_argument("--dedup", action="store_true",
                    help="SURVEY 8f-3 (reported separately, never the headline): encode each distinct item id of the batch "
                         "once (padding = id 0) and scatter the taps back; images are then drawn per item id")
    ap.add_argument("--overlap-towers", dest="overlap_towers", action="store_true", default=True,
                    help="(the default since round 6 = the product default `mm_encoder.overlap_towers`) ViT tower on a high-priority HIP stream, "
                         "BERT tower on a normal-priority one beside it — same kernels, bit-identical results, -0.3 .. -0.4 ms per step "
                         "(profiles/r5_overlap.md)")
    ap.add_argument("--no-overlap-towers", dest="overlap_towers", action="store_false",
                    help="both towers back to back on one stream: for PROFILING runs (kernels of two overlapped towers share the CUs, so "
                         "their per-kernel durations — HIP events and rocprofv3 alike — are not kernel measures)")
    ap.add_argument("--cached", choices=["fp32", "fp16", "bf16"], default=None,
                    help="secondary workload (BASELINE config 3, never the headline): Code_Cached IISAN fed from a "
                         "device-resident packed tap store of the given precision; use with --bs 1024")
    ap.add_argument("--versa", action="store_true",
                    help="with --cached: BASELINE config 5 shapes (IISAN-Versa, ViT-L 1024-wide image taps and Llama-3-70B "
                         "8192-wide text taps, Code_Cached_Asym tap lists) instead of config 3")
    ap.add_argument("--x3", type=int, default=1, choices=[0, 1, 2],
                    help="route of the side network's large Linear layers: 1 = product default, 0 = f32 matrix cores only, "
                         "2 = split-operand fp16 GEMM wherever the shape allows (A/B knob)")
    ap.add_argument("--cls-prune", action="store_true",
                    help="work pruning, reported separately (SURVEY 8d): the last executed encoder block computes K/V for all "
                         "tokens but attention/O/MLP for the CLS rows only, since only hidden_states[i][:,0] is consumed (same "
                         "taps).  Default: every block on every token, as HF does")
    ap.add_argument("--full-blocks", action="store_true", help="(the default since round 3; accepted for old command lines)")
    ap.add_argument("--eval", action="store_true",
                    help="secondary workload (SURVEY 8f-2, never the headline): the eval path at Scientific size — users/s of "
                         "evaluate_ranks end to end, of iisan_score_rank alone, and items/s of item_table from cached taps")
    a = ap.parse_args(argv)
    if a.bs is None:
        a.bs = 128 if (a.versa or not a.cached) else 1024
    a.full_blocks = not a.cls_prune
    return a


def launch_command(a, argv, port):
    """`python bench.py --gpus N` without a rank environment starts its own N ranks: fresh processes, one per GPU, created
    BEFORE this process touches the GPU (a process that has initialised HIP must never exec or fork GPU children)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def check_one_device_per_rank(backend, ranks):
    """Under backend nccl (RCCL over xGMI) every rank must own its GPU: two ranks that report the same (host, device uuid /
    index) are a mis-launch, and a line measured that way would be mislabelled — refuse it.  (gloo with every rank on cuda:0
    is the single-GPU test transport and passes.)"""
    # (the device INDEX a rank has selected, per host: uuids are reported but not trusted — a runtime that answers the same or
    #  an empty uuid for every GPU must not make a correct launch look like a collision)
    keys = [(d["host"], d["device"]) for d in ranks]
    if backend == "nccl" and len(set(keys)) != len(ranks):
        raise SystemExit(f"bench.py: backend nccl (RCCL) with two ranks on one device: {ranks}")


class Clock:
    """Barrier + device sync on both sides of the timed region, MAX over ranks (the driver's contract)."""

    def __init__(self, dev, world):
        self.dev, self.world = dev, world

    def sync(self):
        if self.world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def run(self, step, warmup, steps, lib=None, timed=False, cls=1):
        """`timed`: HIP events around every launch of kernel family `cls` (1 = gemm16, 2 = the gemm32.hip family) on its stream."""
        for _ in range(warmup):
            out = step()
        self.sync()
        if lib is not None:
            lib.iisan_timing_enable(cls if timed else 0)
        t0 = time.perf_counter()
        for _ in range(steps):
            out = step()
        self.sync()
        elapsed = time.perf_counter() - t0
        if lib is not None:
            lib.iisan_timing_enable(0)
        if self.world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=self.dev if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed, out


# ---------------------------------------------------------------------------------------------------------------
# Cached / Versa (BASELINE configs 3 and 5)
# ---------------------------------------------------------------------------------------------------------------

def cached_pmc_traffic(a):
    """HBM-side bytes of one whole Cached / Versa step (all kernels) from the committed rocprofv3 PMC passes
    (profiles/pmc_traffic_cached.json, written by tools/pmc_traffic.py --step); null unless this run is that configuration."""
    from iisan_amd import _lib
    path = os.path.join(ROOT, "profiles", "pmc_traffic_cached.json")
    # the committed passes are of ONE configuration: a single rank, the library's own routes (no --x3 override, no dev knobs)
    if not os.path.exists(path) or a.dedup or a.x3 != 1 or int(os.environ.get("WORLD_SIZE", "1")) != 1 or _lib.dev_knobs():
        return None
    with open(path) as f:
        d = json.load(f)
    key = f"{'versa' if a.versa else 'cached'}_{a.cached}_bs{a.bs}"
    return float(d[key]["bytes_per_step"]) if key in d else None


def cached_line(a, lib, dev, rank, world, steps, warmup):
    """Code_Cached IISAN (side network + SASRec + in-batch CE + Adam, fwd+bwd) on taps gathered on the device from a packed
    store.  HBM-bound: algorithmic bytes 43,008 B per item slot (7 layers x 2 modalities x 768 fp32);
    Versa: 2 B x (7 x 1024 + 7 x 8192) = 129,024 B."""
    import numpy as np
    from iisan_amd import factory, synth, tapstore, trainer
    n = synth.SCI_ITEM_NUM
    ids_np, log_mask = synth.make_ids(a.bs, 10, n, np.random.RandomState(12345 + rank))
    ids = torch.from_numpy(ids_np).view(-1).to(dev)
    log_mask = torch.from_numpy(log_mask).to(dev)
    g = torch.Generator(device=dev).manual_seed(1)

    def store(nl, d):       # synthetic catalogue taps drawn on the device (the Versa text store is 2.3 GB)
        t = torch.randn(n + 1, nl, d, generator=g, device=dev, dtype=torch.float16 if a.cached == "fp16" else torch.float32)
        return tapstore.TapStore(t.mul_(0.25), range(nl), dev, a.cached)

    if a.versa:
        args = factory.make_args(text_embedding_dim=8192, image_embedding_dim=1024, side_adapter_vit_list="3,7,11,15,19,23",
                                 side_adapter_bert_list="4,19,34,49,64,79", image_layers=24, text_layers=80)
        model = factory.build_model(args, n, synth.make_pop_prob(n), cached="versa", device=dev)
        lay_cv, lay_tx = model.mm_encoder.packed_layers()
        model.tap_stores = (store(len(lay_cv), 1024), store(len(lay_tx), 8192))
        esz = model.tap_stores[0].table.element_size()
        alg = float(esz * (len(lay_cv) * 1024 + len(lay_tx) * 8192))
        name = "Code_Cached_Asym IISAN-Versa (ViT-L + Llama-3-70B taps)"
    else:
        args = factory.make_args()
        model = factory.build_model(args, n, synth.make_pop_prob(n), cached=True, device=dev)
        layers = model.mm_encoder.packed_layers()
        model.tap_stores = (store(len(layers), 768), store(len(layers), 768))
        alg, name = 43008.0, "Code_Cached IISAN"
    model.train()
    model.dedup_items = bool(a.dedup)
    tr = trainer.FlatTrainer(model, args, world)
    tr.broadcast_params()
    elapsed, loss = Clock(dev, world).run(lambda: tr.step(ids, None, None, log_mask), warmup, steps)
    if not torch.isfinite(loss).item():
        raise SystemExit("bench.py: cached loss is not finite")
    # second pass of the same steps with HIP events around every launch of the dominant kernel family (two event records per
    # launch on a ~100-launch, ~5 ms step are not free: `value` comes from the pass above, without them)
    Clock(dev, world).run(lambda: tr.step(ids, None, None, log_mask), 0, steps, lib, timed=rank == 0, cls=2)
    ms, fl = C.c_double(0), C.c_double(0)
    n_launch = lib.iisan_timing_collect(C.byref(ms), C.byref(fl)) if rank == 0 else 0
    fam_bytes = lib.iisan_timing_last_bytes()
    fam_tf = fl.value / (ms.value * 1e-3) / 1e12 if ms.value > 0 else 0.0
    slots = a.bs * 11
    distinct = int(torch.unique(ids).numel())
    value = slots * world * steps / elapsed
    st = model.tap_stores
    n_params = int(sum(p.numel() for p in model.parameters() if p.requires_grad))
    return {
        "metric": f"items/s (fwd+bwd) {name}, packed device tap store, Scientific-shaped", "value": value,
        "unit": "items/s", "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": elapsed / steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{name}, bs={a.bs}/GPU ({slots} item slots"
                               + (f", side network on the {distinct} distinct item ids only" if a.dedup else "") + f"), tap stores {a.cached} "
                               f"{tuple(st[0].table.shape)} + {tuple(st[1].table.shape)} = "
                               f"{(st[0].nbytes() + st[1].nbytes()) / 1e6:.0f} MB in HBM",
                   "loss": float(loss.item())},
        "roofline": {"bound": "hbm", "achieved": value / world * alg / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                     "frac": value / world * alg / HBM_PEAK,
                     # HBM-side bytes of one WHOLE step from the committed PMC passes of this configuration (null for others)
                     "traffic": cached_pmc_traffic(a), "traffic_algorithmic": alg * slots,
                     # SURVEY 8d's denominator counts the tap reads only; a step also has to move its trainable state —
                     # every parameter read by the forward and by the backward pass, its gradient written and read, and Adam's p / m / v read and
                     # written: 11 fp32 words per parameter — which at Versa's bs = 128 (69.7 M parameters) is ten times the tap bytes
                     "traffic_floor_incl_optimizer": alg * slots + 44.0 * n_params,
                     "trainable_parameters": n_params,
                     "note": f"whole step against the algorithmic {alg:.0f} B/slot of SURVEY 8d (tap reads only); `traffic_floor_incl_optimizer` adds 44 B per "
                             "trainable parameter (weights read forward + backward, gradient written + read, Adam p/m/v read + written)",
                     # what the step is actually bound by: the f32-matrix-core GEMM family of gemm32.hip (SANB products, weight
                     # gradients, heads; HIP events around every launch of the family on its stream inside the timed region)
                     "dominant_kernels": {"kernel": "gemm32.hip family (gemm32_kernel / gemm32_k64_kernel / gemm32_n64f_kernel / "
                                                    "gemm32_dw_kernel + split-K reducers)", "bound": "mfma (f32 inputs)",
                                          "launches": int(n_launch), "launches_per_step": n_launch / max(steps, 1),
                                          "avg_launch_ms": ms.value / max(n_launch, 1), "ms_per_step": ms.value / max(steps, 1),
                                          "flop_per_step": fl.value / max(steps, 1), "achieved": fam_tf, "peak": F32_MFMA_PEAK / 1e12,
                                          "unit": "TFLOP/s", "frac": fam_tf * 1e12 / F32_MFMA_PEAK,
                                          "operand_bytes_per_step": fam_bytes / max(steps, 1)}},
    }


# ---------------------------------------------------------------------------------------------------------------
# Eval path (reference: Code_Uncached/data_utils/metrics.py:59-67,69-107,157-246) — secondary line
# ---------------------------------------------------------------------------------------------------------------

SCI_USERS = 12076                # users of Amazon-Scientific after filtering


def eval_pmc_traffic(a, world):
    """HBM-side bytes per `score_rank_mfma_kernel` launch at Scientific size from the committed PMC passes (profiles/pmc_traffic_eval.json,
    tools/evidence_r4.sh); null unless this is that configuration (one rank, the library's own routes)."""
    from iisan_amd import _lib
    path = os.path.join(ROOT, "profiles", "pmc_traffic_eval.json")
    if not os.path.exists(path) or world != 1 or _lib.dev_knobs():
        return None
    with open(path) as f:
        return float(json.load(f)["avg_bytes_per_launch"])


def eval_line(a, lib, dev, rank, world, reps=5):
    """The path that produces HR@10 at Scientific size: item table [20,315 x 64] from cached taps (`item_table`), then for
    12,076 users SASRec -> scores against every item -> history mask -> exact rank of the target (`evaluate_ranks`:
    host packing of the user sequences included; and its rank kernel `iisan_score_rank` alone, timed with HIP events on the
    launching stream).  Roofline of the rank kernel: the score product [U, 64] x [64, N] on the f32 matrix cores (2*U*N*64
    FLOP; its memory side is the 5.2 MB table + 3.1 MB of user vectors — the [U, N] scores are never materialised)."""
    import numpy as np
    from iisan_amd import evaluate, factory, ops, synth
    n, U = synth.SCI_ITEM_NUM, SCI_USERS
    args = factory.make_args()
    model = factory.build_model(args, n, synth.make_pop_prob(n), cached=True, device=dev)
    model.eval()
    g = torch.Generator(device=dev).manual_seed(3)
    tc = torch.randn(n + 1, 13, 768, generator=g, device=dev).mul_(0.25)      # the reference's [N, 13, 768] fp32 layout
    tt = torch.randn(n + 1, 13, 768, generator=g, device=dev).mul_(0.25)

    def wall(fn, warm=1):
        for _ in range(warm):
            out = fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            out = fn()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        ts.sort()
        return ts[len(ts) // 2], out

    t_tab, table = wall(lambda: evaluate.item_table(model, tc, tt, batch=2048, rank=rank, world=world))
    rs = np.random.RandomState(11 + rank)
    ls = np.array(sorted(synth.SCI_LEN_HIST))
    pr = np.array([synth.SCI_LEN_HIST[int(l)] for l in ls], dtype=np.float64)
    lens = rs.choice(ls, size=U, p=pr / pr.sum())
    seqs = [rs.choice(np.arange(1, n + 1), size=int(l), replace=False).tolist() for l in lens]
    hists = [s_[:-1] for s_ in seqs]
    t_e2e, ranks = wall(lambda: evaluate.evaluate_ranks(model, table, seqs, hists, max_seq_len=10, batch=4096, rank=rank, world=world))
    hit, ndcg = evaluate.hit_ndcg(ranks)
    # the rank kernel alone: one launch over all users, HIP events on torch's current stream (= the stream ops.score_rank
    # hands to the ABI)
    prec = torch.randn(U, 64, generator=g, device=dev)
    hist = torch.zeros(U, 10, dtype=torch.int32, device=dev)
    tgt = torch.randint(1, n + 1, (U,), generator=g, device=dev, dtype=torch.int32)
    for _ in range(2):
        ops.score_rank(prec, table, hist, tgt)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
    for e0, e1 in ev:
        e0.record()
        ops.score_rank(prec, table, hist, tgt)
        e1.record()
    torch.cuda.synchronize()
    ms = sorted(e0.elapsed_time(e1) for e0, e1 in ev)[len(ev) // 2]
    flops = 2.0 * U * n * 64
    # the recommendation list (round 6: `iisan_score_topk`, k = 10): the same score tiles, selection instead of counting — end to end
    # (`evaluate.recommend_topk`: host packing + SASRec + the kernel) and the kernel alone
    t_top, (top_ids, _) = wall(lambda: evaluate.recommend_topk(model, table, hists, hists, max_seq_len=10, k=10, batch=4096, rank=rank, world=world))
    for _ in range(2):
        ops.score_topk(prec, table, hist, 10)
    for e0, e1 in ev:
        e0.record()
        ops.score_topk(prec, table, hist, 10)
        e1.record()
    torch.cuda.synchronize()
    ms_top = sorted(e0.elapsed_time(e1) for e0, e1 in ev)[len(ev) // 2]
    return {
        "metric": "users/s, eval path (SASRec + scores against every item + history mask + exact target rank), Scientific size",
        "value": U * world / t_e2e, "unit": "users/s", "n_gpus": world, "steps": reps, "warmup": 1,
        "ms_per_step": t_e2e * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"evaluate_ranks end to end (host packing of {U} user sequences included), item table "
                               f"[{n + 1}, 64], histories <= 10, user batch 4096",
                   "hit10": hit, "ndcg10": ndcg,
                   "item_table": {"items_per_s": (n + 1) / t_tab, "ms": t_tab * 1e3,
                                  "what": f"item_table: side network + com_dense forward over {n + 1} items from cached taps "
                                          "[N, 13, 768] fp32 x 2 (reference layout), item batch 2048"},
                   "score_rank_alone": {"users_per_s": U / (ms * 1e-3), "ms_per_launch": ms, "launch": f"{U} users x {n + 1} items"},
                   "recommend_topk": {"users_per_s": U * world / t_top, "ms": t_top * 1e3, "k": 10,
                                      "what": "evaluate.recommend_topk end to end: each user's ten best items outside its history (ids + scores)",
                                      "score_topk_alone": {"users_per_s": U / (ms_top * 1e-3), "ms_per_call": ms_top,
                                                           "achieved_tflops": flops / (ms_top * 1e-3) / 1e12,
                                                           "launch": f"{U} users x {n + 1} items, k = 10 (selection kernel + merge of the item splits)"}}},
        "roofline": {"bound": "mfma", "achieved": flops / (ms * 1e-3) / 1e12, "peak": F32_MFMA_PEAK / 1e12, "unit": "TFLOP/s",
                     "frac": flops / (ms * 1e-3) / F32_MFMA_PEAK, "traffic": eval_pmc_traffic(a, world), "kernel": "score_rank_mfma_kernel",
                     "launches": len(ev), "avg_launch_ms": ms, "flop_per_launch": flops,
                     "traffic_algorithmic": float((n + 1) * 64 * 4 + U * 64 * 4 + U * 10 * 4 + U * 8)},
    }


# ---------------------------------------------------------------------------------------------------------------
# Uncached (BASELINE configs 2 and 4) — the headline
# ---------------------------------------------------------------------------------------------------------------

def pmc_traffic(a):
    """Measured memory-side bytes per gemm16 launch of the DEFAULT configuration, from the committed summary of the two PMC
    passes (tools/pmc_traffic.py); PMC counters cannot be read from inside the timed run, so any other configuration
    reports null."""
    # (counter passes serialise the kernels: the per-launch bytes are those of the towers back to back, whichever way the step schedules them)
    default = (a.bs == 128 and a.dtype == "fp16" and a.full_blocks and not a.dedup
               and not a.cached and a.chunk == 0)
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not default or not os.path.exists(path):
        return None
    with open(path) as f:
        d = json.load(f)
    # the committed passes must be of THIS default (round 3 flipped it to every block on every token)
    # ... and of THIS kernel set (round 4, second half: the O / FC2 products read and write the residual stream in their epilogues — dev switch `ln_fold` 2)
    return float(d["avg_bytes_per_launch"]) if d.get("encoder_blocks") == "all tokens in every block" and d.get("ln_fold") == 2 else None


class Uncached:
    def __init__(self, a, lib, dev, rank, world):
        from iisan_amd import factory, synth, trainer, weights
        self.a, self.lib, self.dev, self.rank, self.world = a, lib, dev, rank, world
        torch.manual_seed(20260 + rank)        # trainable init and the SASRec dropout stream: the reported loss is reproducible
        self.args = factory.make_args()
        self.batch = synth.scientific_batch(bs=a.bs, seed=12345 + rank, device=dev, images_on_device=True, images_by_item=a.dedup)
        vit_w, bert_w = weights.make_vit_weights(), weights.make_bert_weights()
        self.model = factory.build_model(self.args, synth.SCI_ITEM_NUM, self.batch.pop_prob.cpu(), vit_w, weights.VIT_BASE,
                                         bert_w, weights.BERT_BASE, cached=False, device=dev)
        enc = self.model.mm_encoder
        enc.cv_encoder.chunk_items = a.chunk
        enc.bert_encoder.text_encoders["title"].chunk_items = a.chunk
        self.model.dedup_items = a.dedup
        enc.overlap_towers = a.overlap_towers
        self.model.train()
        self.tr = trainer.FlatTrainer(self.model, self.args, world)
        self.tr.broadcast_params()
        self.ids = self.batch.ids.view(-1)

    def set_full_blocks(self, on):
        """Dead-work policy of both towers through the weights structs' `full_blocks` field (include/iisan_hip.h)."""
        enc = self.model.mm_encoder
        enc.cv_encoder.full_blocks = bool(on)
        enc.bert_encoder.text_encoders["title"].full_blocks = bool(on)

    def set_dtype(self, name):
        from iisan_amd import encoders
        enc = self.model.mm_encoder
        for m in (enc.cv_encoder, enc.bert_encoder.text_encoders["title"]):
            if m.dtype16 != encoders.DTYPE_NAMES[name]:
                m.dtype16 = encoders.DTYPE_NAMES[name]
                m._packed = None                       # repack the frozen weights in the new operand type

    def step(self):
        b = self.batch
        return self.tr.step(self.ids, b.images, b.text, b.log_mask)

    def kernel_family_check(self):
        """The bench batch through the product's GEMM dispatch (the persistent 256x256 kernels at this size) vs every encoder
        GEMM forced onto the 128x128 v1 kernels, which the GPU tests pin to the reference's golden taps: the CLS taps of all
        1,408 slots must agree within 16-bit accumulation-order noise (4e-4 per layer, far inside the 1.5e-3 tap budget) and
        the forward loss within the north-star 1e-3.  eval mode (no SASRec dropout), no parameter update."""
        from iisan_amd import _lib
        b, loss, taps = self.batch, {}, {}
        enc = self.model.mm_encoder
        need = sorted(set([0] + list(enc.side_cv_adapter_num_list)))
        self.model.eval()
        self.set_full_blocks(self.a.full_blocks)
        # round 4, second half: at this size the ViT tower applies its LayerNorms and residual adds in the GEMM epilogues (dev switch `ln_fold`, default 2),
        # a different ROUNDING SEQUENCE from the 128x128 kernels' LayerNorm images.  Three legs: "img" = the production kernels on the
        # image route (kernel families alone differ: 4e-4), "auto" = the product default (what the timed steps run: inside the 1.5e-3 tap
        # budget against the pinned kernels, loss within the north-star 1e-3), "v1" = the pinned 128x128 kernels.
        try:
            with torch.no_grad():
                for leg, v, fold in (("auto", 0, 2), ("img", 0, 0), ("v1", 1, 2)):
                    with _lib.dev(gemm16_variant=v, ln_fold=fold):
                        loss[leg] = float(self.model(self.ids, b.images, b.text, b.log_mask, None).item())
                        taps[leg] = (enc.cv_encoder.forward_taps(b.images, need), enc.bert_encoder.forward_taps(b.text, need))
        finally:
            self.set_full_blocks(False)
            self.model.train()

        def worst(leg):
            w = 0.0
            for t0, t1 in zip(taps[leg], taps["v1"]):
                for k in range(1, len(need)):
                    w = max(w, ((t0[:, k] - t1[:, k]).double().norm() / t1[:, k].double().norm()).item())
            return w
        rel = abs(loss["auto"] - loss["v1"]) / abs(loss["v1"])
        rel_img = abs(loss["img"] - loss["v1"]) / abs(loss["v1"])
        tap_rel, tap_img = worst("auto"), worst("img")
        if not (rel < 1e-3 and rel_img < 1e-3 and tap_img < 4e-4 and tap_rel < 1.5e-3):
            raise SystemExit(f"bench.py: production GEMM dispatch vs 128x128 kernels: loss {loss['auto']} / {loss['img']} vs {loss['v1']} (rel {rel:.2e} / {rel_img:.2e}), "
                             f"worst tap layer rel {tap_rel:.2e} (default route) / {tap_img:.2e} (LayerNorm images)")
        return {"loss_auto_dispatch": loss["auto"], "loss_v1_kernels": loss["v1"], "loss_rel": rel, "worst_tap_layer_rel": tap_rel,
                "worst_tap_layer_rel_layernorm_images": tap_img}

    def dist_info(self, steps):
        """What a multi-rank line actually ran on: backend, world size, every rank's device, and the
        device time of the one data-path collective per step.  Two ranks on one device under backend nccl (RCCL) would be a
        mis-launch: refuse to print a line for it."""
        p = torch.cuda.get_device_properties(self.dev)
        mine = {"rank": self.rank, "host": socket.gethostname(), "device": self.dev.index, "name": p.name,
                "uuid": str(getattr(p, "uuid", "")), "pci_bus_id": getattr(p, "pci_bus_id", None)}
        n_ar, ar_ms = self.tr.allreduce_ms()
        mine["allreduce_ms_per_step"] = ar_ms / max(n_ar, 1)
        every = [None] * self.world
        dist.all_gather_object(every, mine)
        backend = dist.get_backend()
        check_one_device_per_rank(backend, every)
        return {"backend": backend, "world": dist.get_world_size(), "devices": [d["device"] for d in every],
                "device_uuids": [d["uuid"] for d in every], "hosts": sorted(set(d["host"] for d in every)),
                "collective": f"one SUM all-reduce of the flat gradient buffer per step ({self.tr.grad.numel() * 4 / 1e6:.2f} MB fp32)",
                "allreduces_timed": n_ar, "steps_timed": steps,
                "allreduce_ms_per_step": [round(d["allreduce_ms_per_step"], 4) for d in every],
                "allreduce_ms_per_step_max": max(d["allreduce_ms_per_step"] for d in every)}

    def line(self, steps, warmup, dtype="fp16", full_blocks=True, headline=True, overlap=None):
        a, lib, world = self.a, self.lib, self.world
        self.set_dtype(dtype)
        self.set_full_blocks(full_blocks)
        enc = self.model.mm_encoder
        prev_overlap = enc.overlap_towers
        if overlap is not None:
            enc.overlap_towers = overlap
        clock = Clock(self.dev, world)
        try:
            # Two passes of the same K steps (round 5; the Cached line has always done this): `value` / `ms_per_step` come from a pass
            # with NO instrumentation inside the timed region, scheduled as the product schedules it (round 6: the two towers on two
            # HIP streams).  The per-launch HIP events of `roofline.dominant_kernel` (194 event records per step on the launching
            # stream) and the all-reduce events of a multi-rank run ride on a second pass with the towers BACK TO BACK on one stream:
            # kernels of two overlapped towers share the CUs and their durations are not kernel measures
            # (profiles/r4_overlap_by_stream.md); the events themselves cost +0.34 ms per step (profiles/r5_overlap.md).
            elapsed, loss = clock.run(self.step, warmup, steps)
            overlapped = enc.overlap_towers
            enc.overlap_towers = False
            if world > 1:
                self.tr.time_allreduce = True
                self.tr.allreduce_ms()
            elapsed_ev, _ = clock.run(self.step, 0, steps, lib, timed=self.rank == 0)
            self.tr.time_allreduce = False
        finally:
            self.set_full_blocks(False)
            enc.overlap_towers = prev_overlap
            self.tr.time_allreduce = False
        dinfo = self.dist_info(steps) if world > 1 else None
        if not torch.isfinite(loss).item():
            raise SystemExit("bench.py: loss is not finite")
        ms, fl = C.c_double(0), C.c_double(0)
        n_launch = lib.iisan_timing_collect(C.byref(ms), C.byref(fl)) if self.rank == 0 else 0
        slots = a.bs * 11
        value = slots * world * steps / elapsed
        gemm_tflops = fl.value / (ms.value * 1e-3) / 1e12 if ms.value > 0 else 0.0
        traffic = pmc_traffic(a) if (headline and dtype == a.dtype and full_blocks == a.full_blocks) else None
        return {
            "metric": "items/s (fwd+bwd) ViT-B+BERT-B IISAN uncached, Scientific, bs=128",
            "value": value, "unit": "items/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": dtype, "data": "synthetic",
            "config": {"workload": "Code_Uncached IISAN ViT-base+BERT-base, Amazon-Scientific-shaped synthetic batch, "
                                   f"bs={a.bs}/GPU ({slots} item slots, " + ("distinct item ids encoded once" if a.dedup else "all encoded") + "), 1xMI355X per rank",
                       "global_batch": a.bs * world, "parallelism": f"dp{world}", "loss": float(loss.item()),
                       **({"distributed": dinfo} if dinfo else {}),
                       "towers": ("image tower on a high-priority HIP stream, text tower on a normal-priority one beside it (the product default: same kernels, "
                                  "bit-identical results); `roofline.dominant_kernel` is timed in a second pass with the towers back to back on one stream") if overlapped
                                 else "both towers back to back on one stream (--no-overlap-towers: the profiling form)",
                       "encoder_blocks": "all tokens in every block" if full_blocks else
                                         "WORK PRUNING: last block computes K/V for all tokens, attention/O/MLP for the CLS rows only (only hidden_states[i][:,0] is consumed; taps identical)"},
            # SURVEY 8d's definition: the WHOLE forward+backward step against the dense 16-bit MFMA peak —
            # items/s per GPU x 40.28 GFLOP of algorithmic work per item slot (ViT 35.126 + BERT 5.129 forward + the side network
            # forward and backward) / 2.5 PFLOP/s.  north_star's target is frac >= 0.40 (24,827 items/s).  With work pruning
            # (`encoder_blocks` says so) the skipped last-block work counts as done: such a line is never the headline.
            "roofline": {"bound": "mfma", "achieved": value / world * FLOP_PER_SLOT / 1e12, "peak": MFMA_PEAK / 1e12, "unit": "TFLOP/s",
                         "frac": value / world * FLOP_PER_SLOT / MFMA_PEAK,
                         "definition": "whole step (SURVEY 8d): items/s per GPU x 40.28e9 FLOP per item slot / 2.5e15",
                         "executed_gemm_flops_frac": fl.value / elapsed_ev / MFMA_PEAK,
                         # bytes per launch of the dominant kernel at the L2's memory side: NOT measured by this run (PMC counters
                         # cannot be read from inside it) but read from the committed summary of two rocprofv3 --pmc passes of this
                         # command with this configuration; null for any other configuration
                         "traffic": traffic, "traffic_source": "profiles/pmc_traffic.json (committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, corrected per MI355X_MICROARCH.md; per gemm16 launch)" if traffic is not None else None,
                         # the dominant kernel class, measured live with HIP events on the launching stream around every launch
                         "dominant_kernel": {
                             "kernel": "gemm16 (gemm16_h256_kernel: QKV/O/FC1/FC2 GEMMs of the frozen encoders"
                                       "; flops = executed, by launch)",
                             "bound": "mfma", "achieved": gemm_tflops, "peak": MFMA_PEAK / 1e12, "unit": "TFLOP/s",
                             "frac": gemm_tflops / (MFMA_PEAK / 1e12),
                             "launches": int(n_launch), "avg_launch_ms": ms.value / max(n_launch, 1),
                             "flop_per_launch": fl.value / max(n_launch, 1),
                             "ms_per_step": ms.value / max(steps, 1),
                             "measured_in": f"a second pass of the same {steps} steps, towers back to back on one stream, HIP events around every launch ({elapsed_ev / steps * 1e3:.3f} ms per step that way)",
                             "traffic": traffic, "traffic_algorithmic": lib.iisan_timing_last_bytes() / max(n_launch, 1)}},
        }


# ---------------------------------------------------------------------------------------------------------------
# CPU baseline
# ---------------------------------------------------------------------------------------------------------------

def host_cpu():
    """(physical cores this process may use, CPU model string) from /proc/cpuinfo and the affinity mask."""
    model, cores, phys, core = "unknown", set(), None, None
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                k, _, v = ln.partition(":")
                k, v = k.strip(), v.strip()
                if k == "model name":
                    model = v
                elif k == "physical id":
                    phys = v
                elif k == "core id":
                    core = v
                elif not k and phys is not None:
                    cores.add((phys, core))
        if phys is not None:
            cores.add((phys, core))
    except OSError:
        pass
    allowed = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    n = min(len(cores), allowed) if cores else allowed
    return max(1, n), model


def cpu_baseline(seed=5, bs=16, warm=3, timed=5, budget_s=100.0):
    """The CPU oracle (a port of the reference path, pinned against the reference's golden vectors) timed on this host:
    BASELINE config 1 = Uncached IISAN, bs=16 (176 item slots), fp32, fwd+bwd+Adam, on all physical cores; warm-up steps,
    then the MEDIAN of the timed steps (3 + 5; fewer only if the time budget runs out, and said so)."""
    import statistics
    from iisan_amd import synth, weights
    from oracle import iisan_oracle as O
    phys, cpu_model = host_cpu()
    # torch-CPU at these sizes stops scaling long before a 2-socket host is full: measured on the pool's 2 x EPYC 9575F
    # (128 physical cores) 7.3 items/s on 128 threads against 15.4-16.7 on 32 — the baseline uses what is fastest
    cores = min(phys, 32)
    torch.set_num_threads(cores)
    vw, bw = weights.make_vit_weights(), weights.make_bert_weights()
    P = {k: v.clone().requires_grad_(True) for k, v in weights.make_trainable_params(seed=99).items()}
    layers = O.side_layer_list("1,3,5,7,9,11", False)
    m = {k: torch.zeros_like(v) for k, v in P.items()}
    v2 = {k: torch.zeros_like(v) for k, v in P.items()}
    b = synth.scientific_batch(bs=bs, seed=seed)

    def step(i):
        t0 = time.perf_counter()
        with torch.no_grad():
            tc = O.vit_cls_taps(b.images, vw, weights.VIT_BASE)
            tt = O.bert_cls_taps(b.text, bw, weights.BERT_BASE)
        loss, _ = O.model_loss_from_taps(b.ids, tc, tt, b.log_mask, b.pop_prob, P, layers)
        for p in P.values():
            p.grad = None
        loss.backward()
        with torch.no_grad():
            for k, p in P.items():
                new, m[k], v2[k] = O.adam_step(p, p.grad, m[k], v2[k], i + 1, 1e-4)
                p.copy_(new)
        return time.perf_counter() - t0

    t_all, i, warm_t, times = time.perf_counter(), 0, [], []
    for _ in range(warm):
        warm_t.append(step(i))
        i += 1
        if (time.perf_counter() - t_all) > 0.3 * budget_s:        # a slow host: keep most of the budget for timed steps
            break
    for _ in range(timed):
        times.append(step(i))
        i += 1
        if (time.perf_counter() - t_all) + times[-1] > budget_s and len(times) >= 2:
            break
    med = statistics.median(times)
    return {"value": bs * 11 / med, "unit": "items/s", "cores": cores, "kind": "port", "cpu_model": cpu_model,
            "physical_cores": phys,
            "sample": f"oracle/iisan_oracle.py, BASELINE config 1: Uncached IISAN bs={bs} ({bs * 11} slots), fp32 torch-CPU on "
                      f"{cores} threads of '{cpu_model}' ({phys} physical cores visible; more threads run slower); "
                      f"{len(warm_t)} warm-up + {len(times)} timed "
                      f"fwd+bwd+Adam steps, median {med:.2f} s (min {min(times):.2f}, max {max(times):.2f})"}


# ---------------------------------------------------------------------------------------------------------------

def secondary_lines(a, unc, lib, dev, rank, world):
    """The other BASELINE configurations / ablations, a few seconds each, driver-timed inside the default run."""
    out = []
    k, w = min(a.steps, 10), min(a.warmup, 3)

    def add(name, fn):
        try:
            ln = fn()
            ln["name"] = name
            out.append(ln)
        except Exception as e:       # a secondary figure must never take the headline down with it
            out.append({"name": name, "error": f"{type(e).__name__}: {e}"})

    add("uncached with WORK PRUNING (SURVEY 8d: reported separately, never the headline): the last encoder block runs attention/O/MLP "
        "on the CLS rows only (taps identical)", lambda: unc.line(k, w, "fp16", False, headline=False))
    add("uncached, bf16 encoder operands (misses the 1e-3 parity tolerance, DESIGN 3)", lambda: unc.line(k, w, "bf16", True, headline=False))
    unc.set_dtype(a.dtype)
    add("uncached with both towers back to back on one stream (`mm_encoder.overlap_towers = False`, the form every profile under profiles/ is taken in)",
        lambda: unc.line(max(k, min(a.steps, 10)), max(w, min(a.warmup, 3)), a.dtype, True, headline=False, overlap=False))
    unc.set_dtype(a.dtype)
    c3 = argparse.Namespace(**{**vars(a), "cached": "fp32", "versa": False, "bs": 1024})
    add("BASELINE config 3: Code_Cached IISAN bs=1024, fp32 tap store", lambda: cached_line(c3, lib, dev, rank, world, 10, 3))
    c3d = argparse.Namespace(**{**vars(c3), "dedup": True})
    add("Code_Cached IISAN bs=1024 with the side network on DISTINCT item ids only (opt-in `model.dedup_items`, loss bit-identical; "
        "items/s still counts all slots — reported separately, not the config-3 figure)", lambda: cached_line(c3d, lib, dev, rank, world, 10, 3))
    torch.cuda.empty_cache()
    c5 = argparse.Namespace(**{**vars(a), "cached": "fp16", "versa": True, "bs": 128})
    add("BASELINE config 5 shapes on one GPU: IISAN-Versa bs=128, fp16 tap stores", lambda: cached_line(c5, lib, dev, rank, world, 10, 3))
    c5d = argparse.Namespace(**{**vars(c5), "dedup": True})
    add("IISAN-Versa bs=128 with the side network on DISTINCT item ids only (opt-in, loss bit-identical; reported separately)",
        lambda: cached_line(c5d, lib, dev, rank, world, 10, 3))
    torch.cuda.empty_cache()
    add("eval path at Scientific size (SURVEY 8f-2): evaluate_ranks / iisan_score_rank users/s, item_table items/s",
        lambda: eval_line(a, lib, dev, rank, world))
    return out


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no GPU call has happened in this process: start one fresh rank per GPU and pass their exit code on
        raise SystemExit(subprocess.call(launch_command(a, sys.argv[1:], _free_port())))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}")
    # Test hook (tests/test_gpu_dp.py): IISAN_BENCH_ONE_GPU=1 puts every rank on cuda:0 and IISAN_BENCH_BACKEND=gloo moves the
    # two collectives through the host — the only way to execute the multi-rank path on a single-GPU box (RCCL refuses two
    # ranks on one device).  Production: one GPU per rank, backend "nccl" (= RCCL over xGMI).
    if os.environ.get("IISAN_BENCH_ONE_GPU") == "1":
        local = 0
    backend = os.environ.get("IISAN_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from iisan_amd import _lib
    lib = _lib.load()
    if a.x3 != 1:                       # 1 = the library's own default: leave the knob untouched
        _lib.dev_set("x3", a.x3)
    if a.eval:
        out = eval_line(a, lib, dev, rank, world)
    elif a.cached:
        out = cached_line(a, lib, dev, rank, world, a.steps, a.warmup)
    else:
        unc = Uncached(a, lib, dev, rank, world)
        check = unc.kernel_family_check() if (world == 1 and a.bs >= 64 and not a.dedup and not a.no_check) else None
        out = unc.line(a.steps, a.warmup, a.dtype, a.full_blocks)
        if check:
            out["config"]["kernel_family_check"] = check
        default = (world == 1 and a.bs == 128 and a.dtype == "fp16" and a.full_blocks and not a.dedup
                   and a.overlap_towers and a.chunk == 0)
        if default and not a.no_secondary:
            out["secondary"] = secondary_lines(a, unc, lib, dev, rank, world)
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
    if _lib.dev_knobs():                 # a run with re-routed kernels must say so in its own line
        out.setdefault("config", {})["dev_knobs"] = _lib.dev_knobs()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
