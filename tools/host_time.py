#!/usr/bin/env python3
"""Is the Cached / Versa step host-bound?  The same loop timed two ways: host time to ENQUEUE n steps (no sync inside) and wall
time including the final sync; plus a cProfile of the enqueue loop."""
import os, sys, time, cProfile, pstats, io, argparse
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from iisan_amd import _lib, factory, synth, tapstore, trainer
lib = _lib.load()
dev = torch.device("cuda", 0)
n = synth.SCI_ITEM_NUM
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ids_np, log_mask = synth.make_ids(bs, 10, n, np.random.RandomState(12345))
ids = torch.from_numpy(ids_np).view(-1).to(dev); log_mask = torch.from_numpy(log_mask).to(dev)
g = torch.Generator(device=dev).manual_seed(1)
args = factory.make_args()
model = factory.build_model(args, n, synth.make_pop_prob(n), cached=True, device=dev)
layers = model.mm_encoder.packed_layers()
model.tap_stores = tuple(tapstore.TapStore(torch.randn(n + 1, len(layers), 768, generator=g, device=dev).mul_(0.25), range(len(layers)), dev, "fp32") for _ in range(2))
model.train()
tr = trainer.FlatTrainer(model, args, 1)
step = lambda: tr.step(ids, None, None, log_mask)
for _ in range(5): step()
torch.cuda.synchronize()
N = 30
t0 = time.perf_counter()
for _ in range(N): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"bs={bs}: host enqueue {1e3 * (t1 - t0) / N:.3f} ms/step, wall {1e3 * (t2 - t0) / N:.3f} ms/step")
pr = cProfile.Profile(); pr.enable()
for _ in range(N): step()
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:5000])
