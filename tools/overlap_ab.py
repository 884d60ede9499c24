#!/usr/bin/env python3
"""settle the overlapped-towers question with evidence.  One process, one box: the headline step (both towers on one
stream) against the text tower on a second HIP stream (both at normal priority / the image tower at high priority) — 20 timed steps per leg, three interleaved rounds,
no per-launch events inside the timed regions.  Prints a markdown table (copy to profiles/r5_overlap.md).

    python tools/overlap_ab.py [rounds=3] [steps=20]
"""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from iisan_amd import _lib  # noqa: E402

lib = _lib.load()
import torch  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
a = bench.parse(["--no-cpu-baseline", "--no-secondary"])
unc = bench.Uncached(a, lib, dev, 0, 1)
unc.set_full_blocks(True)
enc = unc.model.mm_encoder
lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
print(f"stream priority range (least, greatest): {lo}, {hi}", flush=True)
LEGS = [("one stream (headline)", False, None), ("one stream + HIP events around every gemm16 launch (bench.py's instrumented pass)", False, "events"),
        ("towers on two streams, BOTH normal priority (the round-4 opt-in)", True, "both_normal"),
        ("image tower on a HIGH-priority stream, text tower on a normal one (the opt-in `overlap_towers` as shipped)", True, "product")]
res = {name: [] for name, _, _ in LEGS}
loss = {}
clock = bench.Clock(dev, 1)
for rnd in range(rounds):
    for name, ov, kind in LEGS:
        enc.overlap_towers = ov
        enc._tower_streams = (torch.cuda.Stream(), torch.cuda.Stream()) if kind == "both_normal" else None
        if kind == "events":
            import ctypes as C
            el, ls = clock.run(unc.step, 3, steps, lib, timed=True)
            lib.iisan_timing_collect(C.byref(C.c_double(0)), C.byref(C.c_double(0)))
        else:
            el, ls = clock.run(unc.step, 3, steps)
        res[name].append(el / steps * 1e3)
        loss[name] = float(ls.item())
        print(f"round {rnd} {name}: {el / steps * 1e3:.3f} ms/step (loss {loss[name]:.6f})", flush=True)
enc._tower_streams = None
enc.overlap_towers = False
base = statistics.median(res[LEGS[0][0]])
print("\n| leg | ms/step per round | median | vs headline | items/s | whole-step fraction (x 40.28 GF / 2.5 PF) |\n|---|---|---|---|---|---|")
for name, _, _ in LEGS:
    med = statistics.median(res[name])
    ips = a.bs * 11 / (med * 1e-3)
    print(f"| {name} | {' / '.join(f'{x:.2f}' for x in res[name])} | {med:.2f} | {med - base:+.2f} ms | {ips:.0f} | {ips * bench.FLOP_PER_SLOT / bench.MFMA_PEAK:.4f} |")
