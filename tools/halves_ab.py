#!/usr/bin/env python3
"""Would two half-batch image towers on two HIP streams fill each other's tails?  The persistent GEMMs lose their partial last round
(QKV 38.1 rounds -> 39, O / FC2 12.7 -> 13: ~1 ms per step) and every launch its drain; a second, independent kernel stream can use those
CUs — if the towers' persistent workgroups do not trip over each other (profiles/r5_overlap.md: they do at equal priority with a SHORT
second stream).  Times the ViT tower alone: whole batch on one stream / two halves on two streams (own workspaces), interleaved rounds.
    python tools/halves_ab.py [rounds=3] [reps=5]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iisan_amd import _lib, encoders, synth, weights
lib = _lib.load()
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
b = synth.scientific_batch(bs=128, seed=12345, device="cuda", images_on_device=True)
vit = encoders.PackedVit(weights.make_vit_weights(), weights.VIT_BASE, "cuda")
vit.full_blocks = True
sel = [0, 2, 4, 6, 8, 10, 12]
M = b.images.shape[0]
ws2 = encoders._Workspace()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def whole():
    return vit.forward_taps(b.images, sel)


def halves(n_parts=2, streams=(s1, s2)):
    cur = torch.cuda.current_stream()
    outs = []
    edges = [M * i // n_parts for i in range(n_parts + 1)]
    wss = [vit.ws, ws2]
    for i in range(n_parts):
        st = streams[i % 2]
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            old = vit.ws
            vit.ws = wss[i % 2]
            outs.append(vit.forward_taps(b.images[edges[i]:edges[i + 1]], sel))
            vit.ws = old
    for st in streams:
        cur.wait_stream(st)
    return torch.cat(outs)


ref = whole()
got = halves()
torch.cuda.synchronize()
print("halves == whole:", torch.equal(ref, got), flush=True)
for rnd in range(rounds):
    for name, fn in (("whole batch, one stream", whole), ("two halves, two streams", halves)):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        print(f"round {rnd} {name}: {(time.perf_counter() - t0) / reps * 1e3:.3f} ms per ViT forward", flush=True)
