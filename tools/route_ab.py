#!/usr/bin/env python3
"""A/B of the fused SANB launches (dev switch sanb_fused 2 = always) against the unfused route (0) on the Cached step (bs = 1024),
all slots and distinct ids only; one process, interleaved rounds.  (Headline: IISAN_DEV_KNOBS="sanb_fused=2|0" python bench.py.)
Usage on the GPU box: python tools/route_ab.py"""
import contextlib
import io
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from iisan_amd import _lib  # noqa: E402

lib = _lib.load()
import torch  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)


def run(extra, steps=20):
    a = bench.parse(["--cached", "fp32"] + extra)
    with contextlib.redirect_stdout(io.StringIO()):
        ln = bench.cached_line(a, lib, dev, 0, 1, steps, 3)
    return ln["ms_per_step"]


for extra in ([], ["--dedup"]):
    for rnd in range(3):
        for fused in (2, 0):
            _lib.dev_set("sanb_fused", fused)
            print(f"cached {' '.join(extra):8s} round {rnd} fused={fused}: {run(extra):.3f} ms/step", flush=True)
_lib.dev_set("sanb_fused", 1)
