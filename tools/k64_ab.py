#!/usr/bin/env python3
"""A/B of the K = 64 wide-N kernel (gemm32_k64_kernel) against the tiled f32-core GEMM on the Versa step (bs = 128) and on the
Cached step with the fused SANB launch switched off (the route that uses these products at M = 11,264), one process, interleaved
rounds.  Usage on the GPU box: python tools/k64_ab.py"""
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


def run(versa, steps=20):
    a = bench.parse(["--cached", "fp16" if versa else "fp32"] + (["--versa"] if versa else []))
    with contextlib.redirect_stdout(io.StringIO()):
        ln = bench.cached_line(a, lib, dev, 0, 1, steps, 3)
    return ln["ms_per_step"]


modes = [int(x) for x in sys.argv[1:]] or [0, 1]
for rnd in range(3):
    for mode in modes:
        _lib.dev_set("gemm32_k64", mode)
        print(f"round {rnd} versa k64={mode}: {run(True):.3f} ms/step", flush=True)
_lib.dev_set("sanb_fused", 0)
for rnd in range(2):
    for mode in modes:
        _lib.dev_set("gemm32_k64", mode)
        print(f"round {rnd} cached (unfused SANB) k64={mode}: {run(False):.3f} ms/step", flush=True)
_lib.dev_set("sanb_fused", 1)
_lib.dev_set("gemm32_k64", 1)
