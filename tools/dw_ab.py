#!/usr/bin/env python3
"""A/B of the weight-gradient kernel (gemm32_dw_kernel, knob dev switch gemm32_dw) on the Cached (bs = 1024) and Versa (bs = 128)
steps, one process, interleaved rounds.  Usage on the GPU box: python tools/dw_ab.py"""
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


for versa in (False, True):
    for rnd in range(3):
        for mode in (0, 1):
            _lib.dev_set("gemm32_dw", mode)
            print(f"{'versa ' if versa else 'cached'} round {rnd} dw={mode}: {run(versa):.3f} ms/step", flush=True)
_lib.dev_set("gemm32_dw", 1)
