#!/usr/bin/env python3
"""A/B the Cached (bs=1024) and Versa (bs=128) steps under library knobs, one process, interleaved rounds.
Usage on the GPU box: python tools/step_ab.py"""
import argparse
import contextlib
import io
import json
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

SETTINGS = [("default", 192, 1024), ("tm512", 512, 1024), ("tm768", 768, 1024), ("sk512", 192, 512), ("sk2048", 192, 2048),
            ("tm512 sk512", 512, 512)]


def run(versa, steps=10):
    a = bench.parse(["--cached", "fp16" if versa else "fp32"] + (["--versa"] if versa else []))
    with contextlib.redirect_stdout(io.StringIO()):
        ln = bench.cached_line(a, lib, dev, 0, 1, steps, 3)
    return ln["ms_per_step"]


for versa in (False, True):
    for rnd in range(2):
        for name, tm, sk in SETTINGS:
            (_lib.dev_set("gemm32_tm_thresh", tm), _lib.dev_set("gemm32_splitk_target", sk))
            print(f"{'versa ' if versa else 'cached'} round {rnd} {name:14s} {run(versa):.3f} ms/step", flush=True)
(_lib.dev_set("gemm32_tm_thresh", 0), _lib.dev_set("gemm32_splitk_target", 0))
