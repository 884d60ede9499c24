#!/usr/bin/env python3
"""Cached / Versa step time against the split-K target of the weight-gradient products and the way their partial sums are
combined (scratch + reducer vs atomics).  Development aid; run on the GPU box."""
import contextlib
import io
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
from iisan_amd import _lib  # noqa: E402

lib = _lib.load()
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
for argv in (["--cached", "fp32"], ["--cached", "fp16", "--versa"]):
    a = bench.parse(argv)
    for via in (1, 0):
        for target in (128, 256, 384, 512, 1024):
            _lib.dev_set("gemm32_accum_scratch", via)
            (_lib.dev_set("gemm32_tm_thresh", 512), _lib.dev_set("gemm32_splitk_target", target))
            with contextlib.redirect_stdout(io.StringIO()):
                ln = bench.cached_line(a, lib, dev, 0, 1, 10, 3)
            print(f"{' '.join(argv):24s} {'scratch' if via else 'atomics'} split-K target {target:5d}: {ln['ms_per_step']:.3f} ms/step", flush=True)
_lib.dev_set("gemm32_accum_scratch", 1)
(_lib.dev_set("gemm32_tm_thresh", 512), _lib.dev_set("gemm32_splitk_target", 1024))
