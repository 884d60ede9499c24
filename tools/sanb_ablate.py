#!/usr/bin/env python3
"""Ablation of the fused SANB step kernels at the Cached batch size: whole step with the two products switched off in turn
(results are wrong with a bit set: timing only).  Run under rocprofv3 --kernel-trace to read the kernels' own durations."""
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
a = bench.parse(["--cached", "fp32"])
for bits, name in ((0, "full"), (1, "no narrow product"), (2, "no wide product"), (3, "no products (HBM phases only)")):
    _lib.dev_set("sanb_debug", bits)
    with contextlib.redirect_stdout(io.StringIO()):
        ln = bench.cached_line(a, lib, dev, 0, 1, 10, 3)
    print(f"{name:32s} {ln['ms_per_step']:.3f} ms/step", flush=True)
_lib.dev_set("sanb_debug", 0)
for persist, units in ((0, 0), (1, 0), (1, 2), (1, 4), (1, 6), (1, 8), (1, 12)):
    (_lib.dev_set("sanb_persistent", persist), _lib.dev_set("sanb_stagger", units))
    with contextlib.redirect_stdout(io.StringIO()):
        ln = bench.cached_line(a, lib, dev, 0, 1, 10, 3)
    print(f"persistent={persist} stagger={units:2d}  {ln['ms_per_step']:.3f} ms/step", flush=True)
(_lib.dev_set("sanb_persistent", 1), _lib.dev_set("sanb_stagger", 0))
