#!/usr/bin/env python3
"""A/B of a 0/1 route knob of the library (default: the gate-folded dF product, dev switch gemm32_k64_gate; `python tools/gate_ab.py
gemm32_n64f` for the fusion-fed down projection) on the Cached step
(bs = 1024), all slots and distinct ids, and on the Versa step; one process, interleaved rounds."""
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

KNOB = sys.argv[1] if len(sys.argv) > 1 else "gemm32_k64_gate"
MODES = [int(x) for x in sys.argv[2:]] or [0, 1]
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)


def run(extra, steps=20):
    a = bench.parse(["--cached"] + extra)
    with contextlib.redirect_stdout(io.StringIO()):
        ln = bench.cached_line(a, lib, dev, 0, 1, steps, 3)
    return ln["ms_per_step"]


for name, extra in (("cached", ["fp32"]), ("dedup ", ["fp32", "--dedup"]), ("versa ", ["fp16", "--versa"])):
    for rnd in range(3):
        for mode in MODES:
            _lib.dev_set(KNOB, mode)
            print(f"{name} round {rnd} {KNOB}={mode}: {run(extra):.3f} ms/step", flush=True)
_lib.dev_set(KNOB, 1)
