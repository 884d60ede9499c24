#!/usr/bin/env python3
"""Same-process A/B of the headline step with / without the start-time de-synchronisation of the QKV and FC1 GEMMs."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
from iisan_amd import _lib  # noqa: E402

lib = _lib.load()
torch.cuda.set_device(0)
a = bench.parse(["--no-cpu-baseline", "--no-secondary", "--no-check"])
unc = bench.Uncached(a, lib, torch.device("cuda", 0), 0, 1)
for rnd in range(4):
    for on in (0, 1):
        _lib.dev_set("gemm16_desync", on)
        ln = unc.line(8, 2, "fp16", False, headline=False)
        print(f"round {rnd} desync={on}: {ln['ms_per_step']:.2f} ms/step  gemm16 {ln['roofline']['achieved']:.0f} TF", flush=True)
_lib.dev_set("gemm16_desync", 0)
