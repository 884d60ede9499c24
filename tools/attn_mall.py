#!/usr/bin/env python3
"""Does the Infinity Cache (256 MB) pay for the QKV product -> attention pair?  The full-batch pair writes 1.28 GB of head-major
Q / K / V and reads it back from HBM; in chunks of ~146 items (132 MB, one reused buffer) both sides could stay on the die.
Times, same process, interleaved: full batch / chunked, pair and attention alone.   python tools/attn_mall.py
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iisan_amd import _lib  # noqa: E402

lib = _lib.load()
items, S, heads, D = 1408, 197, 12, 768
M = items * S
Mp = (M + 255) // 256 * 256
g = torch.Generator(device="cuda").manual_seed(1)
X = torch.randn(Mp, D, device="cuda", generator=g).half()
W = (torch.randn(3 * D, D, device="cuda", generator=g) * 0.02).half()
b = torch.zeros(3 * D, device="cuda")
rs = torch.ones(Mp, device="cuda")
QKV = torch.empty(items * heads * 3 * S * 64, device="cuda", dtype=torch.float16)
ctx = torch.empty(Mp, D, device="cuda", dtype=torch.float16)
st = torch.cuda.current_stream().cuda_stream


def gemm(m0, n_items, out):
    _lib.check(lib.iisan_gemm16_lna(4, X.data_ptr() + m0 * S * D * 2, W.data_ptr(), b.data_ptr(), out, rs.data_ptr() + m0 * S * 4,
                                    n_items * S, 3 * D, D, S, st), "gemm16_lna")


def attn(m0, n_items, inp):
    _lib.check(lib.iisan_attention16(0, inp, None, ctx.data_ptr() + m0 * S * D * 2, n_items, S, heads, st), "attention16")


def run(chunk, what):
    for m0 in range(0, items, chunk):
        n = min(chunk, items - m0)
        q = QKV.data_ptr() if chunk < items else QKV.data_ptr()
        if what in ("pair", "gemm"):
            gemm(m0, n, q)
        if what in ("pair", "attn"):
            attn(m0, n, q)


def timed(chunk, what, reps=5):
    run(chunk, what)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        run(chunk, what)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


# NOTE: rows of a chunk must start at a multiple of 4 bytes * ... (rowstat pointer offset m0 * S * 4: any m0), A rows at m0*S*1536 B
for rnd in range(3):
    for chunk in (1408, 704, 352, 176, 144, 88):
        t_pair, t_attn, t_gemm = timed(chunk, "pair"), timed(chunk, "attn"), timed(chunk, "gemm")
        print(f"round {rnd} chunk {chunk:5d} items ({chunk * heads * 3 * S * 64 * 2 / 1e6:7.1f} MB of QKV): pair {t_pair:8.1f} us   attention alone {t_attn:8.1f}   "
              f"QKV product alone {t_gemm:8.1f}", flush=True)
