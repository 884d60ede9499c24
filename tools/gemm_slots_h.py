#!/usr/bin/env python3
"""Slot timeline of workgroup 0 of gemm16_h256 (half-slot tile boundary): cycles between consecutive stamps of one whole
tile, both groups.  Needs a library built with -DS256_TIMELINE (tools/lib_timeline.so); debug bit 16 dumps the stamps.
    python tools/gemm_slots_h.py [debug bits ...]      (0 = full, 1 = epilogue work skipped)
Stamps per tile (K = 768, 12 K-steps): first step 7 (Rlo end, barrier passed, Mlo end, barrier passed, Rhi+Ehi end,
barrier passed, Mhi end), ten middle steps x 3 (R end, barrier passed, M end), last step 7."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iisan_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib_timeline.so")
lib = _lib.load()
M = 277376
shape = os.environ.get("SHAPE", "qkv")
N, K, mode = {"qkv": (2304, 768, 0), "o": (768, 768, 0), "fc1": (3072, 768, 1), "fc2": (768, 3072, 0)}[shape]
A = (torch.randn(M + 256, K, device="cuda") * 0.5).half()
W = (torch.randn(N, K, device="cuda") * 0.05).half()
b = torch.randn(N, device="cuda")
out = torch.empty(M + 256, N, device="cuda", dtype=torch.float16)
st = torch.cuda.current_stream().cuda_stream
nk = K // 64
F = ["Rlo", "bar", "Mlo", "bar", "Rhi+E", "bar", "Mhi"]
labels = [f"F.{x}" for x in F] + [f"m{k}.{x}" for k in range(1, nk - 1) for x in ("R", "bar", "M")] + [f"L.{x}" for x in F]
per_tile = len(labels)
for dbg in ([int(a) for a in sys.argv[1:]] or [1, 0]):
    _lib.dev_set("gemm16_variant", 4 + ((dbg | 16) << 8))
    for _ in range(2):
        lib.iisan_gemm16(0, mode, A.data_ptr(), W.data_ptr(), b.data_ptr(), out.data_ptr(), None, M, N, K, st)
        torch.cuda.synchronize()
    t = out.view(-1).view(torch.int32)[8192:8192 + 2048].cpu().view(2, 1024).long()
    print(f"== {shape} debug {dbg}: cycles from the previous stamp, tiles 2 and 3 of workgroup 0")
    for g in range(2):
        for tile in (2, 3):
            base = tile * per_tile
            x = t[g, base - 1:base + per_tile].tolist()
            d = [x[i + 1] - x[i] for i in range(per_tile)]
            print(f" group {'AB'[g]} tile {tile}: total {x[-1] - x[0]}")
            print("   " + " ".join(f"{l}={v}" for l, v in zip(labels, d)))
_lib.dev_set("gemm16_variant", 0)
