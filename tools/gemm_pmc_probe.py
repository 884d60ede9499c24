#!/usr/bin/env python3
"""One encoder GEMM shape a few times (for rocprofv3 --pmc passes): python tools/gemm_pmc_probe.py [qkv|o|fc1|fc2] [variant]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iisan_amd import _lib
lib = _lib.load()
shape = sys.argv[1] if len(sys.argv) > 1 else "qkv"
var = int(sys.argv[2]) if len(sys.argv) > 2 else 0
M = 277376
N, K, mode = {"qkv": (2304, 768, 0), "o": (768, 768, 0), "fc1": (3072, 768, 1), "fc2": (768, 3072, 0)}[shape]
A = (torch.randn(M + 256, K, device="cuda") * 0.5).half()
W = (torch.randn(N, K, device="cuda") * 0.05).half()
b = torch.randn(N, device="cuda")
out = torch.empty(M + 256, N, device="cuda", dtype=torch.float16)
st = torch.cuda.current_stream().cuda_stream
_lib.dev_set("gemm16_variant", var)
if os.environ.get("GEMM_WALK"):          # "c:h" — tile walk of gemm16_h256
    c, h = os.environ["GEMM_WALK"].split(":")
    (_lib.dev_set("gemm16_walk_c", int(c)), _lib.dev_set("gemm16_walk_h", int(h)))
for _ in range(6):
    lib.iisan_gemm16(0, mode, A.data_ptr(), W.data_ptr(), b.data_ptr(), out.data_ptr(), None, M, N, K, st)
torch.cuda.synchronize()
