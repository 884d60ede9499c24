#!/usr/bin/env python3
"""Does the operands' row stride cost L2 channel parallelism?  The encoder GEMM with padded leading dimensions:
    python tools/gemm_ld.py "0,8,32,64,128" [rounds] [walk c:h]
Each arm pads lda AND ldw (and ldo) by that many elements; results are compared with the unpadded arm."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iisan_amd import _lib
lib = _lib.load()
pads = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "0,8,64").split(",")]
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
walk = (sys.argv[3] if len(sys.argv) > 3 else "0:0").split(":")
M = 277376
st = torch.cuda.current_stream().cuda_stream
shapes = [("qkv", 2304, 768, 0), ("o", 768, 768, 0), ("fc1", 3072, 768, 1), ("fc2", 768, 3072, 0)]
_lib.dev_set("gemm16_variant", 4)
(_lib.dev_set("gemm16_walk_c", int(walk[0])), _lib.dev_set("gemm16_walk_h", int(walk[1])))
which = os.environ.get("PAD_WHICH", "awo")      # which leading dimensions get the pad
data = {}
for name, N, K, mode in shapes:
    g = torch.Generator(device="cuda").manual_seed(N + K)
    A0 = (torch.randn(M + 256, K, device="cuda", generator=g) * 0.5).half()
    W0 = (torch.randn(N, K, device="cuda", generator=g) * 0.05).half()
    b = torch.randn(N, device="cuda", generator=g)
    for pad in pads:
        pa, pw, po = (pad if "a" in which else 0), (pad if "w" in which else 0), (pad if "o" in which else 0)
        A = torch.zeros(M + 256, K + pa, device="cuda", dtype=torch.float16); A[:, :K] = A0
        W = torch.zeros(N, K + pw, device="cuda", dtype=torch.float16); W[:, :K] = W0
        out = torch.empty(M + 256, N + po, device="cuda", dtype=torch.float16)
        data[(name, pad)] = (A, W, b, out, K + pa, K + pw, N + po)
    del A0, W0


def run(name, N, K, mode, pad, iters):
    A, W, b, out, lda, ldw, ldo = data[(name, pad)]
    for _ in range(iters):
        rc = lib.iisan_gemm16_ld(0, mode, A.data_ptr(), W.data_ptr(), b.data_ptr(), out.data_ptr(), M, N, K, lda, ldw, ldo, st)
        assert rc == 0, lib.iisan_last_error()


for name, N, K, mode in shapes:
    ref = None
    for pad in pads:
        run(name, N, K, mode, pad, 1)
        torch.cuda.synchronize()
        o = data[(name, pad)][3][:M, :N].clone()
        if ref is None: ref = o
        elif not torch.equal(o, ref): print(f"{name} pad {pad}: DIFFERENT", flush=True)
for r in range(rounds):
    for pad in pads:
        row = []
        for name, N, K, mode in shapes:
            run(name, N, K, mode, pad, 2)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run(name, N, K, mode, pad, 10)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 10
            row.append(f"{name} {2.0 * M * N * K / dt / 1e12:6.0f}")
        print(f"round {r} pad {pad:4d} ({which}): " + "  ".join(row), flush=True)
