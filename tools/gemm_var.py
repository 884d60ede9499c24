#!/usr/bin/env python3
"""Same-process A/B of gemm16 kernel variants on the four ViT-B encoder shapes (M = 277,376 token rows):
    python tools/gemm_var.py [variants, default 3,4] [rounds, default 3]
variant 3 = staggered 256x256 kernel (gemm16_s256.hip), 4 = the same with the half-slot tile boundary (gemm16_h256.hip).
Prints TFLOP/s per shape and round, and checks that the variants' outputs are bit-identical (same accumulation order)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iisan_amd import _lib

lib = _lib.load()
variants = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "3,4").split(",")]
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
M = int(os.environ.get("GEMM_M", 277376))
st = torch.cuda.current_stream().cuda_stream
shapes = [("qkv", 2304, 768, 0), ("o", 768, 768, 0), ("fc1", 3072, 768, 1), ("fc2", 768, 3072, 0)]
data = {}
for name, N, K, mode in shapes:
    g = torch.Generator(device="cuda").manual_seed(N + K)
    A = (torch.randn(M + 256, K, device="cuda", generator=g) * 0.5).half()
    W = (torch.randn(N, K, device="cuda", generator=g) * 0.05).half()
    b = torch.randn(N, device="cuda", generator=g)
    data[name] = (A, W, b, torch.empty(M + 256, N, device="cuda", dtype=torch.float16))


def run(name, N, K, mode, v, iters):
    A, W, b, out = data[name]
    _lib.dev_set("gemm16_variant", v)
    for _ in range(iters):
        rc = lib.iisan_gemm16(0, mode, A.data_ptr(), W.data_ptr(), b.data_ptr(), out.data_ptr(), None, M, N, K, st)
        assert rc == 0, lib.iisan_last_error()
    _lib.dev_set("gemm16_variant", 0)


# bit-identity of the variants
for name, N, K, mode in shapes:
    ref = None
    for v in variants:
        data[name][3].zero_()
        run(name, N, K, mode, v, 1)
        torch.cuda.synchronize()
        o = data[name][3][:M].clone()
        if ref is None:
            ref = o
        else:
            same = torch.equal(o, ref)
            d = (o.float() - ref.float()).abs().max().item()
            print(f"{name}: variant {v} vs {variants[0]}: {'bit-identical' if same else f'DIFFERENT max|d| {d:.3e}'}", flush=True)
for r in range(rounds):
    for v in variants:
        row = []
        for name, N, K, mode in shapes:
            run(name, N, K, mode, v, 3)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run(name, N, K, mode, v, 15)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 15
            row.append(f"{name} {2.0 * M * N * K / dt / 1e12:6.0f}")
        print(f"round {r} variant {v}: " + "  ".join(row), flush=True)
