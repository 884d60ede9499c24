#!/usr/bin/env python3
"""Same-process A/B of tile walks of gemm16_h256 (dev switch gemm16_walk) on the ViT-B encoder shapes (M = 277,376 token rows):
    python tools/gemm_walk.py "0:0,3:0,3:16,4:16" [rounds] [shapes, default qkv,fc1] [debug bits, e.g. 64 = nt stores]
Each walk is c:h (panel width in column tiles : sub-slab height in row tiles; 0:0 = the row-major list).  A "+64" suffix on a walk
adds debug bits for that arm (e.g. 4:16+64).  Prints TFLOP/s per shape and round (interleaved) and checks bit-identity against the first walk."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iisan_amd import _lib

lib = _lib.load()
arms = []
for w in (sys.argv[1] if len(sys.argv) > 1 else "0:0,3:0,4:0").split(","):
    dbg = 0
    if "+" in w:
        w, d = w.split("+"); dbg = int(d)
    c, h = w.split(":")
    arms.append((int(c), int(h), dbg))
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
want = (sys.argv[3] if len(sys.argv) > 3 else "qkv,fc1").split(",")
M = int(os.environ.get("GEMM_M", 277376))
st = torch.cuda.current_stream().cuda_stream
shapes = [s for s in [("qkv", 2304, 768, 0), ("o", 768, 768, 0), ("fc1", 3072, 768, 1), ("fc2", 768, 3072, 0)] if s[0] in want]
data = {}
for name, N, K, mode in shapes:
    g = torch.Generator(device="cuda").manual_seed(N + K)
    A = (torch.randn(M + 256, K, device="cuda", generator=g) * 0.5).half()
    W = (torch.randn(N, K, device="cuda", generator=g) * 0.05).half()
    b = torch.randn(N, device="cuda", generator=g)
    data[name] = (A, W, b, torch.empty(M + 256, N, device="cuda", dtype=torch.float16))


def run(name, N, K, mode, arm, iters):
    A, W, b, out = data[name]
    c, h, dbg = arm
    _lib.dev_set("gemm16_variant", 4 | (dbg << 8))
    (_lib.dev_set("gemm16_walk_c", c), _lib.dev_set("gemm16_walk_h", h))
    for _ in range(iters):
        rc = lib.iisan_gemm16(0, mode, A.data_ptr(), W.data_ptr(), b.data_ptr(), out.data_ptr(), None, M, N, K, st)
        assert rc == 0, lib.iisan_last_error()
    _lib.dev_set("gemm16_variant", 0)
    (_lib.dev_set("gemm16_walk_c", -1), _lib.dev_set("gemm16_walk_h", 0))


for name, N, K, mode in shapes:
    ref = None
    for arm in arms:
        data[name][3].zero_()
        run(name, N, K, mode, arm, 1)
        torch.cuda.synchronize()
        o = data[name][3][:M].clone()
        if ref is None:
            ref = o
        elif not torch.equal(o, ref):
            print(f"{name}: walk {arm} DIFFERENT from {arms[0]}: max|d| {(o.float() - ref.float()).abs().max().item():.3e}", flush=True)
    print(f"{name}: bit-identity of {len(arms)} walks checked", flush=True)
for r in range(rounds):
    for arm in arms:
        row = []
        for name, N, K, mode in shapes:
            run(name, N, K, mode, arm, 2)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run(name, N, K, mode, arm, 10)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 10
            row.append(f"{name} {2.0 * M * N * K / dt / 1e12:6.0f}")
        print(f"round {r} walk {arm[0]}:{arm[1]}+{arm[2]}: " + "  ".join(row), flush=True)
