#!/usr/bin/env python3
"""The residual add in the epilogue of gemm16_h256 (EPI_STREAM16: x += A W^T + b in the fp16 stream, per-slice row sums, CLS rows
receive the delta) + stream_stats_finalize, on the two ViT-B shapes (O: K = 768, FC2: K = 3072; M = 277,376 rows, N = 768):
every element against fp32 arithmetic, rstd against torch, and interleaved timing against the plain 16-bit epilogue.
    python tools/gemm_stream.py [rounds] [M]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iisan_amd import _lib

lib = _lib.load()
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
S, N = 197, 768
items = (int(sys.argv[2]) if len(sys.argv) > 2 else 277376) // S
M = items * S
Mp = (M + 255) // 256 * 256
st = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device="cuda").manual_seed(9)
for name, K in (("o", 768), ("fc2", 3072)):
    A = (torch.randn(Mp, K, device="cuda", generator=g) * 0.5).half()
    W = (torch.randn(N, K, device="cuda", generator=g) * (0.05 if K == 768 else 0.025)).half()
    b = torch.randn(N, device="cuda", generator=g) * 0.3
    x0 = (torch.randn(Mp, N, device="cuda", generator=g) * 1.5 + 0.2).half()
    x0[:, 7] += 20.0
    xc0 = torch.randn(items, N, device="cuda", generator=g) * 1.5
    part = torch.zeros(N // 64, Mp, 2, device="cuda")
    rstat = torch.zeros(Mp, device="cuda")
    _lib.dev_set("gemm16_variant", 4 | (int(os.environ.get('DBG', 0)) << 8))
    for rep in range(3):
        x = x0.clone(); xc = xc0.clone(); part.zero_(); rstat.zero_()
        assert lib.iisan_gemm16_stream(A.data_ptr(), W.data_ptr(), b.data_ptr(), x.data_ptr(), part.data_ptr(), M, N, K, S, st) == 0, lib.iisan_last_error()
        xg = x.clone()
        assert lib.iisan_stream_stats_finalize(part.data_ptr(), N // 64, Mp, x.data_ptr(), xc.data_ptr(), rstat.data_ptr(), 1e-6, items, S, st) == 0
        torch.cuda.synchronize()
        worst = worst_c = worst_r = 0.0
        nbad = 0
        for r0 in range(0, M, 197 * 128):
            r1 = min(M, r0 + 197 * 128)
            d = A[r0:r1].float() @ W.float().t() + b
            want = x0[r0:r1].float() + d
            cls = torch.arange(r0, r1, device="cuda") % S == 0
            e = (xg[r0:r1].float() - torch.where(cls[:, None], d, want)).abs()
            nbad += int((e > 0.05).sum()); worst = max(worst, e.max().item())
            # after the finalize step: CLS rows = fp16(xc0 + fp16 delta), fp32 stream updated; rstd of the rounded rows
            i0, i1 = r0 // S, (r1 + S - 1) // S
            xc_want = xc0[i0:i1] + d[cls].half().float()
            worst_c = max(worst_c, (xc[i0:i1] - xc_want).abs().max().item(), (x[r0:r1][cls].float() - xc_want.half().float()).abs().max().item())
            xr = x[r0:r1].float()
            rs = torch.rsqrt(xr.var(1, unbiased=False) + 1e-6)
            worst_r = max(worst_r, ((rstat[r0:r1] - rs).abs() / rs).max().item())
        print(f"{name}: check {rep}: {nbad} elements off by more than 0.05 (worst {worst:.3e}); CLS rows worst {worst_c:.3e}; rstd worst rel {worst_r:.3e}", flush=True)
    out = torch.empty(Mp, N, device="cuda", dtype=torch.float16)
    x = x0.clone(); xc = xc0.clone()

    def run_plain(it):
        for _ in range(it):
            assert lib.iisan_gemm16(0, 0, A.data_ptr(), W.data_ptr(), b.data_ptr(), out.data_ptr(), None, M, N, K, st) == 0

    def run_stream(it):
        for _ in range(it):
            assert lib.iisan_gemm16_stream(A.data_ptr(), W.data_ptr(), b.data_ptr(), x.data_ptr(), part.data_ptr(), M, N, K, S, st) == 0

    def run_fin(it):
        for _ in range(it):
            lib.iisan_stream_stats_finalize(part.data_ptr(), N // 64, Mp, x.data_ptr(), xc.data_ptr(), rstat.data_ptr(), 1e-6, items, S, st)

    for r in range(rounds):
        row = []
        for nm, fn in (("plain", run_plain), ("stream", run_stream), ("finalize", run_fin)):
            fn(2); torch.cuda.synchronize(); t0 = time.perf_counter(); fn(10); torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 10
            x.copy_(x0)
            row.append(f"{nm} {dt * 1e6:7.1f} us" + (f" ({2.0 * M * N * K / dt / 1e12:5.0f} TF)" if nm != "finalize" else ""))
        print(f"{name} round {r}: " + "   ".join(row), flush=True)
    _lib.dev_set("gemm16_variant", 0)
