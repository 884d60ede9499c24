#!/usr/bin/env python3
"""A/B of gemm16 builds on the SAME box: python tools/gemm_ab.py libA.so libB.so ...  (each timed in a child process,
round-robin, three rounds; development aid — boxes differ by up to 10 %, so only same-box ratios mean anything)."""
import sys, os, subprocess
CHILD = r'''
import sys, os, time, torch
sys.path.insert(0, os.getcwd())
from iisan_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
import ctypes
_probe = ctypes.CDLL(_lib.LIB_PATH)          # older builds lack newer tooling symbols: bind what exists
for _t in (_lib.SIGNATURES,):
    for _n in [n for n in _t if not hasattr(_probe, n)]: del _t[_n]
lib = _lib.load()
M = 277376
st = torch.cuda.current_stream().cuda_stream
out_s = []
for name, N, K, mode in [("qkv", 2304, 768, 0), ("o", 768, 768, 0), ("fc1", 3072, 768, 1), ("fc2", 768, 3072, 0)]:
    A = (torch.randn(M + 256, K, device="cuda") * 0.5).half(); W = (torch.randn(N, K, device="cuda") * 0.05).half()
    b = torch.randn(N, device="cuda"); bp = None if os.environ.get("NOBIAS") else b.data_ptr(); out = torch.empty(M + 256, N, device="cuda", dtype=torch.float16)
    for _ in range(5): lib.iisan_gemm16(0, mode, A.data_ptr(), W.data_ptr(), bp, out.data_ptr(), None, M, N, K, st)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): lib.iisan_gemm16(0, mode, A.data_ptr(), W.data_ptr(), bp, out.data_ptr(), None, M, N, K, st)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    out_s.append(f"{name} {2.0 * M * N * K / dt / 1e12:6.0f}")
print("  ".join(out_s))
'''
for rnd in range(int(os.environ.get("ROUNDS", "3"))):
    for lib in sys.argv[1:]:
        r = subprocess.run([sys.executable, "-c", CHILD, lib], capture_output=True, text=True)
        print(f"{os.path.basename(lib):24s} {r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:]}", flush=True)
