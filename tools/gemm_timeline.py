"""Dump the s_memtime timeline of workgroup 0 (waves 0 and 4) of the staggered GEMM (development aid)."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iisan_amd import _lib
lib = _lib.load()
M, N, K, mode = 277376, int(sys.argv[1]) if len(sys.argv) > 1 else 2304, 768, 0
A = (torch.randn(M + 256, K, device="cuda") * 0.5).half(); W = (torch.randn(N, K, device="cuda") * 0.05).half()
b = torch.randn(N, device="cuda"); out = torch.empty(M + 256, N, device="cuda", dtype=torch.float16)
log = torch.zeros(8192 * 2, dtype=torch.int64, device="cuda")
st = torch.cuda.current_stream().cuda_stream
lib.iisan_set_gemm16_variant(3)
for _ in range(2):
    lib.iisan_gemm16(0, mode, A.data_ptr(), W.data_ptr(), b.data_ptr(), out.data_ptr(), None, M, N, K, st)
lib.iisan_set_gemm16_variant(3 + (16 << 8))
lib.iisan_gemm16(0, mode, A.data_ptr(), W.data_ptr(), b.data_ptr(), out.data_ptr(), log.data_ptr(), M, N, K, st)
torch.cuda.synchronize()
lib.iisan_set_gemm16_variant(0)
L = log.cpu().numpy().reshape(-1, 2)
names = {1: "R-start", 2: "R-end(wait done)", 3: "M-start(after barrier)", 4: "mfma done", 5: "vmcnt done", 6: "epilogue done"}
for g, off in (("A", 0), ("B", 2048)):
    ev = L[off:off + 2000]; ev = ev[ev[:, 1] > 0]
    t0 = ev[0, 0]
    print(f"group {g}: {len(ev)} stamps; first 3 tiles (cycles since start, delta):")
    prev = t0
    for i, (t, tag) in enumerate(ev[50:5 * 12 + 14]):
        print(f"  {i:4d} {names[int(tag)]:24s} {t - t0:9d}  +{t - prev}")
        prev = t
    break
