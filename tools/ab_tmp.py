import sys, os, time, torch
sys.path.insert(0, os.getcwd())
from iisan_amd import _lib
lib = _lib.load()
M = 277376
st = torch.cuda.current_stream().cuda_stream
for rep in range(2):
  for N in (768, 1536, 2304, 3072, 4096):
    K = 768
    A = (torch.randn(M + 256, K, device="cuda") * 0.5).half(); W = (torch.randn(N, K, device="cuda") * 0.05).half()
    b = torch.randn(N, device="cuda"); out = torch.empty(M + 256, N, device="cuda", dtype=torch.float16)
    for _ in range(5): lib.iisan_gemm16(0, 0, A.data_ptr(), W.data_ptr(), b.data_ptr(), out.data_ptr(), None, M, N, K, st)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): lib.iisan_gemm16(0, 0, A.data_ptr(), W.data_ptr(), b.data_ptr(), out.data_ptr(), None, M, N, K, st)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    tiles = 1084 * (N // 256)
    print(f"N={N} W={N*K*2/1e6:.1f}MB tiles/WG={tiles/256:.2f} eff={tiles/256/-(-tiles//256):.3f}  {2.0 * M * N * K / dt / 1e12:6.0f} TF")
    del A, W, out
