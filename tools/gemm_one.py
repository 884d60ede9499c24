import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iisan_amd import _lib
lib = _lib.load()
var = int(sys.argv[1]) if len(sys.argv) > 1 else 3
M, N, K, mode = 277376, 2304, 768, 0
A = (torch.randn(M + 256, K, device="cuda") * 0.5).half()
W = (torch.randn(N, K, device="cuda") * 0.05).half()
b = torch.randn(N, device="cuda")
out = torch.empty(M + 256, N, device="cuda", dtype=torch.float16)
st = torch.cuda.current_stream().cuda_stream
_lib.dev_set("gemm16_variant", var)
for _ in range(3):
    lib.iisan_gemm16(0, mode, A.data_ptr(), W.data_ptr(), b.data_ptr(), out.data_ptr(), None, M, N, K, st)
torch.cuda.synchronize()
