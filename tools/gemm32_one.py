import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iisan_amd import _lib
lib = _lib.load()
M, N, K = 11264, 64, 768
A = torch.randn(M, K, device="cuda"); B = torch.randn(N, K, device="cuda"); C_ = torch.zeros(M, N, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for _ in range(5):
    lib.iisan_gemm32(A.data_ptr(), B.data_ptr(), None, C_.data_ptr(), M, N, K, 0, 0, 0, 0, st)
torch.cuda.synchronize()
