import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iisan_amd import _lib
lib = _lib.load()
items, S, heads = 1408, 197, 12
qkv = (torch.randn(items, heads, 3, S, 64, device="cuda")).half()
ctx = torch.empty(items * S, heads * 64, device="cuda", dtype=torch.float16)
st = torch.cuda.current_stream().cuda_stream
_lib.dev_set("attn_debug", int(sys.argv[1]) if len(sys.argv) > 1 else 0)
for _ in range(3):
    lib.iisan_attention16(0, qkv.data_ptr(), None, ctx.data_ptr(), items, S, heads, st)
torch.cuda.synchronize()
