import sys, os, time, subprocess
CH = r'''
import sys, os, time, torch
sys.path.insert(0, os.getcwd())
from iisan_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
lib = _lib.load()
items, S, heads = 1408, 197, 12
qkv = (torch.randn(items, heads, 3, S, 64, device="cuda")).half()
ctx = torch.empty(items * S, heads * 64, device="cuda", dtype=torch.float16)
st = torch.cuda.current_stream().cuda_stream
for _ in range(3): lib.iisan_attention16(0, qkv.data_ptr(), None, ctx.data_ptr(), items, S, heads, st)
torch.cuda.synchronize(); t0=time.perf_counter()
for _ in range(20): lib.iisan_attention16(0, qkv.data_ptr(), None, ctx.data_ptr(), items, S, heads, st)
torch.cuda.synchronize(); print(f"{(time.perf_counter()-t0)/20*1e6:.1f} us")
'''
for r in range(3):
    for lib in sys.argv[1:]:
        o = subprocess.run([sys.executable, "-c", CH, lib], capture_output=True, text=True)
        print(os.path.basename(lib), o.stdout.strip() or o.stderr[-300:], flush=True)
