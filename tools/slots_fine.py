#!/usr/bin/env python3
"""Finer slot timeline of gemm16_h256 (development aid of round 3): like tools/gemm_slots_h.py, with extra stamps inside the
last step's [Rhi + E] slot (after the arithmetic, after the stores, after lgkmcnt, after vmcnt).  Needs tools/lib_timeline.so built with
-DS256_TIMELINE -DS256_FINE and the `stamp2()` calls of that experiment in the kernel (they are not in the tree any more; kept for
the record of how DESIGN 6e's per-stage cycle counts were taken).  python tools/slots_fine.py [debug bits ...]"""
import sys, os
import torch
sys.path.insert(0, "/root/repo")
from iisan_amd import _lib
_lib.LIB_PATH = "/root/repo/tools/lib_timeline.so"
lib = _lib.load()
M=277376; N,K,mode=2304,768,0
A=(torch.randn(M+256,K,device="cuda")*0.5).half(); W=(torch.randn(N,K,device="cuda")*0.05).half(); b=torch.randn(N,device="cuda")
out=torch.empty(M+256,N,device="cuda",dtype=torch.float16); st=torch.cuda.current_stream().cuda_stream
F=["Rlo","bar","Mlo","bar","Rhi+E","bar","Mhi"]
L=["Rlo","bar","Mlo","bar","arith","rdhi+store","lgkm","vmcnt","bar","Mhi"]
labels=[f"F.{x}" for x in F]+[f"m{k}.{x}" for k in range(1,11) for x in ("R","bar","M")]+[f"L.{x}" for x in L]
n=len(labels)
for dbg in [int(a) for a in sys.argv[1:]] or [0]:
    lib.iisan_set_gemm16_variant(4+((dbg|16)<<8))
    for _ in range(2):
        lib.iisan_gemm16(0,mode,A.data_ptr(),W.data_ptr(),b.data_ptr(),out.data_ptr(),None,M,N,K,st); torch.cuda.synchronize()
    t=out.view(-1).view(torch.int32)[8192:8192+2048].cpu().view(2,1024).long()
    for g in range(2):
        base=2*n; x=t[g,base-1:base+n].tolist(); d=[x[i+1]-x[i] for i in range(n)]
        print(f"dbg {dbg} group {'AB'[g]} total {x[-1]-x[0]}: "+" ".join(f"{l}={v}" for l,v in zip(labels,d) if l[0]!='m'))
