#!/usr/bin/env python3
"""Device time of the SASRec forward / backward C-ABI calls alone (HIP events around the ctypes calls, buffers preallocated)."""
import os, sys, ctypes as C, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import golden_io as gio
from iisan_amd import _lib, ops
lib = _lib.load()
S, E, H, L = 10, 64, 2, 2
P = {k: v for k, v in gio.weights.make_trainable_params(seed=99).items() if k.startswith("user_encoder.")}
order = ops.sasrec_param_order(L)
st = torch.cuda.current_stream().cuda_stream
for B in (128, 1024):
    for p in (0.0, 0.1):
        for fused in (1, 0):
            _lib.dev_set("sasrec_fused", fused)
            params = [P["user_encoder.transformer_encoder." + k].cuda().contiguous() for k in order]
            grads = [torch.zeros_like(t) for t in params]
            x = torch.randn(B, S, E, device="cuda"); lm = torch.ones(B, S, device="cuda"); y = torch.empty_like(x); dy = torch.randn_like(x); dx = torch.empty_like(x)
            cfg = ops.make_sasrec_cfg(S, E, H, L, p, 12345)
            ws = torch.empty(lib.iisan_sasrec_ws_bytes(C.byref(cfg), B), dtype=torch.uint8, device="cuda")
            pt, gt = ops._ptr_table(params), ops._ptr_table(grads)
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            tf = tb = 0.0
            for it in range(6):
                ev[0].record()
                assert lib.iisan_sasrec_fwd(C.byref(cfg), x.data_ptr(), lm.data_ptr(), B, pt, y.data_ptr(), ws.data_ptr(), ws.numel(), st) == 0
                ev[1].record()
                assert lib.iisan_sasrec_bwd(C.byref(cfg), x.data_ptr(), lm.data_ptr(), B, pt, dy.data_ptr(), dx.data_ptr(), gt, ws.data_ptr(), ws.numel(), st) == 0
                ev[2].record()
                torch.cuda.synchronize()
                if it:
                    tf += ev[0].elapsed_time(ev[1]) / 5; tb += ev[1].elapsed_time(ev[2]) / 5
            print(f"B={B} p={p} fused={fused}: fwd {tf*1e3:.0f} us  bwd {tb*1e3:.0f} us  (ws {ws.numel()/1e6:.1f} MB)", flush=True)
_lib.dev_set("sasrec_fused", 1)
