#!/usr/bin/env python3
"""Stage timeline of workgroup 0 of the fused SASRec forward (cycle stamps, development aid)."""
import os, sys, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import golden_io as gio
from iisan_amd import _lib, ops
lib = _lib.load()
S, E, H, L = 10, 64, 2, 2
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
P = {k: v for k, v in gio.weights.make_trainable_params(seed=99).items() if k.startswith("user_encoder.")}
order = ops.sasrec_param_order(L)
params = [P["user_encoder.transformer_encoder." + k].cuda() for k in order]
x = torch.randn(B, S, E, device="cuda"); lm = torch.ones(B, S, device="cuda")
st = torch.zeros(64, dtype=torch.int64, device="cuda")
cfg = ops.make_sasrec_cfg(S, E, H, L, 0.0, 1)
for it in range(3):
    _lib.dev_set("sasrec_stamps", st.data_ptr() or 0)
    y = ops.SasrecFn.apply(cfg, x, lm, *params)
    torch.cuda.synchronize()
    _lib.dev_set("sasrec_stamps", None or 0)
    t = st.cpu().tolist()
    n = max(i for i, v in enumerate(t) if v) + 1
    print("run", it, "total cycles", t[n - 1] - t[0], "stages:", [t[i + 1] - t[i] for i in range(n - 1)])
