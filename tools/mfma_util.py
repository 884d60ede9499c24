#!/usr/bin/env python3
"""MFMA utilisation per kernel from a rocprofv3 PMC pass holding SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES, SQ_WAVE_CYCLES, SQ_WAIT_ANY,
SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_ANY, SQ_INSTS_VALU_MFMA_MOPS / SQ_INSTS_MFMA (whatever of them the pass collected) plus the kernel trace of
the same run (durations):
    python tools/mfma_util.py <rocpd db> [kernel-substring ...]
Per kernel (launches clustered by their MFMA-busy count, so the QKV / O / FC1 / FC2 products of one GEMM kernel separate):
    matrix-pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (SIMDs x kernel cycles), kernel cycles from SQ_BUSY_CYCLES (per shader engine, max) or,
    when absent, from the trace duration x an assumed clock; wave-time fractions = SQ_WAIT_ANY / SQ_WAIT_INST_ANY / SQ_ACTIVE_INST_ANY over
    SQ_WAVE_CYCLES.  MI355X: 256 CUs x 4 SIMDs = 1024 SIMDs, 32 shader engines."""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1])
filts = sys.argv[2:] or [""]
rows = db.execute("select kernel_name, dispatch_id, counter_name, sum(value) from counters_collection group by kernel_name, dispatch_id, counter_name").fetchall()
dur = {d: (e - s) for d, s, e in db.execute("select dispatch_id, start, end from kernels").fetchall()}
per = collections.defaultdict(dict)
for k, d, c, v in rows:
    per[(k, d)][c] = v
agg = collections.OrderedDict()
for (k, d), c in per.items():
    if not any(f in k for f in filts):
        continue
    mf = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    key = (k, round(mf / 2e7))             # cluster launches of one kernel by their matrix work
    a = agg.setdefault(key, collections.defaultdict(float))
    a["n"] += 1
    a["ns"] += dur.get(d, 0)
    for cn, v in c.items():
        a[cn] += v
print("| kernel | launches | avg us (trace, profiled run) | MFMA busy cycles / launch (M) | SQ busy cycles / launch (M, summed over 32 SEs) | matrix pipe busy | waves waiting | issue-stalled | issuing |")
print("|---|---|---|---|---|---|---|---|---|")
for (k, _), a in sorted(agg.items(), key=lambda kv: -kv[1]["ns"]):
    n = a["n"]
    mf, sqb, wc = a["SQ_VALU_MFMA_BUSY_CYCLES"] / n, a["SQ_BUSY_CYCLES"] / n, a["SQ_WAVE_CYCLES"] / n
    # SQ_BUSY_CYCLES is reported per shader engine and summed: kernel cycles = sum / 32; 1024 SIMDs share the MFMA-busy sum
    kc = sqb / 32 if sqb else 0
    util = mf / (1024 * kc) if kc else float("nan")
    f = lambda x: f"{100 * a[x] / n / wc:.1f} %" if wc else "-"
    nm = k.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    print(f"| `{nm[:70]}` | {int(n)} | {a['ns'] / n / 1e3:.1f} | {mf / 1e6:.2f} | {sqb / 1e6:.2f} | {100 * util:.1f} % | {f('SQ_WAIT_ANY')} | {f('SQ_WAIT_INST_ANY')} | {f('SQ_ACTIVE_INST_ANY')} |")
