#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd (.db) kernel trace: per-kernel calls / total / average / share, optionally split by
grid size (so the QKV / O / FC1 / FC2 launches of one GEMM kernel can be told apart).  Writes markdown to stdout."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
by_grid = len(sys.argv) > 2 and sys.argv[2] == "--by-grid"
by_stream = len(sys.argv) > 2 and sys.argv[2] == "--by-stream"      # one table per HIP stream (overlapped towers: main stream = ViT)
cur = db.cursor()
if len(sys.argv) > 2 and sys.argv[2] == "--schema":
    print([r[1] for r in cur.execute("pragma table_info(kernels)").fetchall()])
    sys.exit(0)
if by_stream:
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)").fetchall()]
    scol = next((c for c in ("stream_id", "stream", "queue_id", "queue") if c in cols), None)
    if scol is None:
        print("no stream / queue column in the kernels view:", cols); sys.exit(0)
    allrows = cur.execute(f"select {scol}, name, duration from kernels").fetchall()
    streams = {}
    for sid, name, dur in allrows:
        a = streams.setdefault(sid, {}).setdefault(name, [0, 0]); a[0] += 1; a[1] += dur
    for sid, agg in sorted(streams.items(), key=lambda kv: -sum(a[1] for a in kv[1].values())):
        tot = sum(a[1] for a in agg.values())
        print(f"\n### {scol} {sid}: {sum(a[0] for a in agg.values())} dispatches, {tot/1e6:.2f} ms of kernel time\n")
        print("| kernel | calls | total ms | avg us | share |\n|---|---|---|---|---|")
        for name, (n, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
            nm = name if len(name) < 110 else name[:107] + "..."
            print(f"| `{nm}` | {n} | {d/1e6:.3f} | {d/n/1e3:.1f} | {100*d/tot:.1f}% |")
    sys.exit(0)
rows = cur.execute("select name, grid_x, duration from kernels").fetchall()
agg = {}
for name, gx, dur in rows:
    key = (name, gx) if by_grid else (name, 0)
    a = agg.setdefault(key, [0, 0])
    a[0] += 1
    a[1] += dur
tot = sum(a[1] for a in agg.values())
print(f"| kernel | {'grid_x | ' if by_grid else ''}calls | total ms | avg us | share |")
print(f"|---|{'---|' if by_grid else ''}---|---|---|---|")
for (name, gx), (n, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    nm = name if len(name) < 110 else name[:107] + "..."
    g = f"{gx} | " if by_grid else ""
    print(f"| `{nm}` | {g}{n} | {d/1e6:.3f} | {d/n/1e3:.1f} | {100*d/tot:.1f}% |")
print(f"\ntotal kernel time {tot/1e6:.2f} ms over {len(rows)} dispatches")
