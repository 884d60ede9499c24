#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd (.db) kernel trace: per-kernel calls / total / average / share, optionally split by
grid size (so the QKV / O / FC1 / FC2 launches of one GEMM kernel can be told apart).  Writes markdown to stdout."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
by_grid = len(sys.argv) > 2 and sys.argv[2] == "--by-grid"
cur = db.cursor()
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
