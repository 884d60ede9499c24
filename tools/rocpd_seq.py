#!/usr/bin/env python3
"""Launch sequence of the LAST bench step in a rocprofv3 rocpd (.db) kernel trace: one line per dispatch (start offset, duration,
grid, short kernel name), from the last adam_kernel-but-one to the last adam_kernel.  Development aid: who launches what, where."""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.cursor().execute("select name, start, end, grid_x, grid_y, grid_z from kernels order by start").fetchall()
ad = [i for i, r in enumerate(rows) if "adam_kernel" in r[0]]
lo, hi = (ad[-2] + 1, ad[-1] + 1) if len(ad) >= 2 else (0, len(rows))
t0 = rows[lo][1]
for name, s, e, gx, gy, gz in rows[lo:hi]:
    short = re.sub(r"\(anonymous namespace\)::|void |at::native::", "", name)
    short = re.sub(r"\(.*", "", short)[:60]
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f} us  grid {gx}x{gy}x{gz}  {short}")
