#!/usr/bin/env python3
"""Print per-kernel averages of the PMC counters stored in a rocprofv3 rocpd database."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
filt = sys.argv[2] if len(sys.argv) > 2 else ""
try:
    rows = cur.execute("select * from counters_collection limit 1").fetchall()
    cols = [d[0] for d in cur.description]
except Exception as e:
    print("no counters_collection view:", e); sys.exit(0)
q = "select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name"
try:
    res = cur.execute(q).fetchall()
except Exception:
    print("columns:", cols); sys.exit(0)
for k, c, v, n in res:
    if filt in k:
        print(f"{k[:70]:70s} {c:32s} {v:18.1f} (n={n})")
