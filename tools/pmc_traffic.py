#!/usr/bin/env python3
"""HBM-side traffic per launch from two rocprofv3 PMC passes of the same command (FETCH_SIZE and WRITE_SIZE do not fit
one pass on gfx950: MI355X_MICROARCH.md, "rocprofv3 PMC slots"):

    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d A -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline
    rocprofv3 --pmc WRITE_SIZE --kernel-trace -d B -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline
    python tools/pmc_traffic.py A/p_results.db B/p_results.db [kernel-substring]

The launch order is deterministic, so the i-th dispatch of a kernel in one pass is the i-th in the other.  Units: both
counters are KiB; FETCH_SIZE is doubled (gfx950 tallies the 128-byte requests of wide coalesced reads at 64 B — same
guide, HBM section); WRITE_SIZE is reported as counted (uncalibrated: partial-line writes are rounded up).
Launches of one kernel are clustered by their byte counts (the persistent GEMM always has a 256-workgroup grid, so the
QKV / O / FC1 / FC2 products of the two towers can only be told apart by their traffic)."""
import sqlite3, sys, json, collections

def per_dispatch(path, counter):
    db = sqlite3.connect(path)
    rows = db.execute("select kernel_name, dispatch_id, sum(value) from counters_collection where counter_name=? "
                      "group by kernel_name, dispatch_id order by dispatch_id", (counter,)).fetchall()
    out = collections.defaultdict(list)
    for k, _, v in rows:
        out[k].append(v)
    return out

fetch = per_dispatch(sys.argv[1], "FETCH_SIZE")
write = per_dispatch(sys.argv[2], "WRITE_SIZE")
filt = sys.argv[3] if len(sys.argv) > 3 else ""
tot_launch, tot_bytes = 0, 0.0
print("| kernel | launches | read MB (2 x FETCH_SIZE) | written MB (WRITE_SIZE) | total MB per launch |")
print("|---|---|---|---|---|")
for k in sorted(fetch):
    if filt not in k or k not in write or len(write[k]) != len(fetch[k]):
        continue
    cl = collections.OrderedDict()
    for f, w in zip(fetch[k], write[k]):
        key = (round(f / 1024 / 8), round(w / 1024 / 8))         # 8-MiB buckets
        a = cl.setdefault(key, [0, 0.0, 0.0])
        a[0] += 1; a[1] += 2 * f * 1024; a[2] += w * 1024
    for (_, _), (n, fb, wb) in sorted(cl.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
        print(f"| `{k[:90]}` | {n} | {fb / n / 1e6:.1f} | {wb / n / 1e6:.1f} | {(fb + wb) / n / 1e6:.1f} |")
        tot_launch += n; tot_bytes += fb + wb
if tot_launch:
    print(f"\nall listed launches: {tot_launch}, average {tot_bytes / tot_launch / 1e6:.1f} MB per launch")
    print(json.dumps({"launches": tot_launch, "avg_bytes_per_launch": tot_bytes / tot_launch}))
