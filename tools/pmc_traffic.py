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

def step_mode(fetch_db, write_db, steps, key=None, out=None,
              skip=("distribution_elementwise", "launch_clamp", "float16tofloat32_copy", "float16_copy_kernel")):
    """`python tools/pmc_traffic.py --step <fetch db> <write db> <steps> [json key] [json file]`: memory-side bytes of one
    WHOLE step (every kernel launched per step, our library's and torch's), per kernel and in total — for the Cached / Versa
    configurations, whose time is spread over ~100 launches of a dozen kernels.  Kernels launched fewer times than there are
    steps (set-up: random fills of the synthetic tap stores), torch's dtype-conversion copies of the stores and device copies
    of more than 50 MB (set-up too: a step's own copies are a few MB) are left out."""
    f, w = per_dispatch(fetch_db, "FETCH_SIZE"), per_dispatch(write_db, "WRITE_SIZE")
    for k in list(f):
        if "copyBuffer" in k and k in w and len(w[k]) == len(f[k]):
            keep = [i for i in range(len(f[k])) if (2 * f[k][i] + w[k][i]) * 1024 < 50e6]
            f[k], w[k] = [f[k][i] for i in keep], [w[k][i] for i in keep]
    rows, tot_r, tot_w = [], 0.0, 0.0
    for k in sorted(f):
        if k not in w or len(w[k]) != len(f[k]) or len(f[k]) < steps or any(x in k for x in skip):
            continue
        rb, wb = 2 * sum(f[k]) * 1024 / steps, sum(w[k]) * 1024 / steps
        rows.append((rb + wb, k, len(f[k]) / steps, rb, wb))
        tot_r += rb; tot_w += wb
    rows.sort(reverse=True)
    print("| kernel | launches per step | read MB per step (2 x FETCH_SIZE) | written MB per step (WRITE_SIZE) | total MB per step |")
    print("|---|---|---|---|---|")
    for t, k, n, rb, wb in rows:
        if t / 1e6 >= 0.05:
            print(f"| `{k[:100]}` | {n:.1f} | {rb / 1e6:.1f} | {wb / 1e6:.1f} | {t / 1e6:.1f} |")
    print(f"\nwhole step: {tot_r / 1e6:.1f} MB read + {tot_w / 1e6:.1f} MB written = {(tot_r + tot_w) / 1e6:.1f} MB")
    if key and out:
        import os
        d = json.load(open(out)) if os.path.exists(out) else {}
        d[key] = {"bytes_per_step": tot_r + tot_w, "read_bytes_per_step": tot_r, "written_bytes_per_step": tot_w, "steps_in_trace": steps,
                  "unit": "bytes per step = 2 x FETCH_SIZE KiB + WRITE_SIZE KiB (gfx950 correction of MI355X_MICROARCH.md), all kernels of a step"}
        json.dump(d, open(out, "w"), indent=1)


def default_mode():
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



if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--step":
        step_mode(sys.argv[2], sys.argv[3], int(sys.argv[4]), *(sys.argv[5:7]))
    else:
        default_mode()
