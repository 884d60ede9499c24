#!/usr/bin/env python3
"""ISA screen of a hipcc -S dump: per kernel, how many SGPR-spill lane ops / scratch ops / s_waitcnt vmcnt(0) sit close to MFMAs.
usage: python tools/asm_check.py file.s [name-filter]"""
import re, sys, bisect
txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
parts = re.split(r'\n(_Z\w+):[^\n]*\n', txt)
for i in range(1, len(parts), 2):
    name, body = parts[i], parts[i + 1].split('.Lfunc_end')[0]
    if flt not in name:
        continue
    lines = [l for l in body.split('\n') if l.strip() and not l.strip().startswith(';')]
    im = [j for j, l in enumerate(lines) if 'v_mfma' in l]
    def near(pred, dist=6):
        idx = [j for j, l in enumerate(lines) if pred(l)]
        n = 0
        for j in idx:
            k = bisect.bisect(im, j)
            d = min([abs(im[x] - j) for x in (k - 1, k) if 0 <= x < len(im)] or [10**9])
            n += d < dist
        return len(idx), n
    lane = near(lambda l: 'v_writelane' in l or 'v_readlane' in l)
    scr = near(lambda l: 'scratch_' in l)
    print(f"{name[-44:]}: lines {len(lines)} mfma {len(im)} lane-ops {lane[0]} (near mfma {lane[1]}) scratch {scr[0]} (near {scr[1]})")
