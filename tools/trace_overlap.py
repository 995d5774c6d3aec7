#!/usr/bin/env python3
"""Which kernels run beside which: reads a rocprofv3 --kernel-trace CSV and prints, for the middle part of the run, the time
covered by every combination of kernel groups (developer tool).

  python tools/trace_overlap.py <kernel_trace.csv> [lo_frac hi_frac]
"""
import csv
import re
import sys
from collections import defaultdict

GROUPS = (("k_ham", "ham"), ("k_tail_rows", "rows"), ("k_match_gs", "finish"), ("k_match_finish", "finish"), ("k_match_select", "select"),
          ("k_match_init", "init"), ("k_trk", "tracks"), ("k_dewarp", "detect"), ("k_fast", "detect"), ("k_nms", "detect"), ("k_brief", "detect"))


def group(name):
    for pat, g in GROUPS:
        if pat in name:
            return g
    return "other"


def main():
    path = sys.argv[1]
    lo, hi = (float(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (0.3, 0.7)
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    # the window is placed by launch index of the distance kernel (the run has long idle stretches around the timed steps)
    ham = [r for r in rows if "k_ham" in r[2]] or rows
    a, b = ham[int(len(ham) * lo)][0], ham[min(len(ham) - 1, int(len(ham) * hi))][0]
    ev = []
    per = defaultdict(lambda: [0, 0.0])
    for s, e, n in rows:
        if e <= a or s >= b:
            continue
        g = group(n)
        s2, e2 = max(s, a), min(e, b)
        ev.append((s2, 1, g))
        ev.append((e2, -1, g))
        per[g][0] += 1
        per[g][1] += e2 - s2
    byname = defaultdict(lambda: [0, 0.0])
    for s, e, n in rows:
        if s >= a and e <= b:
            mm = re.search(r"k_\w+(<[^>]*>)?", n)
            nm = mm.group(0) if mm else n[:60]
            byname[nm][0] += 1
            byname[nm][1] += e - s
    ev.sort()
    active = defaultdict(int)
    cover = defaultdict(float)
    last = a
    for t, d, g in ev:
        key = "+".join(sorted(k for k, v in active.items() if v > 0)) or "(idle)"
        cover[key] += t - last
        last = t
        active[g] += d
    cover["(idle)"] += b - last
    span = b - a
    print("window %.3f ms" % (span / 1e6))
    for k, v in sorted(cover.items(), key=lambda kv: -kv[1]):
        print("  %-28s %8.3f ms  %5.1f %%" % (k, v / 1e6, 100 * v / span))
    print("per kernel (whole launches inside the window): launches, mean us, summed ms")
    for n, (k, d) in sorted(byname.items(), key=lambda kv: -kv[1][1]):
        print("  %-60s %6d %9.1f %9.3f" % (n, k, d / k / 1e3, d / 1e6))
    print("per group: launches, summed duration (ms), share of window")
    for g, (n, d) in sorted(per.items(), key=lambda kv: -kv[1][1]):
        print("  %-8s %6d %9.3f  %5.1f %%" % (g, n, d / 1e6, 100 * d / span))


if __name__ == "__main__":
    main()
