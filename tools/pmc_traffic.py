#!/usr/bin/env python3
"""Turn two rocprofv3 PMC passes into profiles/<round>_traffic.json (HBM bytes per frame and kernel group).

Collect (on the GPU box, separate passes, counters only with --kernel-trace, as MI355X_MICROARCH.md prescribes):

  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmc_f -o f --output-format csv -- \
      python3 bench.py --steps 3 --warmup 1 --pairs-per-step 8 --cpu-sample 0 --overlap-streams 0
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/pmc_w -o w --output-format csv -- \
      python3 bench.py --steps 3 --warmup 1 --pairs-per-step 8 --cpu-sample 0 --overlap-streams 0

then:  python tools/pmc_traffic.py gpurun_out/pmc_f/f_counter_collection.csv \
                                   gpurun_out/pmc_w/w_counter_collection.csv 16 profiles/r01_traffic.json

FETCH_SIZE / WRITE_SIZE are KB per dispatch.  gfx950 correction (guide, HBM section): FETCH_SIZE counts half
the bytes of wide (16 B/lane) coalesced reads; WRITE_SIZE is exact for 16-B streaming stores.  The read side
is therefore given raw AND doubled for kernels that stream with 16-B loads; bench.py uses the conservative
(larger) figure so that `traffic` is never understated.
"""
import csv
import json
import re
import sys
from collections import defaultdict

# kernel-name substring -> (bench.py kernel group, reads are wide 16-B/lane streams?)
GROUPS = [
    ("k_dewarp_gray", "dewarp_gray", True),
    ("k_fast_planes", "fast", False), ("k_seg_scan", "fast", False), ("k_fast_compact", "fast", False),
    ("k_nms_", "nms", False),
    ("k_brief", "brief", False),
    ("k_ham_mfma", "ham_argmin", True), ("k_ham_valu", "ham_argmin", False),
    ("k_match_select", "match_select", False), ("k_tail_fill", "tail_fill", False),
    ("k_match_finish", "match_finish", False), ("k_match_init", "match_init", False),
]


def load(path, counter):
    per_kernel = defaultdict(list)
    with open(path, newline="") as fh:
        for row in csv.DictReader(fh):
            if row["Counter_Name"] != counter:
                continue
            m = re.search(r"\bk_[a-z0-9_]+", row["Kernel_Name"])
            if not m:
                continue
            short = m.group(0)
            per_kernel[short].append(float(row["Counter_Value"]))
    return per_kernel


def main():
    fpath, wpath, frames, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    fetch, write = load(fpath, "FETCH_SIZE"), load(wpath, "WRITE_SIZE")
    raw = {}
    for k in sorted(set(fetch) | set(write)):
        if not k.startswith("k_"):
            continue
        raw[k] = {}
        for nm, d in (("FETCH_SIZE", fetch), ("WRITE_SIZE", write)):
            if k in d:
                raw[k][nm] = {"sum_KB": sum(d[k]), "launches": len(d[k])}
    # a "step" launches every kernel group once per batch; normalise per detect launch (= per step) and per frame
    steps = raw["k_dewarp_gray"]["FETCH_SIZE"]["launches"]
    groups = defaultdict(lambda: {"fetch_raw": 0.0, "fetch_wide_x2": 0.0, "write": 0.0})
    for k, v in raw.items():
        for sub, grp, wide in GROUPS:
            if sub in k:
                f = v.get("FETCH_SIZE", {}).get("sum_KB", 0.0) * 1024 / steps
                w = v.get("WRITE_SIZE", {}).get("sum_KB", 0.0) * 1024 / steps
                groups[grp]["fetch_raw"] += f
                groups[grp]["fetch_wide_x2"] += f * (2 if wide else 1)
                groups[grp]["write"] += w
                break
    detect = ("dewarp_gray", "fast", "nms", "brief")
    res = {
        "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, with --kernel-trace) on "
                  "`bench.py --steps 3 --warmup 1 --pairs-per-step %d --cpu-sample 0 --overlap-streams 0` (%d frames per launch), MI355X; "
                  "made by tools/pmc_traffic.py" % (frames // 2, frames),
        "units": "FETCH_SIZE/WRITE_SIZE are KB per dispatch; bytes = value*1024",
        "correction": "MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE reports 1/2 of the bytes of wide (16 B/lane) "
                      "coalesced reads; WRITE_SIZE is exact for 16-B streaming stores; other widths uncalibrated. "
                      "fetch_wide_x2 doubles the read side of kernels that stream with 16-B loads (upper bound).",
        "steps_profiled": steps,
        "frames_per_step": frames,
        "raw_per_kernel": raw,
        "bytes_per_step": {g: v for g, v in groups.items()},
        "bytes_per_frame": {g: (groups[g]["fetch_wide_x2"] + groups[g]["write"]) / frames for g in detect if g in groups},
        "bytes_per_pair": {g: (v["fetch_wide_x2"] + v["write"]) / (frames // 2) for g, v in groups.items() if g not in detect},
    }
    with open(out, "w") as fh:
        json.dump(res, fh, indent=1)
    for g, v in res["bytes_per_frame"].items():
        print("%-14s %8.2f MB/frame" % (g, v / 1e6))
    for g, v in res["bytes_per_pair"].items():
        print("%-14s %8.2f MB/pair" % (g, v / 1e6))


if __name__ == "__main__":
    main()
