#!/usr/bin/env python3
"""Reduce the rocprofv3 output of tools/pmc_collect.sh into small JSON summaries under profiles/.

  python tools/pmc_summary.py gpurun_out/r02_pmc profiles/r02

writes  <prefix>_kernel_stats.csv   (the rocprofv3 --stats kernel table of the 30-step bench run)
        <prefix>_pmc.json           (per kernel: dispatch count and per-counter sum / mean over dispatches)
        <prefix>_traffic.json       (HBM bytes per frame / per image pair for bench.py's `traffic` fields)
FETCH_SIZE / WRITE_SIZE are KB per dispatch.  gfx950 (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts HALF the bytes of wide
(16 B/lane) coalesced streaming reads -- only k_dewarp_gray's source/map reads and k_fast_planes' row loads stream that
way here; both figures (raw and doubled) are kept and bench.py reports the raw one with the doubled one beside it.
"""
import csv
import glob
import json
import os
import re
import shutil
import sys
from collections import defaultdict

GROUPS = [("k_dewarp_gray", "dewarp_gray"), ("k_fast_planes", "fast"), ("k_seg_scan", "fast"), ("k_fast_compact", "fast"),
          ("k_nms", "nms"), ("k_brief", "brief"), ("k_ham_mfma", "ham_argmin"), ("k_ham_fp4", "ham_argmin"), ("k_ham_valu", "ham_argmin"),
          ("k_match_select", "match_select"), ("k_tail_rows", "tail_rows"), ("k_match_gs", "match_finish"),
          ("k_match_finish", "match_finish"), ("k_match_init", "match_init"), ("k_match_order", "match_finish"), ("k_trk_", "tracks")]
DETECT = ("dewarp_gray", "fast", "nms", "brief")


def short(name):
    m = re.search(r"\bk_[a-z0-9_]+", name)
    return m.group(0) if m else None


def load_counters(root):
    per = defaultdict(lambda: defaultdict(list))   # kernel -> counter -> [values per dispatch]
    for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as fh:
            for row in csv.DictReader(fh):
                k = short(row["Kernel_Name"])
                if k:
                    per[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return per


def main():
    root, prefix = sys.argv[1], sys.argv[2]
    frames, pairs = 64, 2016   # bench.py's headline job at N = 1
    stats = glob.glob(os.path.join(root, "stats", "**", "*kernel_stats.csv"), recursive=True)
    if stats:
        shutil.copy(stats[0], prefix + "_kernel_stats.csv")
    per = load_counters(root)
    pmc = {}
    for k, ctrs in sorted(per.items()):
        pmc[k] = {c: {"dispatches": len(v), "sum": sum(v), "mean": sum(v) / len(v)} for c, v in sorted(ctrs.items())}
    # derived ratios the design discussion uses
    derived = {}
    for k, c in pmc.items():
        d = {}
        g = lambda n: c[n]["sum"] if n in c else None
        if g("TCC_HIT_sum") is not None and g("TCC_MISS_sum") is not None and g("TCC_HIT_sum") + g("TCC_MISS_sum") > 0:
            d["l2_hit_rate"] = g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum"))
        if g("TCP_TOTAL_ACCESSES_sum") and g("TCP_TCC_READ_REQ_sum") is not None:
            d["tcp_to_tcc_read_requests_per_l1_access"] = g("TCP_TCC_READ_REQ_sum") / g("TCP_TOTAL_ACCESSES_sum")
        if g("SQ_WAVE_CYCLES") and g("SQ_WAIT_ANY") is not None:
            d["waves_parked_fraction"] = g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES")
        if g("TA_TA_BUSY_sum") and g("TA_ADDR_STALLED_BY_TC_CYCLES_sum") is not None:
            d["ta_busy_cycles_stalled_by_l1"] = g("TA_ADDR_STALLED_BY_TC_CYCLES_sum") / g("TA_TA_BUSY_sum")
        if g("TCP_TCC_READ_REQ_sum") and g("SQ_INSTS_VMEM_RD"):
            d["l2_read_requests_per_vmem_read_instruction"] = g("TCP_TCC_READ_REQ_sum") / g("SQ_INSTS_VMEM_RD")
        if g("SQ_WAVE_CYCLES") and g("SQ_WAIT_INST_ANY") is not None:
            d["issue_stall_fraction"] = g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES")
        if g("SQ_VALU_MFMA_BUSY_CYCLES") and g("GRBM_GUI_ACTIVE"):   # means: GRBM_GUI_ACTIVE is collected in more than one pass
            d["mfma_pipe_busy_fraction"] = c["SQ_VALU_MFMA_BUSY_CYCLES"]["mean"] / (c["GRBM_GUI_ACTIVE"]["mean"] / 8.0 * 1024.0)
            if g("SQ_VALU_MFMA_COEXEC_CYCLES") is not None:
                d["mfma_valu_coexec_share_of_mfma_busy"] = g("SQ_VALU_MFMA_COEXEC_CYCLES") / g("SQ_VALU_MFMA_BUSY_CYCLES")
            d["mfma_pipe_busy_note"] = "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE per XCD x 1024 SIMDs): busy share at the clock the chip held"
        if d:
            derived[k] = d
    json.dump({"source": "rocprofv3 --pmc passes (tools/pmc_collect.sh: `bench.py --no-extra-configs --no-cpu-baseline --steps 3 "
                         "--warmup 1`, one pass per counter set, each with --kernel-trace only), MI355X",
               "per_kernel": pmc, "derived": derived}, open(prefix + "_pmc.json", "w"), indent=1)
    # traffic
    steps = len(per.get("k_dewarp_gray", {}).get("FETCH_SIZE", [])) or 1
    groups = defaultdict(lambda: {"fetch_raw": 0.0, "write": 0.0})
    for k, c in per.items():
        for sub, grp in GROUPS:
            if sub in k:
                groups[grp]["fetch_raw"] += sum(c.get("FETCH_SIZE", [])) * 1024 / steps
                groups[grp]["write"] += sum(c.get("WRITE_SIZE", [])) * 1024 / steps
                break
    res = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over %d profiled steps of bench.py's headline job "
                     "(%d frames, %d image pairs per step), MI355X; made by tools/pmc_summary.py" % (steps, frames, pairs),
           "units": "bytes per step / per frame / per image pair; FETCH_SIZE raw (gfx950 counts half the bytes of 16-B/lane streaming reads: "
                    "for dewarp_gray the true read traffic lies between fetch_raw and 2 x fetch_raw)",
           "bytes_per_step": {g: v for g, v in groups.items()},
           "bytes_per_frame": {g: (groups[g]["fetch_raw"] + groups[g]["write"]) / frames for g in DETECT if g in groups},
           "bytes_per_frame_reads_doubled": {g: (2 * groups[g]["fetch_raw"] + groups[g]["write"]) / frames for g in DETECT if g in groups},
           "bytes_per_pair": {g: (v["fetch_raw"] + v["write"]) / pairs for g, v in groups.items() if g not in DETECT}}
    json.dump(res, open(prefix + "_traffic.json", "w"), indent=1)
    for g, v in res["bytes_per_frame"].items():
        print("%-14s %8.2f MB/frame" % (g, v / 1e6))
    for g, v in res["bytes_per_pair"].items():
        print("%-14s %8.3f MB/pair" % (g, v / 1e6))
    for k, d in derived.items():
        print(k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in d.items() if not a.endswith("note")})


if __name__ == "__main__":
    main()
