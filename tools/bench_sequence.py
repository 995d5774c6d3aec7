#!/usr/bin/env python3
"""BASELINE configs[2]: a 64-frame 1920x1080 sequence (frame_i = frame_0 translated by
(3i, i) px, the stand-in for the absent Blender render), detect every frame, match ALL ordered pairs i < j
(2016 image pairs x up to 4096^2) on one MI355X.  A few pairs are re-checked against the CPU oracle."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import photogrammetry_amd as pg
from photogrammetry_amd import dist as pdist
from photogrammetry_amd import synth

W, H, P, WORDS, NKP, RADIUS, THRESH, CAP = 1920, 1080, 256, 8, 4096, 16, 0.1, 8192


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--check", type=int, default=3)
    args = ap.parse_args()
    F = args.frames
    dev = torch.device("cuda", 0)
    base = synth.make_frame(W, H, seed=4321, n_shapes=20000)
    frames = np.stack([synth.shift_frame(base, 3 * i, i) for i in range(F)])
    eng = pg.Engine(0)
    pairs_tbl = pg.make_brief_pairs(0, 50, P)
    dmap = pg.build_dewarp_map(W, H, [3e-4, 1e-7, 0, 0, 0])
    eng.set_brief_pairs(pairs_tbl)
    eng.set_detect_params(THRESH, RADIUS)
    eng.set_capacity(1 << 18, CAP)
    eng.set_dewarp_map(dmap)
    d_frames = torch.from_numpy(frames).to(dev)
    d_kp = torch.zeros((F, CAP, 4), dtype=torch.int32, device=dev)
    d_desc = torch.zeros((F, CAP, WORDS), dtype=torch.int32, device=dev)
    d_counts = torch.zeros(F, dtype=torch.int32, device=dev)
    d_nraw = torch.zeros(F, dtype=torch.int32, device=dev)
    pl = pdist.all_pairs(F)
    M = len(pl)
    pairlist = torch.tensor(pl, dtype=torch.int32, device=dev)
    d_out = torch.zeros((M, CAP, 3), dtype=torch.int32, device=dev)

    def run():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.detect_batch_dev(d_frames, F, W, H, d_kp, d_desc, d_counts, d_nraw, CAP)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        eng.match_batch_dev(d_desc, d_counts, CAP, WORDS, pairlist, M, d_out, max_count=NKP)
        torch.cuda.synchronize()
        return t1 - t0, time.perf_counter() - t1

    run()
    eng.check_status()
    td, tm = run()
    eng.check_status()
    counts = np.minimum(d_counts.cpu().numpy(), NKP)
    npairs = float(sum(int(counts[a]) * int(counts[b]) for a, b in pl))
    res = {"workload": "configs[2]: %d-frame 1920x1080 sequence, all %d ordered pairs, <=%d keypoints per frame" % (F, M, NKP),
           "detect_s": td, "detect_frames_per_s": F / td, "match_s": tm, "descriptor_pairs": npairs,
           "match_pairs_per_s": npairs / tm, "end_to_end_pairs_per_s": npairs / (td + tm),
           "keypoints_min_max": [int(counts.min()), int(counts.max())]}
    # spot-check against the oracle (sorted-scan formulation, exact)
    if args.check:
        from oracle import cref
        desc = d_desc.cpu().numpy().view(np.uint32)
        out = d_out.cpu().numpy()
        rng = np.random.default_rng(0)
        ok = True
        for m in rng.choice(M, size=args.check, replace=False):
            a, b = pl[m]
            exp = cref.match_sorted(desc[a][:counts[a]], desc[b][:counts[b]])
            got = out[m][:counts[a]]
            ok &= bool((got[:, 0] == exp["k1"]).all() and (got[:, 1] == exp["k2"]).all() and (got[:, 2] == exp["dist"]).all())
        res["oracle_spot_check"] = {"pairs_checked": int(args.check), "bit_exact": ok}
        g = pdist.build_track_graph(counts, pl[:F - 1], out[:F - 1], max_dist=40)  # frame 0 against every other frame
        res["tracks_from_frame0_pairs"] = len(g.tracks())
    print(json.dumps(res))
    eng.close()


if __name__ == "__main__":
    main()
