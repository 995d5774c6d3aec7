#!/usr/bin/env python3
"""BASELINE configs[2]: a 64-frame 1920x1080 sequence (frame_i = frame_0 translated by
(3i, i) px, the stand-in for the absent Blender render), detect every frame, match ALL ordered pairs i < j
(2016 image pairs x up to 4096^2) on one MI355X.  A few pairs are re-checked against the CPU oracle.

The same tool runs SURVEY 8d config 4's one-GPU variant (3840x2160, 8192 keypoints, r = 30, sliding window):
  python tools/bench_sequence.py --width 3840 --height 2160 --nkp 8192 --radius 30 --frames 1024 --window 16
Frames are made ON THE DEVICE from one host-made base frame (wrap-around translation by (3i, i)), and detected in
batches of --detect-batch frames so the workspaces stay bounded."""
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

P, WORDS, THRESH = 256, 8, 0.1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--check", type=int, default=3)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--nkp", type=int, default=4096, help="lists are cut to their first NKP keypoints in NMS order")
    ap.add_argument("--radius", type=int, default=16)
    ap.add_argument("--window", type=int, default=0, help="match pairs with 0 < j - i <= window (0 = all pairs)")
    ap.add_argument("--detect-batch", type=int, default=128)
    args = ap.parse_args()
    F, W, H, NKP, RADIUS = args.frames, args.width, args.height, args.nkp, args.radius
    CAP = 2 * NKP
    dev = torch.device("cuda", 0)
    base = synth.make_frame(W, H, seed=4321, n_shapes=int(20000 * (W * H) / (1920 * 1080)))
    d_base = torch.from_numpy(base).to(dev)
    d_frames = torch.empty((F, H, W, 4), dtype=torch.uint16, device=dev)
    b64, f64 = d_base.view(torch.int64), d_frames.view(torch.int64)   # one RGBA64 pixel = one int64 (roll lacks uint16)
    for i in range(F):   # wrap-around translation on the device (config 4: "generated on device", no 68 GB of H2D)
        f64[i] = torch.roll(b64, shifts=(i % H, (3 * i) % W), dims=(0, 1))
    eng = pg.Engine(0)
    pairs_tbl = pg.make_brief_pairs(0, 50, P)
    dmap = pg.build_dewarp_map(W, H, [3e-4, 1e-7, 0, 0, 0])
    eng.set_brief_pairs(pairs_tbl)
    eng.set_detect_params(THRESH, RADIUS)
    eng.set_capacity(int((1 << 18) * max(1.0, (W * H) / (1920 * 1080))), CAP)
    eng.set_dewarp_map(dmap)
    d_kp = torch.zeros((F, CAP, 4), dtype=torch.int32, device=dev)
    d_desc = torch.zeros((F, CAP, WORDS), dtype=torch.int32, device=dev)
    d_counts = torch.zeros(F, dtype=torch.int32, device=dev)
    d_nraw = torch.zeros(F, dtype=torch.int32, device=dev)
    pl = pdist.all_pairs(F) if args.window <= 0 else [(i, j) for i in range(F) for j in range(i + 1, min(F, i + args.window + 1))]
    M = len(pl)
    pairlist = torch.tensor(pl, dtype=torch.int32, device=dev)
    d_out = torch.zeros((M, CAP, 3), dtype=torch.int32, device=dev)
    DB = max(1, args.detect_batch)

    def run():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for f0 in range(0, F, DB):
            n = min(DB, F - f0)
            eng.detect_batch_dev(d_frames[f0:f0 + n], n, W, H, d_kp[f0:f0 + n], d_desc[f0:f0 + n], d_counts[f0:f0 + n],
                                 d_nraw[f0:f0 + n], CAP)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        eng.match_batch_dev(d_desc, d_counts, CAP, WORDS, pairlist, M, d_out, max_count=NKP)
        torch.cuda.synchronize()
        return t1 - t0, time.perf_counter() - t1

    def status():
        try:
            eng.check_status()
        except pg.CapacityError as e:   # more survivors than CAP: the lists are cut anyway (max_count)
            print("note:", e, file=sys.stderr)

    run()
    status()
    td, tm = run()
    status()
    counts = np.minimum(d_counts.cpu().numpy(), NKP)
    npairs = float(sum(int(counts[a]) * int(counts[b]) for a, b in pl))
    res = {"workload": "%d-frame %dx%d sequence, %d ordered pairs (%s), <=%d keypoints per frame, r=%d"
                       % (F, W, H, M, "all i<j" if args.window <= 0 else "0<j-i<=%d" % args.window, NKP, RADIUS),
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
        nf0 = sum(1 for a, b in pl if a == 0)   # the pairs of frame 0 come first in both orderings
        res["tracks_from_frame0_pairs"] = len(pdist.tracks_host(counts, pl[:nf0], out[:nf0], max_dist=40)[0])
    print(json.dumps(res))
    eng.close()


if __name__ == "__main__":
    main()
