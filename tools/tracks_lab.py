#!/usr/bin/env python3
"""Developer tool (GPU box): the track graph alone on the bench job's own match lists.

    python tools/tracks_lab.py [--frames 64] [--reps 50] [--max-dist 64]

Runs detect + match of the bench sequence once, then pgx_tracks_dev `reps` times on the resident lists and prints the
average time per call (HIP events through torch on the job's stream), the summary, a sha256 of the result arrays (to compare
builds bit for bit) and the oracle check.  Under `rocprofv3 --kernel-trace --stats` the k_trk_* rows give the split."""
import argparse
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--reps", type=int, default=50)
    ap.add_argument("--max-dist", type=int, default=64)
    ap.add_argument("--no-check", action="store_true")
    args = ap.parse_args()
    import numpy as np
    import torch
    import bench
    import photogrammetry_amd as pg
    from photogrammetry_amd import dist as pdist
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    W, H, NKP = bench.W, bench.H, bench.NKP
    F = args.frames
    e = pg.Engine(0)
    e.set_brief_pairs(pg.make_brief_pairs(0, 50, 256))
    e.set_detect_params(bench.THRESH, bench.RADIUS)
    e.set_capacity(1 << 18, NKP)
    e.set_dewarp_map(pg.build_dewarp_map(W, H, [3e-4, 1e-7, 0, 0, 0]))
    pl = pdist.all_pairs(F)
    stream = torch.cuda.Stream(device=dev)
    job = pdist.ShardedSequence(e, W, H, F, pl, NKP, 8, dev, stream=stream, tracks={"max_dist": args.max_dist, "min_len": 2})
    with torch.cuda.stream(stream):
        base = torch.from_numpy(bench.base_frame(W, H, 4321)).to(dev)
        d_frames = bench.roll_frames(torch, base, [(3 * i, i) for i in range(F)])
    torch.cuda.synchronize()
    job.step(d_frames)
    e.check_status()
    with torch.cuda.stream(stream):
        for _ in range(3):
            job._build_tracks(0)
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record(stream)
        for _ in range(args.reps):
            job._build_tracks(0)
        t1.record(stream)
    torch.cuda.synchronize()
    e.check_status()
    summ = job.track_summary()
    nt, nn = summ["n_tracks"], summ["n_nodes"]
    h = hashlib.sha256()
    for t in (job.trk_offsets[:nt + 1], job.trk_nodes[:nn], job.track_of):
        h.update(t.cpu().numpy().tobytes())
    res = {"ms_per_call": t0.elapsed_time(t1) / args.reps, "frames": F, "image_pairs": len(pl), "entries": len(pl) * NKP,
           "summary": summ, "sha256": h.hexdigest()}
    if not args.no_check:
        from oracle import tracks_np
        counts = job.counts()
        m = job.out_all.cpu().numpy()
        e_off, e_nodes, e_tof, e_s = tracks_np.tracks_arrays(counts, pl, m, NKP, args.max_dist, 2)
        res["oracle_ok"] = bool(summ == e_s and (job.trk_offsets[:nt + 1].cpu().numpy() == e_off).all()
                                and (job.trk_nodes[:nn].cpu().numpy() == e_nodes).all() and (job.track_of.cpu().numpy() == e_tof).all())
    print(json.dumps(res))
    e.close()


if __name__ == "__main__":
    main()
