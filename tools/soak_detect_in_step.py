#!/usr/bin/env python3
"""Soak of the detect chain (dewarp -> grey -> FAST -> NMS -> BRIEF) under the schedule bench.py runs by default: two jobs in
flight, the detect chain of step k + 1 beside the distance kernel of step k (developer tool; DESIGN.md section 4, "the
predicated-load hazard": the shipped NMS kernels are qualified alone by tests/nmsexp/diag.py -- this is the same question asked
where they really run, with another job's kernels on the same CUs).

Every step detects the same 64 frames, so every step must produce the same keypoints, counts and descriptors: after each detect
a comparison with the reference step is enqueued on the job's own stream (no synchronisation inside the loop) and mismatching
steps are counted on the device.  The reference step itself is checked against the CPU oracle for EVERY frame (test
infrastructure; outside any timing).

  python tools/soak_detect_in_step.py --steps 1500
"""
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

W, H, F, NKP, WORDS, P = 1920, 1080, 64, 4096, 8, 256


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=1500)
    ap.add_argument("--radius", type=int, default=16)
    ap.add_argument("--oracle-frames", type=int, default=64)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.init()
    pairs = pg.make_brief_pairs(0, 50, P)
    dmap = pg.build_dewarp_map(W, H, [3e-4, 1e-7, 0, 0, 0])
    base = synth.make_frame(W, H, seed=4321, n_shapes=20000)
    d_base = torch.from_numpy(base).to(dev)
    d_frames = torch.empty((F, H, W, 4), dtype=torch.uint16, device=dev)
    for i in range(F):
        d_frames.view(torch.int64)[i] = torch.roll(d_base.view(torch.int64), shifts=(i % H, (3 * i) % W), dims=(0, 1))
    pl = pdist.all_pairs(F)
    engs, jobs = [], []
    for _ in range(2):
        e = pg.Engine(0)
        e.set_brief_pairs(pairs)
        e.set_detect_params(np.float32(0.1), args.radius)
        e.set_capacity(1 << 18, NKP)
        e.set_dewarp_map(dmap)
        engs.append(e)
        jobs.append(pdist.ShardedSequence(e, W, H, F, pl, NKP, WORDS, dev, stream=torch.cuda.Stream(device=dev)))
    # reference step, one job alone
    jobs[0].step(d_frames)
    torch.cuda.synchronize()
    engs[0].check_status()
    ref_kp, ref_cnt, ref_desc = jobs[0].kp_l.clone(), jobs[0].counts_all.clone(), jobs[0].desc_all.clone()
    ref_out = jobs[0].out_all.clone()
    bad = torch.zeros(4, dtype=torch.int32, device=dev)   # steps whose keypoints / counts / descriptors / match lists differ

    def gates(s):
        return (engs[(s - 1) % 2], None, 2) if s > 0 else None   # --gate none,rows

    def check_front(j):
        with torch.cuda.stream(j.stream):
            bad[0] += (j.kp_l != ref_kp).any().to(torch.int32)
            bad[1] += (j.counts_all != ref_cnt).any().to(torch.int32)
            bad[2] += (j.desc_all != ref_desc).any().to(torch.int32)

    def check_back(j):
        with torch.cuda.stream(j.stream):
            bad[3] += (j.out_all != ref_out).any().to(torch.int32)

    n = args.steps
    t0 = time.perf_counter()
    jobs[0].front(d_frames, gates(0))
    check_front(jobs[0])
    for s in range(n):
        if s + 1 < n:
            jobs[(s + 1) % 2].front(d_frames, gates(s + 1))
            check_front(jobs[(s + 1) % 2])
        jobs[s % 2].back(gates(s))
        check_back(jobs[s % 2])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    for e in engs:
        e.check_status()
    b = bad.cpu().tolist()
    # the reference step against the oracle, frame by frame
    from oracle import cref
    frames_h = d_frames[:args.oracle_frames].cpu().numpy()
    kp_h, cnt_h, desc_h = ref_kp.cpu().numpy(), ref_cnt.cpu().numpy(), ref_desc.cpu().numpy().view(np.uint32)
    wrong = []
    for f in range(args.oracle_frames):
        g = cref.gray(cref.apply_distortion(frames_h[f], dmap))
        raw = cref.detect(g, np.float32(0.1))
        kept = raw[cref.nms(raw, args.radius)][:NKP]
        n_f = int(cnt_h[f])
        ok = n_f == len(kept) and np.array_equal(kp_h[f, :n_f, 0], kept["x"]) and np.array_equal(kp_h[f, :n_f, 1], kept["y"])
        if ok:
            d = cref.brief(g, np.stack([kept["x"], kept["y"]], 1), pairs)
            ok = np.array_equal(desc_h[f, :n_f], d)
        if not ok:
            wrong.append(f)
    res = {"steps": n, "frames_detected_in_step": n * F, "radius": args.radius, "seconds": dt, "ms_per_step": dt / n * 1e3,
           "steps_with_different_keypoints": b[0], "steps_with_different_counts": b[1], "steps_with_different_descriptors": b[2],
           "steps_with_different_match_lists": b[3], "reference_frames_checked_against_oracle": args.oracle_frames,
           "reference_frames_wrong": wrong,
           "schedule": "two jobs in flight, gate none,rows; every comparison enqueued on the job's own stream (they add about 0.3 ms per step)"}
    print(json.dumps(res))
    for e in engs:
        e.close()
    return 0 if not any(b) and not wrong else 1


if __name__ == "__main__":
    sys.exit(main())
