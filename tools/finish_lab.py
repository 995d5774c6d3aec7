#!/usr/bin/env python3
"""Matcher A/B bench on the bench sequence's own descriptor sets (developer tool).

Detects the 64-frame sequence of bench.py once, then times the matcher over all 2016 image pairs with the library named by
PGX_LIB (one library per process): pipelined (the shipped three-stream form) and stage by stage (pgx_profile_serialize),
and prints the sha256 of the match lists, so that builds can be compared bit for bit.  --check N re-checks N image pairs
against the CPU oracle (test infrastructure; never part of a timed region).

  PGX_LIB=build/variants/libpgx_x.so python tools/finish_lab.py --steps 10 --tag x
"""
import argparse
import hashlib
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
    ap.add_argument("--nkp", type=int, default=4096)
    ap.add_argument("--radius", type=int, default=16)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--check", type=int, default=0)
    ap.add_argument("--chunk", type=int, default=0)
    ap.add_argument("--tag", default="")
    ap.add_argument("--no-serial", action="store_true", help="pipelined timing only (for a kernel trace)")
    ap.add_argument("--random", type=int, default=0, help="instead of the sequence: this many image pairs of random 4096-point sets")
    args = ap.parse_args()
    F, W, H, NKP = args.frames, 1920, 1080, args.nkp
    CAP = NKP
    dev = torch.device("cuda", 0)
    eng = pg.Engine(0)
    if args.chunk:
        eng.set_match_chunk(args.chunk)
    if args.random:
        M = args.random
        F = 2 * M
        descs = [synth.random_descriptors(NKP, 8, 10 + k) for k in range(F)]
        d_desc = torch.from_numpy(np.stack(descs).view(np.int32)).to(dev)
        d_counts = torch.full((F,), NKP, dtype=torch.int32, device=dev)
        pl = [(2 * m, 2 * m + 1) for m in range(M)]
    else:
        base = synth.make_frame(W, H, seed=4321, n_shapes=20000)
        d_base = torch.from_numpy(base).to(dev)
        d_frames = torch.empty((F, H, W, 4), dtype=torch.uint16, device=dev)
        b64, f64 = d_base.view(torch.int64), d_frames.view(torch.int64)
        for i in range(F):
            f64[i] = torch.roll(b64, shifts=(i % H, (3 * i) % W), dims=(0, 1))
        eng.set_brief_pairs(pg.make_brief_pairs(0, 50, P))
        eng.set_detect_params(THRESH, args.radius)
        eng.set_capacity(1 << 18, NKP)
        eng.set_dewarp_map(pg.build_dewarp_map(W, H, [3e-4, 1e-7, 0, 0, 0]))
        d_kp = torch.zeros((F, CAP, 4), dtype=torch.int32, device=dev)
        d_desc = torch.zeros((F, CAP, WORDS), dtype=torch.int32, device=dev)
        d_counts = torch.zeros(F, dtype=torch.int32, device=dev)
        d_nraw = torch.zeros(F, dtype=torch.int32, device=dev)
        eng.detect_batch_dev(d_frames, F, W, H, d_kp, d_desc, d_counts, d_nraw, CAP)
        torch.cuda.synchronize()
        del d_frames
        pl = pdist.all_pairs(F)
    M = len(pl)
    pairlist = torch.tensor(pl, dtype=torch.int32, device=dev)
    d_out = torch.zeros((M, CAP, 3), dtype=torch.int32, device=dev)

    def match():
        eng.match_batch_dev(d_desc, d_counts, CAP, WORDS, pairlist, M, d_out, max_count=NKP)

    for _ in range(2):
        match()
    torch.cuda.synchronize()
    eng.check_status()
    eng.debug_counters()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        match()
    torch.cuda.synchronize()
    ms_pipe = (time.perf_counter() - t0) * 1e3 / args.steps
    dbg = [x / float(args.steps * M) for x in eng.debug_counters()]
    eng.check_status()
    out = d_out.cpu().numpy()
    digest = hashlib.sha256(out.tobytes()).hexdigest()[:16]
    # the stages in order on one stream, no profiling events
    eng.profile_serialize(True)
    match()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        match()
    torch.cuda.synchronize()
    ms_inorder = (time.perf_counter() - t0) * 1e3 / args.steps
    eng.profile_serialize(False)
    if args.no_serial:
        print(json.dumps({"tag": args.tag, "pairs": M, "chunk": args.chunk or 256, "ms_pipelined": round(ms_pipe, 4), "sha": digest}), flush=True)
        eng.close()
        return
    # stage by stage
    eng.profile_serialize(True)
    eng.profile_reset()
    eng.profile_enable(True)
    match()
    torch.cuda.synchronize()
    eng.debug_counters()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        match()
    torch.cuda.synchronize()
    ms_serial = (time.perf_counter() - t0) * 1e3 / args.steps
    eng.profile_enable(False)
    dbg_serial = [x / float(args.steps * M) for x in eng.debug_counters()]
    kern = {}
    for name in ("match_init", "ham_argmin", "match_select", "tail_rows", "match_finish"):
        n, t = eng.profile_get(name)
        if n:
            kern[name] = round(t / (args.steps + 1), 4)
    eng.profile_serialize(False)
    eng.check_status()
    digest2 = hashlib.sha256(d_out.cpu().numpy().tobytes()).hexdigest()[:16]
    res = {"tag": args.tag, "lib": os.path.basename(os.environ.get("PGX_LIB", "libpgx.so")), "pairs": M, "chunk": args.chunk or 256,
           "ms_pipelined": round(ms_pipe, 4), "ms_in_order": round(ms_inorder, 4), "ms_serial": round(ms_serial, 4), "kernels_ms": kern, "sha": digest, "sha_serial": digest2,
           "dbg_per_pair": [round(x, 2) for x in dbg], "dbg_per_pair_serial": [round(x, 2) for x in dbg_serial]}
    if args.check:
        from oracle import cref
        desc = d_desc.cpu().numpy().view(np.uint32)
        counts = np.minimum(d_counts.cpu().numpy(), NKP)
        rng = np.random.default_rng(0)
        ok = True
        for m in rng.choice(M, size=args.check, replace=False):
            a, b = pl[m]
            exp = cref.match_sorted(desc[a][:counts[a]], desc[b][:counts[b]])
            got = out[m][:counts[a]]
            ok &= bool((got[:, 0] == exp["k1"]).all() and (got[:, 1] == exp["k2"]).all() and (got[:, 2] == exp["dist"]).all())
        res["oracle_ok"] = ok
    print(json.dumps(res), flush=True)
    eng.close()


if __name__ == "__main__":
    main()
