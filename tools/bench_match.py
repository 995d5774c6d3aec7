#!/usr/bin/env python3
"""Matcher-only micro-bench: M image pairs of N x N descriptors resident in HBM.
kind = random (worst case for rounds) | true (permuted copy with 15% bit flips, 1 round)."""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import photogrammetry_amd as pg
from photogrammetry_amd import synth


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=4096)
    ap.add_argument("--pairs", type=int, default=32)
    ap.add_argument("--kind", default="true")
    ap.add_argument("--steps", type=int, default=10)
    args = ap.parse_args()
    N, M = args.n, args.pairs
    dev = torch.device("cuda", 0)
    eng = pg.Engine(0)
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    descs = []
    for m in range(M):
        if args.kind == "true":
            a, b, _ = synth.true_match_descriptors(N, 8, 10 + m)
        else:
            a, b = synth.random_descriptors(N, 8, 10 + m), synth.random_descriptors(N, 8, 1000 + m)
        descs += [a, b]
    d_desc = torch.from_numpy(np.stack(descs).view(np.int32)).to(dev)
    d_counts = torch.full((2 * M,), N, dtype=torch.int32, device=dev)
    pairlist = torch.tensor([[2 * m, 2 * m + 1] for m in range(M)], dtype=torch.int32, device=dev)
    d_out = torch.zeros((M, N, 3), dtype=torch.int32, device=dev)
    for _ in range(2):
        eng.match_batch_dev(d_desc, d_counts, N, 8, pairlist, M, d_out)
    torch.cuda.synchronize()
    eng.check_status()
    eng.profile_reset()
    eng.profile_enable(True)
    import time
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.match_batch_dev(d_desc, d_counts, N, 8, pairlist, M, d_out)
    torch.cuda.synchronize()   # device-wide: the context runs on its own stream when torch's is the null stream
    ms = (time.perf_counter() - t0) * 1e3 / args.steps
    eng.profile_enable(False)
    kern = {}
    for name in ("match_init", "ham_argmin", "match_select", "tail_fill", "tail_rows", "match_finish"):
        n, t = eng.profile_get(name)
        if n:
            kern[name] = round(t / args.steps, 4)
    rounds, evals, ev0 = eng.match_stats()
    ham_ms = kern.get("ham_argmin", 0)
    out = {"n": N, "pairs": M, "kind": args.kind, "ms_per_step": ms, "pairs_per_s": M * N * N / (ms * 1e-3),
           "kernels_ms": kern, "wide_rounds": rounds, "evaluations": evals,
           "ham_TOPs": evals * 512 / (ham_ms * 1e-3) / 1e12 if ham_ms else None,
           "ham_frac_of_fp4_peak": evals * 512 / (ham_ms * 1e-3) / 10e15 if ham_ms else None}   # dense FP4 MFMA peak: 10 POP/s
    print(json.dumps(out))


if __name__ == "__main__":
    main()
