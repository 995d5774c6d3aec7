#!/usr/bin/env python3
"""Latency of the host-buffer entry points on one GPU: pgx_match (one image pair per call) and pgx_match_batch at 8 and 64 image
pairs per call, N = 4096, random (many rounds) and true-match (one round) descriptor sets.  DESIGN 9 quotes it."""
import sys, os, time, json, numpy as np
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import photogrammetry_amd as pg
from photogrammetry_amd import synth
eng = pg.Engine(0)
N = 4096
res = {}
for kind in ("random", "true"):
    if kind == "random":
        sets = [(synth.random_descriptors(N, 8, 10 + m), synth.random_descriptors(N, 8, 1000 + m)) for m in range(8)]
    else:
        sets = [synth.true_match_descriptors(N, 8, 10 + m)[:2] for m in range(8)]
    for a, b in sets[:2]:
        eng.match(a, b)
    t0 = time.perf_counter()
    for rep in range(5):
        for a, b in sets:
            eng.match(a, b)
    res["pgx_match_ms_" + kind] = (time.perf_counter() - t0) / 40 * 1e3
    descs = [x for ab in sets for x in ab]
    for M in (8, 64):
        pl = [(2 * (m % 8), 2 * (m % 8) + 1) for m in range(M)]
        eng.match_batch(descs, pl)
        t0 = time.perf_counter()
        for rep in range(5):
            eng.match_batch(descs, pl)
        res["batch%d_ms_per_pair_%s" % (M, kind)] = (time.perf_counter() - t0) / 5 / M * 1e3
print(json.dumps({k: round(v, 4) for k, v in res.items()}))
