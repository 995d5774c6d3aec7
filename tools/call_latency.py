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
def frame_sets():
    """Descriptor sets of real detect runs: a bench-like synthetic 1920x1080 frame and eight translations of it."""
    eng.set_brief_pairs(pg.make_brief_pairs(0, 50, 256))
    eng.set_detect_params(0.1, 16)
    eng.set_capacity(1 << 18, N)
    base = synth.make_frame(1920, 1080, seed=4321, n_shapes=20000)
    d0 = eng.detect(base, capacity=N)[1]
    return [(d0, eng.detect(synth.shift_frame(base, 3 * i + 2 * (i % 3), i), capacity=N)[1]) for i in (1, 2, 5, 11, 20, 37, 50, 63)]


def timed(fn, reps=5, trials=5):
    """Median over `trials` of the mean time of `reps` calls (ms)."""
    fn()
    out = []
    for _ in range(trials):
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        out.append((time.perf_counter() - t0) / reps * 1e3)
    return float(np.median(out))


for kind in ("random", "true", "frames"):
    if kind == "random":
        sets = [(synth.random_descriptors(N, 8, 10 + m), synth.random_descriptors(N, 8, 1000 + m)) for m in range(8)]
    elif kind == "true":
        sets = [synth.true_match_descriptors(N, 8, 10 + m)[:2] for m in range(8)]
    else:
        sets = frame_sets()

    def one_by_one():
        for a, b in sets:
            eng.match(a, b)
    res["pgx_match_ms_" + kind] = timed(one_by_one) / len(sets)
    descs = [x for ab in sets for x in ab]
    for M in (8, 64):
        pl = [(2 * (m % 8), 2 * (m % 8) + 1) if (m // 8) % 2 == 0 else (2 * (m % 8) + 1, 2 * (m % 8)) for m in range(M)]
        res["batch%d_ms_per_pair_%s" % (M, kind)] = timed(lambda: eng.match_batch(descs, pl)) / M
print(json.dumps({k: round(v, 4) for k, v in res.items()}))
