"""How deep does a row walk into its preference list if columns that are certainly dead after the first r rounds of deferred
acceptance are filtered out of the list beforehand?  (Design study for the list-based finish.)"""
import sys, numpy as np
from gs_sim import Z, dist_matrix, wide_round
BIG = 1 << 40

def da_rounds(D, nrounds=None):
    """Plain synchronous DA with exact 'first eligible column' per free row.  Returns held after each round and the final held."""
    R, C = D.shape
    key = (D.astype(np.int64) << 12) | np.arange(C)[None, :]
    mine = (D.astype(np.int64) << 12) | np.arange(R)[:, None]
    held = np.full(C, BIG, np.int64)
    free = np.arange(R)
    snaps = []
    nprop = 0
    while len(free):
        el = mine[free] < held[None, :]
        k = np.where(el, key[free], BIG)
        tgt = k.argmin(1)
        ok = el[np.arange(len(free)), tgt]
        fr, tg = free[ok], tgt[ok]
        nprop += len(fr)
        m = mine[fr, tg]
        newheld = held.copy()
        np.minimum.at(newheld, tg, m)
        # who is free next: proposers that did not win + bumped holders
        win = newheld[tg] == m
        bumped = held[(newheld < held) & (held < BIG)] & 0xFFF
        free = np.concatenate([fr[~win], bumped]).astype(int)
        held = newheld
        snaps.append(held.copy())
    return snaps, held, nprop

for a, b in [(0, 1), (0, 20), (0, 63), (10, 50)]:
    D = dist_matrix(Z["d%d" % a], Z["d%d" % b])
    rows, cols, nacc = wide_round(D)
    Dr = D[np.ix_(rows, cols)]
    R, C = Dr.shape
    snaps, held, nprop = da_rounds(Dr)
    key = (Dr.astype(np.int64) << 12) | np.arange(C)[None, :]
    mine = (Dr.astype(np.int64) << 12) | np.arange(R)[:, None]
    partner = np.full(R, -1)
    for j in range(C):
        if held[j] < BIG: partner[int(held[j] & 0xFFF)] = j
    pk = np.where(partner >= 0, key[np.arange(R), np.maximum(partner, 0)], BIG)
    print("pair", (a, b), "residual", Dr.shape, "rounds", len(snaps), "proposals", nprop)
    for r in (None, 0, 1, 2, 4, 8):
        if r is None: alive = np.ones_like(key, bool)
        elif r >= len(snaps): continue
        else: alive = mine < snaps[r][None, :]
        # entries before (and including) the partner that survive the filter
        depth = ((key <= pk[:, None]) & alive).sum(1)
        tot_alive = alive.sum(1)
        print("   filter after round %s: depth mean %.1f p50 %d p90 %d p99 %d p99.9 %d max %d; alive per row mean %.0f; rows deeper than 32/48/64/96: %d %d %d %d" % (
            r, depth.mean(), np.median(depth), np.percentile(depth, 90), np.percentile(depth, 99), np.percentile(depth, 99.9), depth.max(), tot_alive.mean(),
            (depth > 32).sum(), (depth > 48).sum(), (depth > 64).sum(), (depth > 96).sum()))
