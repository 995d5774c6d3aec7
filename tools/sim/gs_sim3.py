"""Design L: per-row lists = all columns alive after round 0 with key <= Bnd_i, Bnd_i from residue-class minima.
Counts list lengths, bytes, rows that exhaust their list (fallback rescans) and the DA's rounds/proposals."""
import sys, numpy as np
from gs_sim import Z, dist_matrix, wide_round
BIG = 1 << 40

def run(Dr, nclass=32, kth=1, ncache=2, use_filter=True):
    R, C = Dr.shape
    key = (Dr.astype(np.int64) << 12) | np.arange(C)[None, :]
    mine = (Dr.astype(np.int64) << 12) | np.arange(R)[:, None]
    # pass A: nearest column, class minima
    cls = np.arange(C) % nclass
    cm = np.full((R, nclass), BIG, np.int64)
    cm2 = np.full((R, nclass), BIG, np.int64)
    for c in range(nclass):
        sub = np.sort(key[:, cls == c], axis=1)
        cm[:, c] = sub[:, 0]
        if sub.shape[1] > 1: cm2[:, c] = sub[:, 1]
    bnd = (cm if kth == 1 else cm2).max(1)
    near = key.argmin(1)
    held = np.full(C, BIG, np.int64)
    np.minimum.at(held, near, mine[np.arange(R), near])
    held0 = held.copy()
    # pass B: lists
    inlist = key <= bnd[:, None]
    if use_filter: inlist &= mine < held0[None, :]
    # holders of round 0 keep their own column in the list (mine == held0 is not < held0): irrelevant, they hold it
    llen = inlist.sum(1)
    # DA with lists
    win0 = held0[near] == mine[np.arange(R), near]
    free = list(np.nonzero(~win0)[0])
    cache = [[] for _ in range(R)]
    exhausted = np.zeros(R, bool)
    stats = dict(rounds=1, proposals=R, liststeps=0, fallbacks=0, fb_rows=0)
    while free:
        stats["rounds"] += 1
        props = {}
        nxt = []
        h0 = held.copy()
        for i in free:
            tgt = None
            if not exhausted[i]:
                stats["liststeps"] += 1
                el = np.nonzero(inlist[i] & (mine[i] < h0))[0]
                if len(el): tgt = el[key[i, el].argmin()]
                else: exhausted[i] = True; stats["fb_rows"] += 1
            if tgt is None:
                while cache[i]:
                    j = cache[i].pop(0)
                    if mine[i, j] < h0[j]: tgt = j; break
            if tgt is None:
                stats["fallbacks"] += 1
                el = np.nonzero((mine[i] < h0) & (key[i] > bnd[i]))[0]
                if len(el) == 0: continue
                o = np.argsort(key[i, el], kind="stable")[:1 + ncache]
                tgt = el[o[0]]; cache[i] = list(el[o[1:]])
            props.setdefault(tgt, []).append(i)
            stats["proposals"] += 1
        for j, lst in props.items():
            best = min(lst, key=lambda i: mine[i, j])
            for i in lst:
                if i != best: nxt.append(i)
            if mine[best, j] < held[j]:
                if held[j] < BIG: nxt.append(int(held[j] & 0xFFF))
                held[j] = mine[best, j]
            else: nxt.append(best)
        free = nxt
    return stats, llen, (key <= bnd[:, None]).sum(1)

for a, b in [(0, 1), (0, 20), (0, 63), (10, 50)]:
    D = dist_matrix(Z["d%d" % a], Z["d%d" % b])
    rows, cols, nacc = wide_round(D)
    Dr = D[np.ix_(rows, cols)]
    print("pair", (a, b), "residual", Dr.shape)
    for nclass, kth in ((32, 1), (32, 2), (64, 1), (64, 2), (128, 1)):
        st, llen, ulen = run(Dr, nclass, kth)
        print("   classes %3d kth %d: list len mean %.0f p50 %d p99 %d max %d (unfiltered mean %.0f max %d) KB %.0f | rounds %d proposals %d list-steps %d rows exhausted %d fallback scans %d" % (
            nclass, kth, llen.mean(), np.median(llen), np.percentile(llen, 99), llen.max(), ulen.mean(), ulen.max(), llen.sum() * 4 / 1024,
            st["rounds"], st["proposals"], st["liststeps"], st["fb_rows"], st["fallbacks"]))
