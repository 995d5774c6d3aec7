"""Two-tier lists.  Tier 1: bound per row from the wide round's 32 column classes over ALL columns (class = original column
index mod 32), list = residual columns with key <= bound (unfiltered).  DA until every free row is stalled; tier 2 for the
stalled rows: all columns alive under the held keys of that moment and beyond the old bound; repeat."""
import sys, numpy as np
from gs_sim import Z, dist_matrix, wide_round
BIG = 1 << 40

def run(D, nclass=32, cap2=256):
    n1, n2 = D.shape
    rows, cols, nacc = wide_round(D)
    keyall = (D.astype(np.int64) << 12) | np.arange(n2)[None, :]
    # class minima over all columns, per row
    cm = np.full((n1, nclass), BIG, np.int64)
    for c in range(nclass):
        cm[:, c] = keyall[:, c::nclass].min(1)
    bnd_all = cm.max(1)                       # in terms of (d, original column)
    Dr = D[np.ix_(rows, cols)]
    R, C = Dr.shape
    # the key order by original column index equals the order by residual position (stable compaction)
    key = (Dr.astype(np.int64) << 12) | np.arange(C)[None, :]
    korig = (Dr.astype(np.int64) << 12) | cols[None, :]
    mine = (Dr.astype(np.int64) << 12) | np.arange(R)[:, None]
    bnd = bnd_all[rows]
    inlist = korig <= bnd[:, None]
    l1 = inlist.sum(1)
    held = np.full(C, BIG, np.int64)
    free = list(range(R))
    stalled = []
    st = dict(rounds=0, proposals=0, liststeps=0, iters=1, stalled=[], l2len=[], overflow=0, done_rows=0)
    complete = np.zeros(R, bool)   # list holds every alive column (tier 2 without overflow)
    while True:
        while free:
            st["rounds"] += 1
            props = {}; nxt = []; h0 = held.copy()
            for i in free:
                st["liststeps"] += 1
                el = np.nonzero(inlist[i] & (mine[i] < h0))[0]
                if len(el) == 0:
                    if complete[i]: st["done_rows"] += 1
                    else: stalled.append(i)
                    continue
                tgt = el[key[i, el].argmin()]
                props.setdefault(tgt, []).append(i); st["proposals"] += 1
            for j, lst in props.items():
                best = min(lst, key=lambda i: mine[i, j])
                for i in lst:
                    if i != best: nxt.append(i)
                if mine[best, j] < held[j]:
                    if held[j] < BIG: nxt.append(int(held[j] & 0xFFF))
                    held[j] = mine[best, j]
                else: nxt.append(best)
            free = nxt
        if not stalled: break
        st["iters"] += 1; st["stalled"].append(len(stalled))
        for i in stalled:
            alive = (mine[i] < held) & (korig[i] > bnd[i])
            n = alive.sum(); st["l2len"].append(int(n))
            if n > cap2:
                st["overflow"] += 1
                ks = np.sort(korig[i, alive])[cap2 - 1]
                inlist[i] = alive & (korig[i] <= ks); bnd[i] = ks
            else:
                inlist[i] = alive; complete[i] = True
        free = stalled; stalled = []
    return st, l1, R, C

tot = dict(l1=0, rows=0)
for a, b in [(0, 1), (0, 8), (0, 20), (0, 40), (0, 63), (10, 50), (30, 34), (5, 60)]:
    D = dist_matrix(Z["d%d" % a], Z["d%d" % b])
    for nclass in (32, 64):
        st, l1, R, C = run(D, nclass)
        l2 = np.array(st["l2len"]) if st["l2len"] else np.array([0])
        print("pair %s residual %dx%d classes %d: tier-1 len mean %.0f p50 %d max %d (%.0f KB) | iters %d stalled %s tier-2 len mean %.0f max %d (%.0f KB) overflow %d | rounds %d proposals %d list-steps %d" % (
            (a, b), R, C, nclass, l1.mean(), np.median(l1), l1.max(), l1.sum() * 4 / 1024, st["iters"], st["stalled"], l2.mean(), l2.max(), l2.sum() * 4 / 1024, st["overflow"],
            st["rounds"], st["proposals"], st["liststeps"]))
