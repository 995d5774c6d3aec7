"""Simulation of the matcher's tail on the bench sequence's descriptors (tools/sim/make_seq.py): how many rounds, proposals
and row re-reads the deferred-acceptance finish needs under different candidate-cache policies.  Design study only."""
import sys, numpy as np
Z = np.load("/tmp/pgx_sim/seq_desc.npz")
NONE = np.uint32(0xFFFFFFFF)

def dist_matrix(a, b):
    A = np.unpackbits(a.view(np.uint8), axis=1).astype(np.float32)
    B = np.unpackbits(b.view(np.uint8), axis=1).astype(np.float32)
    return (A @ (1 - B).T + (1 - A) @ B.T).astype(np.int32)

def wide_round(D):
    n1, n2 = D.shape
    rk = D * 4096 + np.arange(n2)[None, :]
    ck = D * 4096 + np.arange(n1)[:, None]
    rb = rk.argmin(1); cb = ck.argmin(0)
    acc = cb[rb] == np.arange(n1)
    rows = np.nonzero(~acc)[0]
    colacc = np.zeros(n2, bool); colacc[rb[acc]] = True
    cols = np.nonzero(~colacc)[0]
    return rows, cols, acc.sum()

def da_sim(D, ncache=2, prefix=0, verbose=False):
    """Synchronous rounds like k_match_gs.  D: residual R x C.  prefix: unfiltered K-prefix available per row before any scan."""
    R, C = D.shape
    key = (D.astype(np.int64) << 12) | np.arange(C)[None, :]        # row-side ranking key per (i, j)
    held = np.full(C, 1 << 40, np.int64)                              # (d << 12 | row)
    order = None
    if prefix:
        order = np.argsort(key, axis=1, kind="stable")[:, :prefix]   # unfiltered prefix
    ppos = np.zeros(R, int)
    cache = [[] for _ in range(R)]
    free = list(range(R))
    stats = dict(rounds=0, proposals=0, rescans=0, per_round=[], walk=0)
    first = True
    done = np.zeros(R, bool)
    while free:
        stats["rounds"] += 1
        props = {}
        nresc = 0
        nxt = []
        held0 = held.copy()
        for i in free:
            mine_of = lambda j: (int(D[i, j]) << 12) | i
            target = None
            # cached (filtered) candidates first
            while cache[i]:
                j = cache[i].pop(0)
                if mine_of(j) < held0[j]:
                    target = j; break
            if target is None and order is not None:
                while ppos[i] < order.shape[1]:
                    j = order[i, ppos[i]]; ppos[i] += 1; stats["walk"] += 1
                    if mine_of(j) < held0[j]:
                        target = j; break
            if target is None:
                nresc += 1
                mine = (D[i].astype(np.int64) << 12) | i
                el = np.nonzero(mine < held0)[0]
                if order is not None:   # everything in the prefix has been consumed: only keys beyond it
                    pass
                if len(el) == 0:
                    done[i] = True; continue
                ks = key[i, el]; o = np.argsort(ks, kind="stable")[:1 + ncache]
                target = el[o[0]]; cache[i] = list(el[o[1:]])
            props.setdefault(target, []).append(i)
            stats["proposals"] += 1
        for j, lst in props.items():
            best = min(lst, key=lambda i: (int(D[i, j]) << 12) | i)
            mine = (int(D[best, j]) << 12) | best
            for i in lst:
                if i != best: nxt.append(i)
            if mine < held[j]:
                if held[j] < (1 << 40): nxt.append(int(held[j] & 0xFFF))
                held[j] = mine
            else:
                nxt.append(best)
        stats["rescans"] += nresc
        stats["per_round"].append((len(free), nresc))
        free = nxt
    return stats, held

def full_walk_depth(D, held_final):
    """If every row had its complete sorted preference list: entries consumed per row = rank of its final partner (or C)."""
    R, C = D.shape
    key = (D.astype(np.int64) << 12) | np.arange(C)[None, :]
    partner = np.full(R, -1)
    for j in range(C):
        if held_final[j] < (1 << 40): partner[int(held_final[j] & 0xFFF)] = j
    depth = np.empty(R, int)
    for i in range(R):
        if partner[i] < 0: depth[i] = C
        else: depth[i] = (key[i] < key[i, partner[i]]).sum() + 1
    return depth

if __name__ == "__main__":
    pairs = [(0, 1), (0, 8), (0, 20), (0, 40), (0, 63), (30, 34), (10, 50)]
    for a, b in pairs:
        D = dist_matrix(Z["d%d" % a], Z["d%d" % b])
        rows, cols, nacc = wide_round(D)
        Dr = D[np.ix_(rows, cols)]
        print("pair", (a, b), "N", D.shape, "accepted in round 1:", nacc, "residual", Dr.shape)
        for nc, pf in ((2, 0), (4, 0), (2, 8), (2, 16), (2, 32), (2, 64)):
            st, held = da_sim(Dr, ncache=nc, prefix=pf)
            pr = st["per_round"]
            print("   cache %d prefix %2d: rounds %3d proposals %5d rescans %5d walk %6d  first rounds (free, rescans): %s" % (nc, pf, st["rounds"], st["proposals"], st["rescans"], st["walk"], pr[:6]))
        dep = full_walk_depth(Dr, held)
        print("   full-list walk depth: mean %.1f median %d p90 %d p99 %d max %d, unmatched rows %d; sum %d" % (dep.mean(), np.median(dep), np.percentile(dep, 90), np.percentile(dep, 99), dep.max(), (dep == Dr.shape[1]).sum(), dep.sum()))
