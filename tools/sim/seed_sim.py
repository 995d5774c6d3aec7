import sys, heapq
import numpy as np
sys.path.insert(0,'/root/repo/tools/sim')
from gs_sim import Z, dist_matrix, wide_round
from collections import deque
def run(D, nseed=0, ncache=2):
    R, C = D.shape
    key = (D.astype(np.int64) << 12) | np.arange(C)[None, :]
    BIG = 1 << 40
    held = np.full(C, BIG, np.int64)
    order = np.argsort(key, axis=1, kind="stable")[:, :1 + nseed]
    cand = [list(order[i, 1:]) for i in range(R)]
    queue = deque()
    nscan = nprop = 0
    def chain(cur, k):
        nonlocal nprop
        while True:
            if k is None:
                if not cand[cur]: queue.append(cur); return
                k = int(cand[cur].pop(0))
            mine = (int(D[cur, k]) << 12) | cur
            nprop += 1
            if mine < held[k]:
                old = held[k]; held[k] = mine
                if old >= BIG: return
                cur = int(old & 0xFFF)
            k = None
    for i in range(R):
        chain(i, int(order[i, 0]))
    while queue:
        i = queue.popleft()
        nscan += 1
        mine_i = (D[i].astype(np.int64) << 12) | i
        el = np.nonzero(mine_i < held)[0]
        if len(el):
            o = el[np.argsort(key[i, el], kind="stable")[:1 + ncache]]
            cand[i] = list(o[1:])
            chain(i, int(o[0]))
    return nscan, nprop
for a, b in [(0, 1), (0, 20), (0, 63), (30, 34), (10, 50)]:
    D = dist_matrix(Z["d%d" % a], Z["d%d" % b])
    rows, cols, nacc = wide_round(D)
    Dr = D[np.ix_(rows, cols)]
    print((a, b), Dr.shape, {ns: run(Dr, ns) for ns in (0, 1, 2, 3, 7, 15)})
