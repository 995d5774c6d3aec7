"""Event-driven model of the queue-driven finish (k_match_gs, round 4): NW wavefronts take free rows off a queue, a scan costs
T_SCAN, a chain step T_STEP; reports scans, proposals and the makespan, i.e. how much of a pair's time is the latency-bound
tail.  Design study only (numpy; bench-sequence descriptors from make_seq.py)."""
import heapq, sys
import numpy as np
from gs_sim import Z, dist_matrix, wide_round

def run(D, nw=16, t_scan=1.0, t_step=0.1, ncache=2):
    R, C = D.shape
    key = (D.astype(np.int64) << 12) | np.arange(C)[None, :]
    BIG = 1 << 40
    held = np.full(C, BIG, np.int64)
    cand = [[] for _ in range(R)]
    # round 0: every row proposes to its nearest column
    q = []
    near = key.argmin(1)
    for i in range(R):
        j = near[i]; mine = (int(D[i, j]) << 12) | i
        if mine < held[j]:
            if held[j] < BIG: q.append(int(held[j] & 0xFFF))
            held[j] = mine
        else: q.append(i)
    from collections import deque
    queue = deque(q)
    waves = [(0.0, w) for w in range(nw)]   # (time free, id)
    heapq.heapify(waves)
    events = []  # (time, row) rows that become queued at a time
    nscan = nprop = 0
    tmax = 0.0
    pend = []  # heap of (ready time, row)
    busy_time = 0.0
    while queue or pend:
        t, w = heapq.heappop(waves)
        while pend and pend[0][0] <= t: queue.append(heapq.heappop(pend)[1])
        if not queue:
            if not pend: break
            t = max(t, pend[0][0])
            while pend and pend[0][0] <= t: queue.append(heapq.heappop(pend)[1])
        i = queue.popleft()
        nscan += 1
        mine_i = (D[i].astype(np.int64) << 12) | i
        el = np.nonzero(mine_i < held)[0]
        tt = t + t_scan
        if len(el):
            o = el[np.argsort(key[i, el], kind="stable")[:1 + ncache]]
            cand[i] = list(o[1:])
            cur, k = i, int(o[0])
            while True:
                if k is None:
                    if not cand[cur]: heapq.heappush(pend, (tt, cur)); break
                    k = int(cand[cur].pop(0))
                mine = (int(D[cur, k]) << 12) | cur
                nprop += 1; tt += t_step
                if mine < held[k]:
                    old = held[k]; held[k] = mine
                    if old >= BIG: break
                    cur = int(old & 0xFFF)
                k = None
        busy_time += tt - t
        tmax = max(tmax, tt)
        heapq.heappush(waves, (tt, w))
    return dict(R=R, C=C, scans=nscan, props=nprop, makespan=tmax, ideal=busy_time / nw), held

if __name__ == "__main__":
    pairs = [(0, 1), (0, 20), (0, 63), (30, 34), (10, 50)]
    for a, b in pairs:
        D = dist_matrix(Z["d%d" % a], Z["d%d" % b])
        rows, cols, nacc = wide_round(D)
        Dr = D[np.ix_(rows, cols)]
        for nw in (16, 64):
            for nc in (2, 4):
                st, _ = run(Dr, nw=nw, ncache=nc)
                print((a, b), "nw", nw, "cache", nc, {k: (round(v, 1) if isinstance(v, float) else v) for k, v in st.items()})
