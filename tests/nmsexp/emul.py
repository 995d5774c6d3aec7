"""Fine-grained CPU emulation of the NMS mask rounds (k_nms.hip) under adversarial interleaving.

Every global-memory access of a lane is its own scheduling step, and the steps of all cells of a launch are interleaved
at random, so any ordering the GPU could produce between one lane's reads and another lane's acceptance writes is
reachable here.  The final survivor set is compared with the oracle's literal greedy NMS.  (Oracle = checker only.)
"""
import os, random, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import cref
from photogrammetry_amd import synth

RR, ND = 2, 5
NO = ND * ND
FLAG, KEYM = 0x80000000, 0x7FFFFFFF
M64 = (1 << 64) - 1


def build(raw, W, H, radius):
    gw, gh = (W + 7) // 8, (H + 7) // 8
    alive = {}; p0 = {}; p1 = {}; p2 = {}
    for x, y, s in zip(raw["x"], raw["y"], raw["fast_score"]):
        c = (int(x) >> 3, int(y) >> 3); b = ((int(y) & 7) << 3) | (int(x) & 7); code = int(s) - 11
        alive[c] = alive.get(c, 0) | (1 << b)
        if code in (5, 3): p0[c] = p0.get(c, 0) | (1 << b)
        if code in (3, 2): p1[c] = p1.get(c, 0) | (1 << b)
        if code in (5, 4): p2[c] = p2.get(c, 0) | (1 << b)
    disk = {}
    for dy in range(-RR, RR + 1):
        for dx in range(-RR, RR + 1):
            for pos in range(64):
                ux, uy = pos & 7, pos >> 3
                m = 0
                for q in range(64):
                    vx, vy = (q & 7) + 8 * dx, (q >> 3) + 8 * dy
                    if (vx - ux) ** 2 + (vy - uy) ** 2 <= radius * radius: m |= 1 << q
                disk[(dx, dy, pos)] = m
    return gw, gh, alive, p0, p1, p2, disk


def champ(a, q0, q1, q2, cx, cy):
    if not a: return 0
    for lvl, mask in ((5, q2 & q0), (4, q2 & ~q0), (3, ~q2 & q1 & q0), (2, ~q2 & q1 & ~q0), (1, ~q2 & ~q1)):
        m = a & mask & M64
        if m:
            b = (m & -m).bit_length() - 1  # lowest bit = raster-earliest inside the cell
            x, y = cx * 8 + (b & 7), cy * 8 + (b >> 3)
            return (lvl << 28) | (0x0FFFFFFF - ((y << 14) | x))
    return 0


def blockers(code, uy, dx, dy, an, dm, q0, q1, q2):
    rows_below = (1 << (8 * uy)) - 1; row_mine = 0xFF << (8 * uy)
    E = M64 if dy < 0 else (0 if dy > 0 else (rows_below | (row_mine if dx < 0 else 0)))
    ge = {5: q2 & q0, 4: q2, 3: q2 | (q1 & q0), 2: q2 | q1, 1: M64, 6: 0}
    return an & dm & (ge[code + 1] | (ge[code] & E)) & M64


def lane(c, S, accepted, r2):
    """generator: one round of cell c; yields before every global access."""
    cx, cy = c
    g = S["get"]
    yield; me = S["ent"].get(c, 0)
    yield; a = S["alive"].get(c, 0)
    q0, q1, q2 = S["p0"].get(c, 0), S["p1"].get(c, 0), S["p2"].get(c, 0)
    if me & FLAG:
        yield; S["ent"][c] = 0
        yield; S["alive"][c] = 0
        return
    if a == 0:
        if me: yield; S["ent"][c] = 0
        return
    pos = 0x0FFFFFFF - (me & 0x0FFFFFFF)
    b = (((pos >> 14) & 7) << 3) | (pos & 7)
    if me == 0 or not (a >> b) & 1:
        me = champ(a, q0, q1, q2, cx, cy)
        yield; S["ent"][c] = me
    mypos = 0x0FFFFFFF - (me & 0x0FFFFFFF)
    mx, my = mypos & 0x3FFF, mypos >> 14
    ux, uy, code = mx & 7, my & 7, me >> 28
    need = []
    for dy in range(-RR, RR + 1):
        for dx in range(-RR, RR + 1):
            if dx == 0 and dy == 0: continue
            yield; e = S["ent"].get((cx + dx, cy + dy), 0)
            ndx = dx * 8 - ux if dx > 0 else (ux - (dx * 8 + 7) if dx < 0 else 0)
            ndy = dy * 8 - uy if dy > 0 else (uy - (dy * 8 + 7) if dy < 0 else 0)
            if (e & KEYM) > me and ndx * ndx + ndy * ndy <= r2: need.append((dx, dy))
    if S["gap"]:
        for _ in range(S["gap"]): yield
    for dx, dy in need:
        n = (cx + dx, cy + dy)
        yield; an = S["alive"].get(n, 0)
        if blockers(code, uy, dx, dy, an, S["disk"][(dx, dy, uy * 8 + ux)], S["p0"].get(n, 0), S["p1"].get(n, 0), S["p2"].get(n, 0)):
            return
    # accept
    yield; S["ent"][c] = me | FLAG
    yield; S["alive"][c] = S["alive"].get(c, 0) & (1 << (uy * 8 + ux))
    for dy in range(-RR, RR + 1):
        for dx in range(-RR, RR + 1):
            if dx == 0 and dy == 0: continue
            d = S["disk"][(dx, dy, uy * 8 + ux)]
            n = (cx + dx, cy + dy)
            if d and n in S["alive"]:
                yield; S["alive"][n] &= ~d & M64
    accepted.append((mx, my, code + 11))


def run(raw, W, H, radius, seed, gap):
    gw, gh, alive, p0, p1, p2, disk = build(raw, W, H, radius)
    S = dict(alive=alive, p0=p0, p1=p1, p2=p2, disk=disk, ent={}, gap=gap, get=None)
    for c, a in alive.items(): S["ent"][c] = champ(a, p0.get(c, 0), p1.get(c, 0), p2.get(c, 0), c[0], c[1])
    rng = random.Random(seed)
    accepted = []
    cells = list(alive.keys())
    for launch in range(200):
        gens = [lane(c, S, accepted, radius * radius) for c in cells if S["ent"].get(c, 0) or S["alive"].get(c, 0)]
        if not gens: break
        while gens:
            i = rng.randrange(len(gens))
            try: next(gens[i])
            except StopIteration:
                gens[i] = gens[-1]; gens.pop()
    return accepted


if __name__ == "__main__":
    W, H, radius = int(os.environ.get("W", 640)), int(os.environ.get("H", 360)), int(os.environ.get("RADIUS", 16))
    frame = synth.make_frame(W, H, seed=4321, n_shapes=int(os.environ.get("SHAPES", 2500)))
    gray = cref.gray(frame)
    raw = cref.detect(gray, np.float32(0.1))
    kept = raw[cref.nms(raw, radius)]
    exp = sorted((int(x), int(y), int(s)) for x, y, s in zip(kept["x"], kept["y"], kept["fast_score"]))
    print(f"{len(raw)} raw points, {len(exp)} survivors expected")
    for gap in (0, 200):
        for seed in range(int(os.environ.get("SEEDS", 5))):
            got = sorted(run(raw, W, H, radius, seed, gap))
            print(f"gap {gap} seed {seed}: {'same' if got == exp else 'DIFFERENT: extra %s missing %s' % (sorted(set(got) - set(exp))[:5], sorted(set(exp) - set(got))[:5])}", flush=True)
