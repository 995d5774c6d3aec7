"""Independent numpy/Python twin of the C oracle -- TEST INFRASTRUCTURE ONLY.

Purpose: the reference's own tests pin only three FAST known-answers, so the C
restatement (pgx_oracle.c) is additionally cross-checked against this file, which
is written from the same C# sources but with DIFFERENT formulations:

  * FAST        -- a 16-bit "similar" mask per pixel and bit tricks for the circular run
                   (KeypointDetection.cs:65-138), vectorised over the whole image;
  * BRIEF       -- Python big integers standing in for System.Numerics.BigInteger
                   (Keypoint.cs:29-57);
  * NMS         -- the parallel "locally best undecided point" rounds, which
                   is the formulation the GPU uses (RedundantKeypointEliminator.cs:16-39);
  * matching    -- the parallel "locally dominant edge" rounds, again the GPU's
                   formulation, plus a pure-Python transcription of the C# loop for tiny cases
                   (KeypointMatching.cs:14-69).

Agreement of literal C, this twin and the HIP path on the same inputs is what the
"parity unpinned" stages rest on.
"""
import numpy as np

INT_MAX = 2**31 - 1

# KeypointDetection.cs:15-19 -- the last entry really is (-3, 1)
CIRCLE = [(-3, 0), (-3, 1), (-2, 2), (-1, 3), (0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1),
          (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, 1)]


def gray(rgba):
    """Grayscale.cs:19-23, float32 throughout."""
    r = rgba[..., 0].astype(np.float32)
    g = rgba[..., 1].astype(np.float32)
    b = rgba[..., 2].astype(np.float32)
    return ((r + b) + g) / np.float32(3 * 65535)


def apply_distortion(rgba, map_uv):
    """DeWarp.cs:19-37 as one gather; raises like the C# on mismatch / OOB."""
    if rgba.shape[:2] != map_uv.shape[:2]:
        raise ValueError("ArgumentException")
    H, W = rgba.shape[:2]
    u = map_uv[..., 0].astype(np.int64) & 0xFFFF
    v = map_uv[..., 1].astype(np.int64) & 0xFFFF
    if (u >= W).any() or (v >= H).any():
        raise IndexError("IndexOutOfRangeException")
    return rgba[v, u]


def fast_scores(img, T):
    """Score map: 0 = not a keypoint, else FastScore in 12..16 (KeypointDetection.cs:42-138)."""
    img = np.asarray(img, dtype=np.float32)
    H, W = img.shape
    out = np.zeros((H, W), dtype=np.int32)
    if H < 7 or W < 7:
        return out
    T = np.float32(T)
    c = img[3:H - 3, 3:W - 3]
    lo = (c - T).astype(np.float32)
    hi = (c + T).astype(np.float32)
    sim = np.zeros(c.shape, dtype=np.uint32)
    for idx, (dx, dy) in enumerate(CIRCLE):
        p = img[3 + dy:H - 3 + dy, 3 + dx:W - 3 + dx]
        sim |= ((p > lo) & (p < hi)).astype(np.uint32) << idx
    diff = (~sim) & 0xFFFF
    # pre-test (:116-133): at most one "similar" among ring entries 0, 4, 8, 12
    pre = np.zeros(c.shape, dtype=np.int32)
    for idx in (0, 4, 8, 12):
        pre += ((sim >> idx) & 1).astype(np.int32)
    # fifth "similar" rejects (:91-92)
    nsim = np.zeros(c.shape, dtype=np.int32)
    for idx in range(16):
        nsim += ((sim >> idx) & 1).astype(np.int32)
    # longest circular run of "different"
    dd = diff | (diff << 16)
    longest = np.zeros(c.shape, dtype=np.int32)
    run = dd.copy()
    for k in range(1, 17):
        longest = np.where(run != 0, k, longest)
        run = run & (dd >> k)
    longest = np.where(sim == 0, 16, np.minimum(longest, 16))
    ok = (pre <= 1) & (nsim <= 4) & (longest >= 12)
    out[3:H - 3, 3:W - 3] = np.where(ok, longest, 0)
    return out


def detect(img, T):
    """Raster-order (x, y, score) triples."""
    s = fast_scores(img, T)
    ys, xs = np.nonzero(s)  # row-major scan == raster order
    return np.stack([xs, ys, s[ys, xs]], axis=1).astype(np.int32)


def brief_bigint(img, x, y, pairs):
    """Keypoint.cs:29-57 with a Python int as the BigInteger."""
    H, W = img.shape
    d = 0
    for (ax, ay, bx, by) in pairs:
        d <<= 1
        x1, y1 = x + int(ax), y + int(ay)
        if not (0 <= x1 < W and 0 <= y1 < H):
            continue
        x2, y2 = x + int(bx), y + int(by)
        if not (0 <= x2 < W and 0 <= y2 < H):
            continue
        if img[y1, x1] < img[y2, x2]:
            d += 1
    return d


def bigint_to_words(d, P):
    words = (P + 31) // 32
    return np.array([(d >> (32 * w)) & 0xFFFFFFFF for w in range(words)], dtype=np.uint32)


def brief(img, xy, pairs):
    P = len(pairs)
    return np.stack([bigint_to_words(brief_bigint(img, int(x), int(y), pairs), P) for x, y in xy]) \
        if len(xy) else np.zeros((0, (P + 31) // 32), dtype=np.uint32)


def nms_rounds(xy, score, radius):
    """Parallel-rounds NMS.  Returns (order, rounds)."""
    n = len(score)
    if n == 0:
        return np.zeros(0, dtype=np.int32), 0
    xy = np.asarray(xy, dtype=np.int64)
    score = np.asarray(score, dtype=np.int64)
    key = (-score) * (n + 1) + np.arange(n)          # smaller = higher priority
    state = np.zeros(n, dtype=np.int8)               # 0 undecided, 1 accepted, 2 suppressed
    dx = xy[:, None, 0] - xy[None, :, 0]
    dy = xy[:, None, 1] - xy[None, :, 1]
    near = (dx * dx + dy * dy) <= (radius * radius if radius >= 0 else -1)
    np.fill_diagonal(near, False)
    rounds = 0
    while (state == 0).any():
        rounds += 1
        und = state == 0
        better = near & und[None, :] & (key[None, :] < key[:, None])
        newly = und & ~better.any(axis=1)
        state[newly] = 1
        sup = (state == 0) & (near & newly[None, :]).any(axis=1)
        state[sup] = 2
    acc = np.nonzero(state == 1)[0]
    return acc[np.argsort(key[acc], kind="stable")].astype(np.int32), rounds


def hamming_matrix(d1, d2):
    a = np.unpackbits(np.ascontiguousarray(d1).view(np.uint8), axis=1).astype(np.int32)
    b = np.unpackbits(np.ascontiguousarray(d2).view(np.uint8), axis=1).astype(np.int32)
    return a.sum(1)[:, None] + b.sum(1)[None, :] - 2 * (a @ b.T)


def match_rounds(d1, d2):
    """Locally-dominant-edge rounds.  Returns (pairs [n1][3], rounds)."""
    n1, n2 = len(d1), len(d2)
    if n1 == 0:
        return np.zeros((0, 3), dtype=np.int64), 0
    if n2 == 0:
        raise IndexError("ArgumentOutOfRangeException")
    D = hamming_matrix(d1, d2).astype(np.int64)
    BIG = 1 << 40
    av1 = np.ones(n1, dtype=bool)
    av2 = np.ones(n2, dtype=bool)
    acc = []
    rounds = 0
    while av1.any() and av2.any():
        rounds += 1
        M = np.where(av1[:, None] & av2[None, :], D, BIG)
        rbest = np.argmin(M * (n2 + 1) + np.arange(n2)[None, :], axis=1)   # min (d, k2) per row
        cbest = np.argmin(M * (n1 + 1) + np.arange(n1)[:, None], axis=0)   # min (d, k1) per column
        for i in np.nonzero(av1)[0]:
            j = rbest[i]
            if av2[j] and cbest[j] == i:
                acc.append((int(D[i, j]), int(i), int(j)))
        for (_, i, j) in acc:
            av1[i] = False
            av2[j] = False
    acc.sort()
    out = [(i, j, d) for (d, i, j) in acc]
    out += [(0, 0, INT_MAX)] * (n1 - len(out))
    return np.array(out, dtype=np.int64).reshape(-1, 3), rounds


def match_literal(d1, d2):
    """Transcription of the C# loop (KeypointMatching.cs:14-69); tiny cases only."""
    n1, n2 = len(d1), len(d2)
    D = hamming_matrix(d1, d2) if n1 and n2 else np.zeros((n1, n2), dtype=np.int64)
    avail1 = list(range(n1))
    avail2 = list(range(n2))
    out = []
    while len(out) < n1:
        smallest, s1, s2 = INT_MAX, 0, 0
        for k1 in avail1:
            for k2 in avail2:
                if smallest <= D[k1, k2]:
                    continue
                smallest, s1, s2 = int(D[k1, k2]), k1, k2
        if n2 == 0:
            raise IndexError("ArgumentOutOfRangeException")
        out.append((s1, s2, smallest))
        if s1 in avail1:
            avail1.remove(s1)
        if s2 in avail2:
            avail2.remove(s2)
    return np.array(out, dtype=np.int64).reshape(-1, 3)
