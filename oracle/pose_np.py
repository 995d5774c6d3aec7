"""numpy restatement of ImageProcessing/CameraPoseEstimation.cs -- TEST INFRASTRUCTURE ONLY (imported by tests/).

PARITY UNPINNED: the reference holds no test, fixture or output for this class, its only caller is commented out
(Photogrammetry/Program.cs:207-249), it draws subsets from an unseeded System.Random (:35) and takes singular vectors from
MathNet.Numerics 5.0.0 (absent from /root/reference).  What is restated literally: the subset-of-pairs RANSAC loop
(:26-94), the normalised 8-point system with its always-1 scale and column-major fill (:204-274), the signed inlier test
with the points in (Keypoint2, Keypoint1) order (:67-77), E = K^T F K with the hard-coded K, the four candidates, the
linear triangulation and the z >= 0 vote (:96-202).  Subsets come from the same seeded splitmix64 stream as the HIP
kernel; a null vector's sign is fixed by making its largest component positive (numpy's and Jacobi's vectors then agree).
"""
import numpy as np

MASK = (1 << 64) - 1
K = np.array([[1000, 0, 1500], [0, 1000, 2000], [0, 0, 1]], dtype=np.float32)          # CameraPoseEstimation.cs:98-99
KI = np.array([[0.001, 0, -1.5], [0, 0.001, -2.0], [0, 0, 1]], dtype=np.float32)


def _splitmix64(state):
    state = (state + 0x9E3779B97F4A7C15) & MASK
    z = state
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK
    return z ^ (z >> 31), state


def sample_indices(seed, m, s, P, n):
    """P distinct positions of an n-entry list for sample s of image pair m (the kernel's generator)."""
    st = (seed ^ ((m & 0xFFFFFFFF) << 32) ^ (((s & 0xFFFFFFFF) * 0xD1B54A32D192ED03) & MASK)) & MASK
    idx = []
    while len(idx) < P:
        v, st = _splitmix64(st)
        c = v % n
        if c not in idx:
            idx.append(int(c))
    return idx


def _fix_sign(v):
    return -v if v[np.argmax(np.abs(v))] < 0 else v


def estimate_fundamental(p1, p2):
    """EstimateFundamentalMatrix (:204-250).  p1, p2: integer pixel coordinates [P][2] of Keypoint1 / Keypoint2."""
    c1, c2 = p1.astype(np.float64).mean(0), p2.astype(np.float64).mean(0)          # CalculateCentroid (:276-288)
    t1, t2 = -c1.astype(np.float32), -c2.astype(np.float32)                        # scale = pow(2 / msd, 1 / 2) = pow(., 0) = 1
    x1, y1 = p1[:, 0].astype(np.float32) + t1[0], p1[:, 1].astype(np.float32) + t1[1]
    x2, y2 = p2[:, 0].astype(np.float32) + t2[0], p2[:, 1].astype(np.float32) + t2[1]
    A = np.stack([x1 * x2, x1 * y2, x1, y1 * x2, y1 * y2, y1, x2, y2, np.ones_like(x1)], 1).astype(np.float32)
    v = _fix_sign(np.linalg.svd(A.astype(np.float64))[2][-1])
    F0 = v.astype(np.float32).reshape(3, 3).T                                       # DenseOfColumnMajor(3, 3, lastRow)
    T1 = np.array([[1, 0, t1[0]], [0, 1, t1[1]], [0, 0, 1]], dtype=np.float32)
    T2 = np.array([[1, 0, t2[0]], [0, 1, t2[1]], [0, 0, 1]], dtype=np.float32)
    return (T2.T @ F0 @ T1).astype(np.float32)


def score(F, p1, p2, threshold):
    """(F * [x2, y2, 1]) . [x1, y1, 1] <= threshold over every keypoint pair (:53-77), float32."""
    h1 = np.concatenate([p1.astype(np.float32), np.ones((len(p1), 1), np.float32)], 1)
    h2 = np.concatenate([p2.astype(np.float32), np.ones((len(p2), 1), np.float32)], 1)
    res = ((h2 @ F.T.astype(np.float32)) * h1).sum(1, dtype=np.float32)
    return res <= np.float32(threshold)


def numerical_rank(F):
    s = np.linalg.svd(F.astype(np.float64), compute_uv=False)
    return int((s > s.max() * 1.1920929e-7 * 3.0).sum())


def ransac_fundamental(p1, p2, n_samples, P, threshold, seed, m=0, rank_check=False):
    """GetFundamentalMatrix (:26-94) -> (F, inlier count, best sample index) or (None, -1, -1)."""
    if P < 8:
        raise ValueError("At least 8 keypoint pairs must be included per sample")
    n = len(p1)
    if n < P:
        return None, -1, -1
    best = (None, 0, -1)
    for s in range(n_samples):
        idx = sample_indices(seed, m, s, P, n)
        F = estimate_fundamental(p1[idx], p2[idx])
        if rank_check and numerical_rank(F) != 2:
            continue
        c = int(score(F, p1, p2, threshold).sum())
        if c > best[1]:
            best = (F, c, s)
    return best if best[0] is not None else (None, -1, -1)


def pose_candidates(F):
    E = (K.T @ F.astype(np.float32) @ K).astype(np.float32)                         # :102
    V = np.linalg.svd(E.astype(np.float64))[2].T
    V = np.stack([_fix_sign(V[:, c]) for c in range(3)], 1)
    U = np.zeros((3, 3))
    for c in range(2):
        u = E.astype(np.float64) @ V[:, c]
        U[:, c] = u / np.linalg.norm(u)
    U[:, 2] = _fix_sign(np.cross(U[:, 0], U[:, 1]))
    Uf, VT = U.astype(np.float32), V.T.astype(np.float32)
    W = np.array([[0, -1, 0], [1, 0, 0], [0, 0, 1]], dtype=np.float32)
    R1, R2 = Uf @ W @ VT, Uf @ W.T @ VT                                             # :112-113
    s1 = 1.0 if np.linalg.det(R1.astype(np.float64)) > 0 else -1.0
    s2 = 1.0 if np.linalg.det(R2.astype(np.float64)) > 0 else -1.0
    u1 = Uf[:, 2]
    return [(u1 * s1, R1 * s1), (-u1 * s1, R1 * s1), (u1 * s2, R2 * s2), (-u1 * s2, R2 * s2)]   # :120-125


def triangulate(R, t, p1, p2):
    """Linear triangulation of every keypoint pair (:143-174) -> R X + t  [N][3]."""
    out = np.zeros((len(p1), 3), dtype=np.float32)
    P1 = np.concatenate([np.eye(3, dtype=np.float32), np.zeros((3, 1), np.float32)], 1)
    P2 = np.concatenate([R.astype(np.float32), t.astype(np.float32).reshape(3, 1)], 1)
    for e in range(len(p1)):
        n1 = KI @ np.array([p1[e, 0], p1[e, 1], 1], dtype=np.float32)
        n2 = KI @ np.array([p2[e, 0], p2[e, 1], 1], dtype=np.float32)
        D = np.stack([P1[0] - P1[2] * n1[0], P1[2] * n1[1] - P1[1], P2[0] - P2[2] * n2[0], P2[2] * n2[1] - P2[1]]).astype(np.float32)
        X = np.linalg.svd(D.astype(np.float64))[2][-1]
        sx = (X[:3] / X[3]).astype(np.float32)
        out[e] = R.astype(np.float32) @ sx + t.astype(np.float32)
    return out


def estimate_pose(F, p1, p2):
    """EstimateCameraPose (:96-202) -> (best index, R, t, votes[4], points of the winner)."""
    cands = pose_candidates(F)
    clouds = [triangulate(R, t, p1, p2) for t, R in cands]
    votes = [int((c[:, 2] >= 0).sum()) for c in clouds]
    b = int(np.argmax(votes))
    return b, cands[b][1], cands[b][0], votes, clouds[b]
