"""RANSAC fundamental matrix + camera pose on the GPU against the numpy restatement (oracle/pose_np.py).

PARITY UNPINNED against the reference (unseeded RNG, MathNet SVD, no fixture: see the oracle's header).  The tolerance is
therefore between two restatements of the same formulas: float32 matrices equal to 2e-3 relative (after the shared sign
rule), inlier counts and depth votes equal up to the handful of pairs whose residual sits within rounding of the threshold."""
import numpy as np
import pytest
import torch

from oracle import pose_np
import photogrammetry_amd as pg

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def engine():
    e = pg.Engine(0)
    yield e
    e.close()


def _two_views(n, n_out, seed):
    """n true correspondences of a synthetic scene seen by two cameras with the reference's K, plus n_out wrong ones."""
    rng = np.random.default_rng(seed)
    K = pose_np.K.astype(np.float64)
    X = np.stack([rng.uniform(-3, 3, n), rng.uniform(-4, 4, n), rng.uniform(4, 9, n)], 1)
    a = 0.07
    R = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
    t = np.array([0.6, 0.05, 0.1])
    x1 = (K @ X.T).T
    x2 = (K @ (R @ X.T + t[:, None])).T
    p1 = np.rint(x1[:, :2] / x1[:, 2:3]).astype(np.int32)
    p2 = np.rint(x2[:, :2] / x2[:, 2:3]).astype(np.int32)
    ok = (p1 >= 0).all(1) & (p2 >= 0).all(1) & (p1[:, 0] < 3000) & (p2[:, 0] < 3000) & (p1[:, 1] < 4000) & (p2[:, 1] < 4000)
    p1, p2 = p1[ok], p2[ok]
    o1 = np.stack([rng.integers(0, 3000, n_out), rng.integers(0, 4000, n_out)], 1).astype(np.int32)
    o2 = np.stack([rng.integers(0, 3000, n_out), rng.integers(0, 4000, n_out)], 1).astype(np.int32)
    return np.concatenate([p1, o1]), np.concatenate([p2, o2])


def _upload(sets, stride):
    """Image pair m = frames (2m, 2m+1); keypoint k of both frames is correspondence k, the match list is (k, k, 0)."""
    M = len(sets)
    kp = np.zeros((2 * M, stride), dtype=pg.KEYPOINT_DTYPE)
    ml = np.zeros((M, stride, 3), dtype=np.int32)
    counts = np.zeros(2 * M, dtype=np.int32)
    for m, (p1, p2) in enumerate(sets):
        n = len(p1)
        perm = np.random.default_rng(m).permutation(n)          # k2 != k1: the indirection through the list is exercised
        kp["x"][2 * m, :n], kp["y"][2 * m, :n] = p1[:, 0], p1[:, 1]
        kp["x"][2 * m + 1, perm], kp["y"][2 * m + 1, perm] = p2[:, 0], p2[:, 1]   # field first: fancy indexing copies
        ml[m, :n, 0], ml[m, :n, 1] = np.arange(n), perm
        counts[2 * m] = counts[2 * m + 1] = n
    t = lambda a: torch.from_numpy(a).to(DEV)
    return (t(kp.view(np.int32).reshape(2 * M, stride, 4)), t(ml), t(counts),
            torch.tensor([[2 * m, 2 * m + 1] for m in range(M)], dtype=torch.int32, device=DEV))


def _normed(F):
    F = np.asarray(F, dtype=np.float64).reshape(3, 3)
    return F / np.linalg.norm(F)


def test_single_samples_match_the_oracle(engine):
    sets = [_two_views(300, 60, 1), _two_views(500, 0, 2), _two_views(40, 200, 3)]
    stride = 1024
    d_kp, d_ml, d_counts, d_pl = _upload(sets, stride)
    M = len(sets)
    d_F = torch.zeros((M, 9), dtype=torch.float32, device=DEV)
    d_in = torch.zeros(M, dtype=torch.int32, device=DEV)
    d_bs = torch.zeros(M, dtype=torch.int32, device=DEV)
    for P in (8, 12):
        for seed in range(6):
            torch.cuda.synchronize()
            engine.fundamental_ransac_dev(d_kp, d_ml, d_counts, d_pl, M, stride, 1, P, 0.001, d_F, d_in, d_bs, seed=seed)
            engine.check_status()
            F, cnt = d_F.cpu().numpy(), d_in.cpu().numpy()
            for m, (p1, p2) in enumerate(sets):
                idx = pose_np.sample_indices(seed, m, 0, P, len(p1))
                Fo = pose_np.estimate_fundamental(p1[idx], p2[idx])
                assert np.abs(_normed(F[m]) - _normed(Fo)).max() < 2e-3, (P, seed, m)
                co = int(pose_np.score(Fo, p1, p2, 0.001).sum())
                assert abs(int(cnt[m]) - co) <= max(3, co // 100), (P, seed, m, cnt[m], co)


def test_ransac_picks_the_first_best_sample_and_argument_errors(engine):
    sets = [_two_views(400, 100, 11), _two_views(200, 300, 12)]
    stride = 1024
    d_kp, d_ml, d_counts, d_pl = _upload(sets, stride)
    M, S, P = len(sets), 96, 8
    d_F = torch.zeros((M, 9), dtype=torch.float32, device=DEV)
    d_in = torch.zeros(M, dtype=torch.int32, device=DEV)
    d_bs = torch.zeros(M, dtype=torch.int32, device=DEV)
    torch.cuda.synchronize()
    engine.fundamental_ransac_dev(d_kp, d_ml, d_counts, d_pl, M, stride, S, P, 0.001, d_F, d_in, d_bs, seed=77)
    engine.check_status()
    for m, (p1, p2) in enumerate(sets):
        Fo, co, so = pose_np.ransac_fundamental(p1, p2, S, P, 0.001, 77, m=m)
        got_c, got_s = int(d_in[m]), int(d_bs[m])
        assert abs(got_c - co) <= max(3, co // 100)
        # the GPU's winner, re-scored by the oracle, really has that many inliers, and it is that sample's matrix
        idx = pose_np.sample_indices(77, m, got_s, P, len(p1))
        Fs = pose_np.estimate_fundamental(p1[idx], p2[idx])
        assert np.abs(_normed(d_F[m].cpu().numpy()) - _normed(Fs)).max() < 2e-3
        assert abs(int(pose_np.score(Fs, p1, p2, 0.001).sum()) - got_c) <= max(3, co // 100)
    with pytest.raises(pg.ArgumentException):          # CameraPoseEstimation.cs:28-29
        engine.fundamental_ransac_dev(d_kp, d_ml, d_counts, d_pl, M, stride, S, 7, 0.001, d_F, d_in, d_bs)
    # a list shorter than the subset (:31-32): reported per image pair as -1
    short = [(sets[0][0][:5], sets[0][1][:5])]
    k2, m2, c2, pl2 = _upload(short, 64)
    torch.cuda.synchronize()
    engine.fundamental_ransac_dev(k2, m2, c2, pl2, 1, 64, 4, 8, 0.001, d_F, d_in, d_bs)
    engine.check_status()
    assert int(d_in[0]) == -1 and int(d_bs[0]) == -1


def test_pose_matches_the_oracle(engine):
    sets = [_two_views(250, 0, 21), _two_views(300, 40, 22)]
    stride = 512
    d_kp, d_ml, d_counts, d_pl = _upload(sets, stride)
    M = len(sets)
    Fs = []
    for p1, p2 in sets:   # a matrix from all true correspondences (normalised 8-point on many pairs)
        Fs.append(pose_np.estimate_fundamental(p1[:200], p2[:200]))
    d_F = torch.from_numpy(np.stack(Fs).reshape(M, 9)).to(DEV)
    d_Rt = torch.zeros((M, 12), dtype=torch.float32, device=DEV)
    d_votes = torch.zeros((M, 4), dtype=torch.int32, device=DEV)
    d_best = torch.zeros(M, dtype=torch.int32, device=DEV)
    d_pts = torch.zeros((M, stride, 3), dtype=torch.float32, device=DEV)
    torch.cuda.synchronize()
    engine.pose_dev(d_kp, d_ml, d_counts, d_pl, M, stride, d_F, d_Rt, d_votes, d_best, d_pts)
    engine.check_status()
    for m, (p1, p2) in enumerate(sets):
        b, R, t, votes, cloud = pose_np.estimate_pose(Fs[m], p1, p2)
        gv = d_votes[m].cpu().numpy()
        assert np.abs(gv - np.array(votes)).max() <= max(3, len(p1) // 50), (gv, votes)
        assert int(d_best[m]) == b
        Rt = d_Rt[m].cpu().numpy()
        assert np.abs(Rt[:9].reshape(3, 3) - R).max() < 2e-3 and np.abs(Rt[9:] - t).max() < 2e-3
        got = d_pts[m, :len(p1)].cpu().numpy()
        good = np.abs(cloud).max(1) < 1e3                    # ill-conditioned points (X[3] ~ 0) are compared by sign only
        assert np.abs(got[good] - cloud[good]).max() < 5e-2 * max(1.0, np.abs(cloud[good]).max())
