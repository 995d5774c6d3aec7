"""BASELINE configs[4] as ONE checked chain on the GPU: dewarp -> detect -> NMS -> BRIEF -> match -> RANSAC fundamental matrix
-> essential matrix / pose, every stage against its oracle ON THE DATA THE PREVIOUS STAGE PRODUCED.

  frames      8 x 640x360: a synthetic scene translated by (5 i, 2 i) px, the last frame a sparse one, so that image pairs
              (i, 7) have N1 > N2 and their lists end in (0, 0, int.MaxValue) entries
  detect      pgx_sequence_step_dev (detect + match in one C call, world size 1) vs cref.gray / detect / nms / brief: exact
  match       all 28 ordered pairs + (7, 0), (7, 3) vs cref.match_sorted on the GPU's own descriptors: exact
  RANSAC      pgx_fundamental_ransac_dev on the real match lists (the reference's commented call: 32 pairs per sample,
              threshold 0.001, Program.cs:229) vs oracle/pose_np.py with the same seed
  pose        pgx_pose_dev on the GPU's matrices vs pose_np.estimate_pose

The tail entries of a list ARE keypoint pairs of the reference -- (keypoints1[0], keypoints2[0], int.MaxValue), emitted by
KeypointMatching.cs:40-42,57-62 -- and GetFundamentalMatrix / EstimateCameraPose take the list as it comes
(CameraPoseEstimation.cs:42: `keypointPairs.OrderBy(...)`, :53: `foreach (var keypointPair in keypointPairs)`, :143): they can be
drawn into a sample, they are scored and they vote, like any other entry.  Both sides of this test do exactly that.

Pose arithmetic is parity-unpinned against the C# (unseeded RNG, MathNet SVD): the tolerances are those of
tests/test_gpu_pose.py, between two restatements of the same formulas."""
import numpy as np
import pytest
import torch

from oracle import cref, pose_np
from photogrammetry_amd import synth, dist as pdist
import photogrammetry_amd as pg

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
W, H, F, NKP, RADIUS, T = 640, 360, 8, 2048, 10, np.float32(0.1)
N_SAMPLES, PPS, THR, SEED = 64, 32, 0.001, 1234


def _frames():
    base = synth.make_frame(W, H, seed=4242, n_shapes=1400)
    fr = [synth.shift_frame(base, 5 * i, 2 * i) for i in range(F - 1)]
    fr.append(synth.shift_frame(synth.make_frame(W, H, seed=4243, n_shapes=150), 3, 1))   # the sparse one
    return np.stack(fr)


def _normed(Fm):
    Fm = np.asarray(Fm, dtype=np.float64).reshape(3, 3)
    return Fm / np.linalg.norm(Fm)


def test_configs4_chain_every_stage_against_its_oracle():
    eng = pg.Engine(0)
    try:
        pairs_tbl = pg.make_brief_pairs(9, 50, 256)
        dmap = pg.build_dewarp_map(W, H, [3e-4, 1e-7, 0, 0, 0])
        eng.set_brief_pairs(pairs_tbl)
        eng.set_detect_params(T, RADIUS)
        eng.set_capacity(1 << 17, NKP)
        eng.set_dewarp_map(dmap)
        frames = _frames()
        pl = pdist.all_pairs(F) + [(7, 0), (7, 3)]
        stream = torch.cuda.Stream(device=DEV)
        job = pdist.ShardedSequence(eng, W, H, F, pl, NKP, 8, DEV, stream=stream, comm="pgx")
        d_frames = torch.from_numpy(frames).to(DEV)
        torch.cuda.synchronize()
        job.step(d_frames)
        eng.check_status()
        counts = job.counts()

        # ---- detect: every frame against the oracle chain ------------------------------------------------------------
        kp_h = job.kp_l.cpu().numpy()
        xy = []
        for f in range(F):
            g = cref.gray(cref.apply_distortion(frames[f], dmap))
            raw = cref.detect(g, T)
            kept = raw[cref.nms(raw, RADIUS)][:NKP]
            edesc = cref.brief(g, np.stack([kept["x"], kept["y"]], 1), pairs_tbl)
            n = int(counts[f])
            assert n == len(kept) and int(job.nraw_l[f]) == len(raw), f
            assert (kp_h[f, :n, 0] == kept["x"]).all() and (kp_h[f, :n, 1] == kept["y"]).all() and (kp_h[f, :n, 2] == kept["fast_score"]).all()
            assert kp_h[f, :n, 3].view(np.float32).tobytes() == kept["value"].tobytes()
            assert (job.descriptors(f).cpu().numpy().view(np.uint32)[:n] == edesc).all()
            xy.append(np.stack([kept["x"], kept["y"]], 1).astype(np.int32))
        assert counts[7] < counts[:7].min() and counts[:7].min() > 300      # the sparse frame really is the smallest

        # ---- match: every list against the oracle, on the GPU's descriptors ----------------------------------------------
        lists = []
        for m, (a, b) in enumerate(pl):
            da = job.descriptors(a).cpu().numpy().view(np.uint32)[:counts[a]]
            db = job.descriptors(b).cpu().numpy().view(np.uint32)[:counts[b]]
            got = job.matches(m).cpu().numpy()[:counts[a]]
            exp = cref.match_sorted(da, db)
            assert (got[:, 0] == exp["k1"]).all() and (got[:, 1] == exp["k2"]).all() and (got[:, 2] == exp["dist"]).all(), (a, b)
            lists.append(got)
        tail = [int((l[:, 2] == cref.INT_MAX).sum()) for l in lists]
        m07 = pl.index((0, 7))
        assert tail[m07] == counts[0] - counts[7] > 0 and tail[pl.index((7, 0))] == 0
        assert (lists[m07][-tail[m07]:, :2] == 0).all()                  # (keypoints1[0], keypoints2[0], int.MaxValue)

        # ---- RANSAC fundamental matrix on the real lists ------------------------------------------------------------------
        M = len(pl)
        i32 = dict(dtype=torch.int32, device=DEV)
        d_pl = torch.tensor(pl, **i32)
        d_F = torch.zeros((M, 9), dtype=torch.float32, device=DEV)
        d_in, d_bs = torch.zeros(M, **i32), torch.zeros(M, **i32)
        with torch.cuda.stream(stream):
            eng.fundamental_ransac_dev(job.kp_l, job.out_all, job.counts_all, d_pl, M, NKP, N_SAMPLES, PPS, THR, d_F, d_in, d_bs, seed=SEED)
        eng.check_status()
        Fg, cnt, bs = d_F.cpu().numpy(), d_in.cpu().numpy(), d_bs.cpu().numpy()
        corr = []   # per image pair: the coordinates behind every list entry, tail entries included (they point at keypoint 0)
        for m, (a, b) in enumerate(pl):
            corr.append((xy[a][lists[m][:, 0]], xy[b][lists[m][:, 1]]))
        checked_F = 0
        for m in [0, 5, m07, pl.index((3, 7)), pl.index((7, 0)), pl.index((7, 3)), M - 3]:
            p1, p2 = corr[m]
            n = len(p1)
            Fo, co, so = pose_np.ransac_fundamental(p1, p2, N_SAMPLES, PPS, THR, SEED, m=m)
            tol = max(3, n // 100)
            # the maximum over all samples: comparable only when the samples have clear null vectors.  A list that is mostly
            # tail entries (680 of 865 here) fills a sample with copies of ONE point pair: its 8-point system is rank
            # deficient, the null space has several dimensions and the two solvers legitimately pick different vectors.
            if tail[m] * 20 < n:
                assert abs(int(cnt[m]) - co) <= tol, (m, cnt[m], co)
            # the GPU's winner: that sample's matrix by the oracle, and that many inliers when the oracle scores the GPU's matrix
            idx = pose_np.sample_indices(SEED, m, int(bs[m]), PPS, n)
            if tail[m]:
                assert max(idx) < n       # positions of the whole list: tail entries can be drawn
            A = _system(p1[idx], p2[idx])
            sv = np.linalg.svd(A, compute_uv=False)
            if sv[-2] > 50 * sv[-1] and sv[-2] > 1e-6 * sv[0]:     # a clear null vector: the two solvers must agree on it
                Fs = pose_np.estimate_fundamental(p1[idx], p2[idx])
                assert np.abs(_normed(Fg[m]) - _normed(Fs)).max() < 2e-3, m
                checked_F += 1
            assert abs(int(pose_np.score(Fg[m].reshape(3, 3), p1, p2, THR).sum()) - int(cnt[m])) <= tol, m
        # single samples on the real lists (n_samples = 1: sample 0 of every image pair is the winner by definition), so that
        # the matrix comparison does not depend on WHICH sample wins
        d_F1 = torch.zeros((M, 9), dtype=torch.float32, device=DEV)
        for seed in range(5):
            with torch.cuda.stream(stream):
                eng.fundamental_ransac_dev(job.kp_l, job.out_all, job.counts_all, d_pl, M, NKP, 1, PPS, THR, d_F1, d_in, d_bs, seed=seed)
            eng.check_status()
            F1, c1 = d_F1.cpu().numpy(), d_in.cpu().numpy()
            for m in [0, 5, 12, pl.index((7, 0)), pl.index((7, 3))]:
                p1, p2 = corr[m]
                idx = pose_np.sample_indices(seed, m, 0, PPS, len(p1))
                sv = np.linalg.svd(_system(p1[idx], p2[idx]), compute_uv=False)
                if sv[-2] > 50 * sv[-1] and sv[-2] > 1e-6 * sv[0]:
                    Fs = pose_np.estimate_fundamental(p1[idx], p2[idx])
                    co1 = int(pose_np.score(Fs, p1, p2, THR).sum())
                    if int(c1[m]) < 0:      # no sample with an inlier: bestF stays null (CameraPoseEstimation.cs:79-89), -1 here
                        assert co1 <= max(3, len(p1) // 100), (seed, m, co1)
                        continue
                    assert np.abs(_normed(F1[m]) - _normed(Fs)).max() < 2e-3, (seed, m)
                    assert abs(co1 - int(c1[m])) <= max(3, len(p1) // 100), (seed, m)
                    checked_F += 1
        assert checked_F >= 8

        # ---- pose from the GPU's matrices -----------------------------------------------------------------------------------
        d_Rt = torch.zeros((M, 12), dtype=torch.float32, device=DEV)
        d_votes, d_best = torch.zeros((M, 4), **i32), torch.zeros(M, **i32)
        with torch.cuda.stream(stream):
            eng.pose_dev(job.kp_l, job.out_all, job.counts_all, d_pl, M, NKP, d_F, d_Rt, d_votes, d_best)
        eng.check_status()
        votes, best, Rt = d_votes.cpu().numpy(), d_best.cpu().numpy(), d_Rt.cpu().numpy()
        checked_pose = 0
        for m in [0, 5, m07, pl.index((7, 0)), M - 3]:
            p1, p2 = corr[m]
            n = len(p1)
            assert (votes[m] >= 0).all() and votes[m].max() <= n and int(best[m]) == int(np.argmax(votes[m]))   # first maximum, every entry votes
            E = (pose_np.K.T @ Fg[m].reshape(3, 3) @ pose_np.K).astype(np.float64)
            s = np.linalg.svd(E, compute_uv=False)
            if s[1] > 1e-3 * s[0] and s[0] - s[1] > 1e-2 * s[0] and s[1] - s[2] > 1e-2 * s[0]:   # distinct singular values: unique vectors
                b, R, t, vo, _ = pose_np.estimate_pose(Fg[m].reshape(3, 3), p1, p2)
                assert np.abs(votes[m] - np.array(vo)).max() <= max(3, n // 50), (m, votes[m], vo)
                if sorted(vo)[-1] - sorted(vo)[-2] > max(3, n // 50):          # a clear winner
                    assert int(best[m]) == b
                    assert np.abs(Rt[m][:9].reshape(3, 3) - R).max() < 2e-3 and np.abs(Rt[m][9:] - t).max() < 2e-3
                checked_pose += 1
        assert checked_pose >= 2
    finally:
        eng.set_stream(0)
        eng.close()


def _system(p1, p2):
    """The 8-point system of pose_np.estimate_fundamental (centred coordinates), for its conditioning only."""
    c1, c2 = p1.astype(np.float64).mean(0), p2.astype(np.float64).mean(0)
    x1, y1 = p1[:, 0] - c1[0], p1[:, 1] - c1[1]
    x2, y2 = p2[:, 0] - c2[0], p2[:, 1] - c2[1]
    return np.stack([x1 * x2, x1 * y2, x1, y1 * x2, y1 * y2, y1, x2, y2, np.ones_like(x1)], 1)
