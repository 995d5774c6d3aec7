"""GPU tests of the device track graph (pgx_tracks_dev) against the sequential oracle (oracle/tracks_np.py).

PARITY UNPINNED BY CONSTRUCTION: the reference has no multi-frame structure (TestService.cs:80-96 handles one
image pair), so there is no reference output.  What is checked is that the parallel build (lock-free union-find, per-frame
hash tables, scan, rank sort) reproduces the order-independent semantics of include/pgx.h bit for bit: the same tracks in
the same order, the same per-node track ids, the same dropped components -- on hand-built cases (conflict, int.MaxValue
tail entries, empty frames), on random lists, through slot-permuted buffers with padding (the rank-major layout of the
gathered buffers), and at the bench job's full size through size-independent properties (known tracks by construction,
invariance under the order of the pairs, idempotence).  The host form (pgx_tracks_*) must agree with both.
"""
import numpy as np
import pytest
import torch

from oracle import tracks_np
import photogrammetry_amd as pg

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
INT_MAX = 2**31 - 1


@pytest.fixture(scope="module")
def engine():
    e = pg.Engine(0)
    yield e
    e.close()


def run_dev(engine, counts, pair_list, matches, stride, max_dist, min_len=2, frame_ids=None, n_frames=None):
    """counts [F] by slot, pair_list [M][2] slots, matches [M][stride][3] -> (offsets, nodes, track_of, summary dict)."""
    i32 = dict(dtype=torch.int32, device=DEV)
    F = len(counts)
    nf = F if n_frames is None else n_frames
    M = len(pair_list)
    d_m = torch.from_numpy(np.ascontiguousarray(matches, dtype=np.int32).reshape(max(M, 1), stride, 3)).to(DEV)
    d_c = torch.tensor(np.asarray(counts, dtype=np.int32), **i32)
    d_pl = torch.tensor(np.asarray(pair_list, dtype=np.int32).reshape(-1, 2) if M else np.zeros((1, 2), np.int32), **i32)
    d_ids = None if frame_ids is None else torch.tensor(np.asarray(frame_ids, dtype=np.int32), **i32)
    track_of = torch.full((nf, stride), 77, **i32)
    offsets = torch.full((nf * stride + 1,), 77, **i32)
    nodes = torch.full((nf * stride, 2), 77, **i32)
    summary = torch.full((8,), 77, **i32)
    torch.cuda.synchronize()
    engine.tracks_dev(d_m, d_c, d_pl, M, F, stride, nf, max_dist, min_len, track_of, offsets, nodes, summary, d_frame_ids=d_ids)
    engine.check_status()
    s = summary.cpu().tolist()
    summ = {"n_tracks": s[0], "n_nodes": s[1], "dropped": s[2], "dropped_nodes": s[3], "edges": s[4], "longest": s[5],
            "largest_dropped": s[6]}
    assert s[7] == 0
    return offsets.cpu().numpy()[:s[0] + 1], nodes.cpu().numpy()[:s[1]], track_of.cpu().numpy(), summ


def as_lists(offsets, nodes):
    return [[(int(f), int(k)) for f, k in nodes[offsets[t]:offsets[t + 1]]] for t in range(len(offsets) - 1)]


def check_against_oracle(engine, counts, pl, m, stride, max_dist, min_len=2):
    exp, exp_tof, exp_s = tracks_np.tracks(counts, pl, m, max_dist, min_len)
    off, nodes, tof, summ = run_dev(engine, counts, pl, m, stride, max_dist, min_len)
    assert as_lists(off, nodes) == exp
    assert summ == exp_s
    w = exp_tof.shape[1]
    assert (tof[:, :w] == exp_tof).all() and (tof[:, w:] == -1).all()
    host, dropped, dropped_nodes = pg.tracks_host(counts, pl, [m[p] for p in range(len(pl))], max_dist, min_len)
    assert host == exp and (dropped, dropped_nodes) == (exp_s["dropped"], exp_s["dropped_nodes"])
    return exp, summ


def test_hand_built_conflict_tail_and_empty_frames(engine):
    """Frame 3 is empty, frame 4 holds one keypoint nobody links; the int.MaxValue tail entry never links even with
    max_dist = int.MaxValue; one edge closes a component over two keypoints of frame 0 -> dropped as a whole."""
    stride = 4
    counts = [2, 2, 2, 0, 1]
    pl = [(0, 1), (1, 2), (0, 2), (3, 4), (4, 0)]
    m = np.zeros((len(pl), stride, 3), dtype=np.int32)
    m[..., 2] = INT_MAX
    m[0, :2] = [[0, 0, 5], [1, 1, 99]]
    m[1, :2] = [[0, 1, 3], [0, 0, INT_MAX]]
    m[2, :2] = [[1, 1, 2], [0, 0, 50]]
    m[4, :1] = [[0, 0, 70]]
    # stale rows beyond a list's length must not count: a "perfect" edge in entry 3 of pair (0, 1), whose list has 2 entries
    m[0, 3] = [1, 1, 0]
    exp, summ = check_against_oracle(engine, counts, pl, m, stride, 10)
    assert exp == [] and summ["dropped"] == 1 and summ["dropped_nodes"] == 4
    exp, summ = check_against_oracle(engine, counts, pl, m, stride, INT_MAX)     # everything but the tail links
    assert summ["edges"] == 6 and summ["dropped"] == 1
    m[2, 0] = [1, 1, 20]
    exp, summ = check_against_oracle(engine, counts, pl, m, stride, 10)
    assert exp == [[(0, 0), (1, 0), (2, 1)]] and summ["dropped"] == 0 and summ["longest"] == 3
    exp, _ = check_against_oracle(engine, counts, pl, m, stride, 10, min_len=1)   # singletons are tracks at min_len 1
    assert len(exp) == 1 + 4 and [(4, 0)] in exp
    exp, _ = check_against_oracle(engine, counts, pl, m, stride, 10, min_len=4)
    assert exp == []


def _random_case(seed, F, stride, dmax=60):
    rng = np.random.default_rng(seed)
    counts = rng.integers(0, stride + 1, F).astype(np.int32)
    counts[rng.integers(0, F)] = 0
    pl = [(a, b) for a in range(F) for b in range(F) if a != b and rng.random() < 0.6]
    m = np.zeros((len(pl), stride, 3), dtype=np.int32)
    m[..., 0] = rng.integers(0, stride, m.shape[:2])
    m[..., 1] = rng.integers(0, stride, m.shape[:2])
    m[..., 2] = rng.integers(0, dmax, m.shape[:2])
    m[rng.random(m.shape[:2]) < 0.1] = [0, 0, INT_MAX]
    return counts, pl, m


@pytest.mark.parametrize("seed", range(6))
def test_random_lists_equal_oracle(engine, seed):
    """Random lists: most components are inconsistent at a loose gate, few at a tight one; both sides of the drop rule."""
    F, stride = 5 + seed, [7, 64, 300, 257, 33, 1024][seed]
    counts, pl, m = _random_case(seed, F, stride)
    for max_dist, min_len in ((0, 2), (1, 2), (4, 1), (30, 3)):
        check_against_oracle(engine, counts, pl, m, stride, max_dist, min_len)


def test_slot_permutation_padding_and_frame_subset(engine):
    """The gathered buffers are rank-major: frame f sits in slot (f mod G) * fs + f / G, some slots are padding, and a rank may
    build the graph for a subset of the frames.  d_frame_ids maps slots to frame numbers; -1 slots and every image pair
    touching one are skipped."""
    F, stride = 7, 96
    counts, pl, m = _random_case(42, F, stride, dmax=40)
    exp, exp_tof, exp_s = tracks_np.tracks(counts, pl, m, 3, 2)
    assert exp_s["n_tracks"] > 3
    # 3 "ranks", 3 slots each: slot of frame f = (f % 3) * 3 + f // 3; slots 5 and 8 are padding
    G, fs = 3, 3
    slot = [(f % G) * fs + f // G for f in range(F)]
    ids = -np.ones(G * fs, dtype=np.int32)
    c_s = np.zeros(G * fs, dtype=np.int32)
    for f in range(F):
        ids[slot[f]] = f
        c_s[slot[f]] = counts[f]
    c_s[ids < 0] = 50                                # garbage counts in padding slots must not matter
    pl_s = [(slot[a], slot[b]) for a, b in pl] + [(8, 0), (-1, -1)]   # a pair naming a padding slot, and a padding row
    m_s = np.concatenate([m, np.zeros((2, stride, 3), dtype=np.int32)])
    off, nodes, tof, summ = run_dev(engine, c_s, pl_s, m_s, stride, 3, 2, frame_ids=ids, n_frames=F)
    assert as_lists(off, nodes) == exp and summ == exp_s
    # a subset: frames {1, 2, 4, 6} numbered 0..3; pairs touching other frames drop out
    sub = [1, 2, 4, 6]
    ids2 = -np.ones(G * fs, dtype=np.int32)
    for i, f in enumerate(sub):
        ids2[slot[f]] = i
    keep = [p for p, (a, b) in enumerate(pl) if a in sub and b in sub]
    exp2, _, exp2_s = tracks_np.tracks(counts[sub], [(sub.index(pl[p][0]), sub.index(pl[p][1])) for p in keep], m[keep], 3, 2)
    off, nodes, tof, summ = run_dev(engine, c_s, pl_s, m_s, stride, 3, 2, frame_ids=ids2, n_frames=len(sub))
    assert as_lists(off, nodes) == exp2 and summ == exp2_s


def _constructed_job(F, K, seed, junk=0.3):
    """F frames x K keypoints with known tracks: ground-truth point g sits at keypoint perm_f[g] of frame f and is visible in
    a frame with probability 0.8; every ordered pair (a < b) lists true correspondences with a small distance and fills the
    other entries with junk matches at distances >= 90 (what the greedy matcher's forced assignments look like)."""
    rng = np.random.default_rng(seed)
    perm = np.stack([rng.permutation(K) for _ in range(F)])          # perm[f][g] = keypoint index of point g in frame f
    inv = np.argsort(perm, axis=1)                                     # inv[f][k] = point at keypoint k
    vis = rng.random((F, K)) < 0.8                                     # vis[f][g]
    pl = [(a, b) for a in range(F) for b in range(a + 1, F)]
    m = np.zeros((len(pl), K, 3), dtype=np.int32)
    for p, (a, b) in enumerate(pl):
        g = inv[a]                                                     # point of each keypoint of frame a
        true = vis[a][g] & vis[b][g]
        m[p, :, 0] = np.arange(K)
        m[p, :, 1] = np.where(true, perm[b][g], rng.integers(0, K, K))
        m[p, :, 2] = np.where(true, rng.integers(0, 20, K), rng.integers(90, 140, K))
        order = rng.permutation(K)                                     # any list order
        m[p] = m[p][order]
    counts = np.full(F, K, dtype=np.int32)
    return counts, pl, m, perm, vis


def test_bench_size_known_tracks_and_invariances(engine):
    """64 frames x 4096 keypoints, all 2016 pairs = 8.3 M match entries (the bench job's size).  By construction the tracks at
    max_dist = 64 are exactly the ground-truth points seen in >= 2 frames, one node per frame where visible; the result is
    the same for any order of the image pairs and when the call is repeated on the same buffers; the vectorised oracle
    (scipy connected components) agrees array for array."""
    F, K = 64, 4096
    counts, pl, m, perm, vis = _constructed_job(F, K, 7)
    off, nodes, tof, summ = run_dev(engine, counts, pl, m, K, 64, 2)
    nvis = vis.sum(0)
    assert summ["n_tracks"] == int((nvis >= 2).sum()) and summ["n_nodes"] == int(nvis[nvis >= 2].sum())
    assert summ["dropped"] == 0 and summ["longest"] == int(nvis.max())
    # every track is one ground-truth point: the point of its first node, in every frame where it is visible
    inv = np.argsort(perm, axis=1)
    t_pt = inv[nodes[off[:-1], 0], nodes[off[:-1], 1]]
    pt_of_node = inv[nodes[:, 0], nodes[:, 1]]
    assert (pt_of_node == np.repeat(t_pt, np.diff(off))).all()
    assert (np.diff(off) == nvis[t_pt]).all()
    # order: tracks by first node, nodes ascending
    nid = nodes[:, 0].astype(np.int64) * K + nodes[:, 1]
    firsts = nid[off[:-1]]
    assert (np.diff(firsts) > 0).all()
    inner = np.ones(len(nid), dtype=bool)
    inner[off[:-1]] = False
    assert (np.diff(nid)[inner[1:]] > 0).all()
    # the vectorised oracle, array for array
    e_off, e_nodes, e_tof, e_s = tracks_np.tracks_arrays(counts, pl, m, K, 64, 2)
    assert (off == e_off).all() and (nodes == e_nodes).all() and (tof == e_tof).all() and summ == e_s
    # any order of the image pairs
    rng = np.random.default_rng(1)
    o = rng.permutation(len(pl))
    off2, nodes2, tof2, summ2 = run_dev(engine, counts, [pl[i] for i in o], m[o], K, 64, 2)
    assert (off2 == off).all() and (nodes2 == nodes).all() and (tof2 == tof).all() and summ2 == summ
    # a loose gate lets the junk matches in: one giant inconsistent component, dropped; array for array again
    off3, nodes3, tof3, summ3 = run_dev(engine, counts, pl, m, K, 200, 2)
    e_off, e_nodes, e_tof, e_s = tracks_np.tracks_arrays(counts, pl, m, K, 200, 2)
    assert (off3 == e_off).all() and (nodes3 == e_nodes).all() and (tof3 == e_tof).all() and summ3 == e_s
    assert summ3["dropped"] >= 1 and summ3["largest_dropped"] > 1000


def test_bad_arguments(engine):
    i32 = dict(dtype=torch.int32, device=DEV)
    t = torch.zeros(64, **i32)
    with pytest.raises(pg.ArgumentException):
        engine.tracks_dev(t, t, t, 1, 2, 4, 3, 10, 2, t, t, t, t)          # n_frames != F without frame ids
    with pytest.raises(pg.ArgumentException):
        engine.tracks_dev(t, t, t, 1, 2, 0, 2, 10, 2, t, t, t, t)          # stride 0
    with pytest.raises(pg.ArgumentException):
        engine.tracks_dev(t, t, t, 1, 1 << 12, 1 << 20, 1 << 12, 10, 2, t, t, t, t)   # more than 2^30 nodes
