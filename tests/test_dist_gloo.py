"""CPU tests of the N > 1 path: world_size-2 `gloo` process groups exercising the sharding and the
two all-gathers of photogrammetry_amd/dist.py.  The GPU compute of each rank is stood in for by the
CPU oracle (allowed here: tests may use the oracle as the checker / stand-in), so what is verified is
that a pair-sharded run reproduces the single-process result exactly, including the track graph."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import cref, tracks_np
from photogrammetry_amd import dist as pdist
from photogrammetry_amd import synth

W, H, CAP, RADIUS = 200, 140, 256, 9
T = np.float32(0.1)
N_FRAMES = 5


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _frames():
    base = synth.make_frame(W, H, seed=3, n_shapes=120)
    return [synth.shift_frame(base, 3 * i, i) for i in range(N_FRAMES)]


def _detect(frame, pairs):
    g = cref.gray(frame)
    raw = cref.detect(g, T)
    kept = raw[cref.nms(raw, RADIUS)][:CAP]
    desc = np.zeros((CAP, 8), dtype=np.uint32)
    desc[:len(kept)] = cref.brief(g, np.stack([kept["x"], kept["y"]], 1), pairs)
    return desc.view(np.int32), len(kept)


def _match(desc_a, n_a, desc_b, n_b):
    out = np.zeros((CAP, 3), dtype=np.int32)
    out[:, 2] = cref.INT_MAX
    if n_a:
        m = cref.match(desc_a[:n_a].view(np.uint32), desc_b[:n_b].view(np.uint32))
        out[:n_a] = np.stack([m["k1"], m["k2"], m["dist"]], 1)
    return out


def _single_process():
    pairs = cref.gaussian_pairs(0, 20, 256)
    fr = _frames()
    det = [_detect(f, pairs) for f in fr]
    desc = np.stack([d for d, _ in det])
    counts = np.array([n for _, n in det], dtype=np.int32)
    pl = pdist.all_pairs(N_FRAMES)
    matches = np.stack([_match(desc[a], counts[a], desc[b], counts[b]) for a, b in pl])
    return desc, counts, pl, matches


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pairs = cref.gaussian_pairs(0, 20, 256)
        fr = _frames()
        # phase 1: detect the frames this rank owns
        mine = pdist.local_items(N_FRAMES, rank, world)
        ns = pdist.slots(N_FRAMES, world)
        desc_l = torch.zeros((ns, CAP, 8), dtype=torch.int32)
        cnt_l = torch.zeros(ns, dtype=torch.int32)
        for k, f in enumerate(mine):
            d, n = _detect(fr[f], pairs)
            desc_l[k] = torch.from_numpy(d)
            cnt_l[k] = n
        # phase 2: one all-gather of the per-frame records
        desc_all, cnt_all = pdist.exchange_descriptors(desc_l, cnt_l, N_FRAMES)
        # phase 3: match the image pairs this rank owns
        pl = pdist.all_pairs(N_FRAMES)
        mine_p = pdist.local_items(len(pl), rank, world)
        out_l = torch.zeros((pdist.slots(len(pl), world), CAP, 3), dtype=torch.int32)
        da, ca = desc_all.numpy(), cnt_all.numpy()
        for k, p in enumerate(mine_p):
            a, b = pl[p]
            out_l[k] = torch.from_numpy(_match(da[a], ca[a], da[b], ca[b]))
        # phase 4: one all-gather of the match lists
        out_all = pdist.exchange_matches(out_l, len(pl))
        tracks, _, _ = pdist.tracks_host(cnt_all, pl, out_all, max_dist=40)   # the C ABI's host form (no GPU work)
        q.put((rank, desc_all.numpy(), cnt_all.numpy(), out_all.numpy(), tracks))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_pair_sharded_run_equals_single_process():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    desc, counts, pl, matches = _single_process()
    ref_tracks = tracks_np.tracks(counts, pl, matches, 40)[0]
    assert len(ref_tracks) > 5
    for rank, d, c, m, tracks in results:
        assert (d == desc).all() and (c == counts).all(), rank
        assert (m == matches).all(), rank
        assert tracks == ref_tracks, rank


def test_sharding_helpers():
    assert pdist.local_items(7, 1, 3) == [1, 4]
    assert pdist.slots(7, 3) == 3 and pdist.slots(6, 3) == 2
    assert pdist.all_pairs(4) == [(0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3)]
    # world 1: the "gather" is the identity on the owned prefix
    t = torch.arange(12, dtype=torch.int32).reshape(4, 3)
    assert (pdist.all_gather_slots(t, 3) == t[:3]).all()


def _random_lists(seed, F=7, kmax=9, dmax=60):
    rng = np.random.default_rng(seed)
    counts = rng.integers(0, kmax, F).astype(np.int32)
    pl = [(a, b) for a in range(F) for b in range(F) if a != b]
    m = np.zeros((len(pl), kmax - 1, 3), dtype=np.int32)
    m[..., 0] = rng.integers(0, kmax, m.shape[:2])
    m[..., 1] = rng.integers(0, kmax, m.shape[:2])
    m[..., 2] = rng.integers(0, dmax, m.shape[:2])
    return counts, pl, m


def test_host_track_graph_equals_oracle():
    """pgx_tracks_* (the C ABI's host form) against the sequential oracle (oracle/tracks_np.py; parity unpinned by
    construction: the reference has no track graph) on the 5-frame sequence and on random lists with conflicts."""
    desc, counts, pl, matches = _single_process()
    for md in (25, 40, 300):
        exp, _, summ = tracks_np.tracks(counts, pl, matches, md)
        got, dropped, dropped_nodes = pdist.tracks_host(counts, pl, matches, max_dist=md)
        assert got == exp and (dropped, dropped_nodes) == (summ["dropped"], summ["dropped_nodes"])
    for seed in range(6):
        counts, pl, m = _random_lists(seed)
        for min_len in (1, 2, 3):
            for md in (5, 30):
                exp, _, summ = tracks_np.tracks(counts, pl, m, md, min_len)
                got, dropped, dropped_nodes = pdist.tracks_host(counts, pl, m, max_dist=md, min_len=min_len)
                assert got == exp and (dropped, dropped_nodes) == (summ["dropped"], summ["dropped_nodes"])


def test_oracle_track_graph_two_formulations_agree():
    """The sequential union-find and scipy's connected_components on the same gated edge set."""
    for seed in range(8):
        counts, pl, m = _random_lists(100 + seed, F=6 + seed % 3)
        for md in (5, 20, 59):
            a = tracks_np.tracks(counts, pl, m, md, 2)
            b = tracks_np.tracks_csgraph(counts, pl, m, md, 2)
            assert a[0] == b[0] and (a[1] == b[1]).all() and a[2] == b[2]
    desc, counts, pl, matches = _single_process()
    a, b = tracks_np.tracks(counts, pl, matches, 40), tracks_np.tracks_csgraph(counts, pl, matches, 40)
    assert a[0] == b[0] and (a[1] == b[1]).all() and a[2] == b[2] and a[2]["n_tracks"] > 5


def test_track_graph_drops_components_with_two_keypoints_of_one_frame():
    counts = [2, 2, 2]
    pl = [(0, 1), (1, 2), (0, 2)]
    lists = [[[0, 0, 5], [1, 1, 99]],          # second edge fails the distance gate
             [[0, 1, 3], [1, 0, 2**31 - 1]],   # the int.MaxValue tail never links, whatever the gate
             [[1, 1, 2], [0, 0, 50]]]          # (0,1)-(2,1) puts (0,0) and (0,1) into one component
    exp, track_of, summ = tracks_np.tracks(counts, pl, lists, 10)
    assert exp == [] and summ["dropped"] == 1 and summ["dropped_nodes"] == 4 and (track_of[[0, 0, 1, 2], [0, 1, 0, 1]] == -2).all()
    assert pdist.tracks_host(counts, pl, np.array(lists), max_dist=10) == ([], 1, 4)
    lists[2] = [[1, 1, 20], [0, 0, 50]]        # without that edge: one consistent track
    exp, track_of, summ = tracks_np.tracks(counts, pl, lists, 10)
    assert exp == [[(0, 0), (1, 0), (2, 1)]] and summ["dropped"] == 0 and track_of[0, 1] == -1
    assert pdist.tracks_host(counts, pl, np.array(lists), max_dist=10) == (exp, 0, 0)
    # the order the pairs arrive in changes nothing
    assert pdist.tracks_host(counts, pl[::-1], np.array(lists[::-1]), max_dist=10) == (exp, 0, 0)


# ---- ShardedSequence (the class bench.py runs on the GPUs) on CPU tensors with a stand-in engine ----------

class _OracleEngine:
    """Stand-in for pg.Engine in the gloo test: the same two batched entry points, computed by the oracle on CPU
    tensors.  What is under test is ShardedSequence's sharding, slot addressing and in-place all-gathers."""

    def __init__(self, pairs):
        self.pairs = pairs

    def detect_batch_dev(self, frames, F, W_, H_, kp, desc, counts, nraw, cap):
        for k in range(F):
            d, n = _detect(frames[k].numpy(), self.pairs)
            desc[k] = torch.from_numpy(d[:cap])
            counts[k] = min(n, cap)

    def match_batch_dev(self, desc, counts, stride, words, pairlist, M, out, max_count=None):
        d, c = desc.numpy(), counts.numpy()
        for m in range(M):
            a, b = int(pairlist[m, 0]), int(pairlist[m, 1])
            out[m] = torch.from_numpy(_match(d[a], int(c[a]), d[b], int(c[b])))

    def tracks_dev(self, matches, counts, pairlist, M, F, stride, n_frames, max_dist, min_len, track_of, offsets, nodes, summary,
                   d_frame_ids=None):
        """pgx_tracks_dev's contract on CPU tensors, computed by the oracle (slots -> frame numbers, -1 slots skipped)."""
        ids = d_frame_ids.numpy() if d_frame_ids is not None else np.arange(F)
        c, pl, m = counts.numpy(), pairlist.numpy(), matches.numpy()
        cnt = np.zeros(n_frames, dtype=np.int64)
        for s_ in range(F):
            if ids[s_] >= 0:
                cnt[ids[s_]] = c[s_]
        rows = [(int(ids[a]), int(ids[b]), m[r]) for r, (a, b) in enumerate(pl[:M]) if a >= 0 and b >= 0 and ids[a] >= 0 and ids[b] >= 0]
        tr, tof, summ = tracks_np.tracks(cnt, [(a, b) for a, b, _ in rows], [x for _, _, x in rows], max_dist, min_len)
        track_of.fill_(-1)
        track_of[:, :tof.shape[1]] = torch.from_numpy(tof)
        flat = [n for t in tr for n in t]
        off = np.concatenate([[0], np.cumsum([len(t) for t in tr])]).astype(np.int32)
        offsets[:len(off)] = torch.from_numpy(off)
        if flat:
            nodes[:len(flat)] = torch.tensor(flat, dtype=torch.int32)
        summary[:] = torch.tensor([summ["n_tracks"], summ["n_nodes"], summ["dropped"], summ["dropped_nodes"], summ["edges"],
                                   summ["longest"], summ["largest_dropped"], 0], dtype=torch.int32)


def _seq_worker(rank, world, port, q, overlap=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pairs = cref.gaussian_pairs(0, 20, 256)
        fr = _frames()
        pl = pdist.all_pairs(N_FRAMES)
        job = pdist.ShardedSequence(_OracleEngine(pairs), W, H, N_FRAMES, pl, CAP, 8, "cpu", overlap_exchange=overlap,
                                    tracks={"max_dist": 40})
        mine = torch.from_numpy(np.stack([fr[f] for f in job.my_frames]))
        job.step(mine)
        job.step(mine)   # a second step over the same buffers must give the same answer
        if overlap:
            job.step(mine)   # three steps: both output buffers have been gathered into and reused
        job.finish()
        desc = np.stack([job.descriptors(f).numpy() for f in range(N_FRAMES)])
        out = np.stack([job.matches(p).numpy() for p in range(len(pl))])
        q.put((rank, desc, job.counts(), out, job.tracks(), job.track_summary()))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world,overlap", [(2, False), (3, False), (2, True), (3, True)])
def test_sharded_sequence_equals_single_process(world, overlap):
    """5 frames / 10 image pairs over 2 and 3 ranks (uneven shares: padding slots on the last ranks); with and without the
    overlapped (asynchronous, double-buffered) exchange of the match lists."""
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_seq_worker, args=(r, world, port, q, overlap)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    desc, counts, pl, matches = _single_process()
    ref_tracks, _, ref_summ = tracks_np.tracks(counts, pl, matches, 40)
    for rank, d, c, m, tracks, summ in results:
        assert (c == counts).all(), rank
        assert (d == desc).all(), rank
        assert (m == matches).all(), rank
        # the track graph of the gathered lists (slot-addressed rank-major buffers, padding rows) = the single-process graph
        assert tracks == ref_tracks and summ["n_tracks"] == ref_summ["n_tracks"] > 5, rank


def _interleaved_worker(rank, world, port, q):
    """Two jobs in flight the way bench.py issues them: front(s + 1) before back(s), overlapped list exchange."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pairs = cref.gaussian_pairs(0, 20, 256)
        fr = _frames()
        pl = pdist.all_pairs(N_FRAMES)
        jobs = [pdist.ShardedSequence(_OracleEngine(pairs), W, H, N_FRAMES, pl, CAP, 8, "cpu", overlap_exchange=True) for _ in range(2)]
        mine = torch.from_numpy(np.stack([fr[f] for f in jobs[0].my_frames]))
        n = 5
        jobs[0].front(mine)
        for s in range(n):
            if s + 1 < n:
                jobs[(s + 1) % 2].front(mine)
            jobs[s % 2].back()
        for j in jobs:
            j.finish()
        res = []
        for j in jobs:
            res.append((np.stack([j.descriptors(f).numpy() for f in range(N_FRAMES)]), j.counts(),
                        np.stack([j.matches(p).numpy() for p in range(len(pl))])))
        q.put((rank, res))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 3])
def test_two_jobs_with_interleaved_halves_equal_single_process(world):
    """front(s + 1) is issued before back(s) (bench.py with two jobs in flight: on the communicator the next step's descriptor
    gather then stands in front of this step's list gather); every rank issues the same sequence of collectives, and both
    jobs must end with the single-process result."""
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_interleaved_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    desc, counts, pl, matches = _single_process()
    for rank, res in results:
        for d, c, m in res:
            assert (c == counts).all(), rank
            assert (d == desc).all(), rank
            assert (m == matches).all(), rank


def test_slot_of_is_a_bijection():
    for world in (1, 2, 3, 8):
        for n in (1, 5, 64, 2016):
            ns = pdist.slots(n, world)
            got = sorted(pdist.slot_of(i, world, ns) for i in range(n))
            assert len(set(got)) == n and got[-1] < world * ns
            for r in range(world):   # rank r's block holds exactly its items, in ownership order
                assert [pdist.slot_of(i, world, ns) for i in pdist.local_items(n, r, world)] == \
                       list(range(r * ns, r * ns + len(pdist.local_items(n, r, world))))


# ---- a rank that fails before the first exchange must not leave its peers waiting (pgx_comm.hip / dist.py) ----

class _FailingEngine(_OracleEngine):
    def detect_batch_dev(self, *a, **k):
        raise ValueError("pgx_set_detect_params not called")   # what an unconfigured context answers


def _fail_worker(rank, world, port, q, bad_rank):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pairs = cref.gaussian_pairs(0, 20, 256)
        fr = _frames()
        pl = pdist.all_pairs(N_FRAMES)
        eng = _FailingEngine(pairs) if rank == bad_rank else _OracleEngine(pairs)
        job = pdist.ShardedSequence(eng, W, H, N_FRAMES, pl, CAP, 8, "cpu")
        mine = torch.from_numpy(np.stack([fr[f] for f in job.my_frames]))
        try:
            job.step(mine)
            q.put((rank, "no error"))
        except Exception as ex:  # noqa: BLE001
            q.put((rank, "%s: %s" % (type(ex).__name__, ex)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
@pytest.mark.parametrize("world,bad_rank", [(2, 1), (3, 0)])
def test_rank_failing_before_the_first_exchange_fails_every_rank(world, bad_rank):
    """One rank's local part raises before any collective: every rank must come back with an error (the failing one with
    its own, the others naming it) instead of waiting in the all-gather -- the whole test would time out otherwise."""
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_fail_worker, args=(r, world, port, q, bad_rank)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=90) for _ in range(world))
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    assert results[bad_rank].startswith("ValueError: pgx_set_detect_params not called")
    for r in range(world):
        if r != bad_rank:
            assert results[r].startswith("RuntimeError: rank(s) [%d] failed before the first exchange" % bad_rank), results[r]


def _short_batch_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pairs = cref.gaussian_pairs(0, 20, 256)
        fr = _frames()
        pl = pdist.all_pairs(N_FRAMES)
        job = pdist.ShardedSequence(_OracleEngine(pairs), W, H, N_FRAMES, pl, CAP, 8, "cpu")
        mine = torch.from_numpy(np.stack([fr[f] for f in job.my_frames]))
        job.step(mine)
        # second step: the LAST rank's batch is one frame short (its other slots keep the first step's records); the ranks
        # must still issue the same collectives -- the whole test would hang otherwise
        job.step(mine, n_local=len(job.my_frames) - 1 if rank == world - 1 else None)
        job.step(mine)
        out = np.stack([job.matches(p).numpy() for p in range(len(pl))])
        q.put((rank, job.counts(), out))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 3])
def test_a_rank_whose_local_share_changes_issues_the_same_collectives(world):
    """ADVICE r4 (pgx_comm.hip / dist.py): the status exchange before the first collective must be decided from rank-symmetric
    state only.  One rank's local frame count changes between steps; nobody may hang and the results stay the single-process ones."""
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_short_batch_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    desc, counts, pl, matches = _single_process()
    for rank, c, m in results:
        assert (c == counts).all() and (m == matches).all(), rank
