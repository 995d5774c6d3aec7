"""GPU parity tests for the configurations beyond one image pair:

  * the class bench.py runs (dist.ShardedSequence, world 1) on a config-3-shaped sequence of full-size 1920x1080
    frames, all ordered image pairs, EVERY pair against the oracle's sorted-edge-scan matcher;
  * one 8192 x 8192 match (BASELINE configs[3]'s keypoint count);
  * one pair of 3840x2160 frames, detect -> match end to end;
  * >= 3 workspace chunks of image pairs WITH wide (whole-chip MFMA) rounds, so the two-stream overlap of
    enqueue_match (wide rounds of chunk i + 1 beside the per-pair finish of chunk i) runs for real;
  * a batch of odd-sized frames (W*H not a multiple of 4) through pgx_detect_batch_dev.
"""
import numpy as np
import pytest
import torch

from oracle import cref, tracks_np
from photogrammetry_amd import dist as pdist
from photogrammetry_amd import synth
import photogrammetry_amd as pg

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
T = np.float32(0.1)


@pytest.fixture(scope="module")
def engine():
    e = pg.Engine(0)
    yield e
    e.close()


def _oracle_detect(frame, dmap, pairs, radius, cap):
    src = cref.apply_distortion(frame, dmap) if dmap is not None else frame
    g = cref.gray(src)
    raw = cref.detect(g, T)
    kept = raw[cref.nms(raw, radius)][:cap]
    return kept, cref.brief(g, np.stack([kept["x"], kept["y"]], 1), pairs), len(raw)


def _same(got, exp):
    return bool((got[:, 0] == exp["k1"]).all() and (got[:, 1] == exp["k2"]).all() and (got[:, 2] == exp["dist"]).all())


def test_sharded_sequence_world1_config3_shape(engine):
    """8 frames of 1920x1080 (frame_i = frame_0 translated by (3i, i)), NMS r = 16, lists cut to
    4096 keypoints by the survivor limit, all 28 ordered image pairs: detect of two frames and EVERY match list
    against the oracle."""
    W, H, NKP, radius, F = 1920, 1080, 4096, 16, 8
    pairs = pg.make_brief_pairs(0, 50, 256)
    dmap = pg.build_dewarp_map(W, H, [3e-4, 1e-7, 0, 0, 0])
    engine.set_brief_pairs(pairs)
    engine.set_detect_params(T, radius)
    engine.set_capacity(1 << 18, NKP)
    engine.set_dewarp_map(dmap)
    base = synth.make_frame(W, H, seed=4321, n_shapes=20000)
    d_base = torch.from_numpy(base).to(DEV)
    d_frames = torch.empty((F, H, W, 4), dtype=torch.uint16, device=DEV)
    for i in range(F):   # wrap-around translation, as bench.py makes them
        d_frames.view(torch.int64)[i] = torch.roll(d_base.view(torch.int64), shifts=(i % H, (3 * i) % W), dims=(0, 1))
    pl = pdist.all_pairs(F)
    stream = torch.cuda.Stream(device=DEV)
    job = pdist.ShardedSequence(engine, W, H, F, pl, NKP, 8, DEV, stream=stream, tracks={"max_dist": 64, "min_len": 2})
    torch.cuda.synchronize()
    job.step(d_frames)
    job.step(d_frames)          # buffers are reused step after step
    engine.check_status()       # the survivor limit cuts silently: no CapacityError although > 4096 survive
    counts = job.counts()
    assert counts.max() == NKP and counts.min() > 3500, counts   # some frames keep a few less than the limit
    desc = [job.descriptors(f).cpu().numpy().view(np.uint32)[:counts[f]] for f in range(F)]
    frames_h = d_frames.cpu().numpy()
    for f in (0, 5):
        kept, edesc, n_raw = _oracle_detect(frames_h[f], dmap, pairs, radius, NKP)
        n = len(kept)
        assert n == counts[f] and int(job.nraw_l[f]) == n_raw
        kp = job.kp_l[f].cpu().numpy()
        assert (kp[:n, 0] == kept["x"]).all() and (kp[:n, 1] == kept["y"]).all()
        assert (desc[f] == edesc).all()
    for m, (a, b) in enumerate(pl):
        got = job.matches(m).cpu().numpy()[:counts[a]]
        assert _same(got, cref.match_sorted(desc[a], desc[b])), (a, b)
    # the track graph the step built on the device from these 28 lists (pgx_tracks_dev; parity unpinned by
    # construction -- the reference has no track graph) against the sequential oracle: same tracks, order, per-node ids
    lists = np.stack([job.matches(m).cpu().numpy() for m in range(len(pl))])
    exp, exp_tof, exp_s = tracks_np.tracks(counts, pl, lists, 64, 2)
    assert job.tracks() == exp and exp_s["n_tracks"] > 1000
    assert {k: v for k, v in job.track_summary().items()} == exp_s
    tof = job.track_of.cpu().numpy()
    assert (tof[:, :exp_tof.shape[1]] == exp_tof).all()
    engine.set_stream(0)
    engine.set_dewarp_map(None)
    engine.set_capacity(1 << 17, 1 << 20)


def _match_dev(engine, descs, pl, stride, max_count=None):
    F = len(descs)
    d = np.zeros((F, stride, 8), dtype=np.uint32)
    for f in range(F):
        d[f, :len(descs[f])] = descs[f]
    d_desc = torch.from_numpy(d.view(np.int32)).to(DEV)
    d_counts = torch.tensor([len(x) for x in descs], dtype=torch.int32, device=DEV)
    d_pl = torch.tensor(pl, dtype=torch.int32, device=DEV)
    d_out = torch.full((len(pl), stride, 3), -7, dtype=torch.int32, device=DEV)
    torch.cuda.synchronize()
    engine.match_batch_dev(d_desc, d_counts, stride, 8, d_pl, len(pl), d_out, max_count=max_count)
    engine.check_status()
    return d_out.cpu().numpy()


def test_match_8192_squared_vs_oracle(engine):
    """BASELINE configs[3]'s list length: uniform-random (tie-heavy: distances ~ Binomial(256, 1/2)) and a
    true-correspondence pair (permuted copy with 15 % of the bits flipped), both directions."""
    N = 8192
    a = synth.random_descriptors(N, 8, 81)
    b = synth.random_descriptors(N, 8, 82)
    t1, t2, _ = synth.true_match_descriptors(N, 8, 83)
    out = _match_dev(engine, [a, b, t1, t2], [(0, 1), (1, 0), (2, 3), (3, 2)], N)
    sets = [a, b, t1, t2]
    for m, (x, y) in enumerate([(0, 1), (1, 0), (2, 3), (3, 2)]):
        assert _same(out[m][:N], cref.match_sorted(sets[x], sets[y])), (x, y)
    rounds, evals, evals0 = engine.match_stats()
    assert rounds >= 3 and evals0 == 4 * N * N


def test_detect_and_match_4k_pair_end_to_end(engine):
    """Two 3840x2160 frames (the second translated by (37, 11)), r = 22, lists cut to 8192: both detect lists and
    the match list against the oracle."""
    W, H, NKP, radius = 3840, 2160, 8192, 22
    pairs = pg.make_brief_pairs(0, 50, 256)
    dmap = pg.build_dewarp_map(W, H, [3e-4, 1e-7, 0, 0, 0])
    engine.set_brief_pairs(pairs)
    engine.set_detect_params(T, radius)
    engine.set_capacity(1 << 20, NKP)
    engine.set_dewarp_map(dmap)
    f0 = synth.make_frame(W, H, seed=5, n_shapes=80000)
    frames = np.stack([f0, synth.shift_frame(f0, 37, 11)])
    d_frames = torch.from_numpy(frames).to(DEV)
    i32 = dict(dtype=torch.int32, device=DEV)
    d_kp, d_desc = torch.zeros((2, NKP, 4), **i32), torch.zeros((2, NKP, 8), **i32)
    d_counts, d_nraw = torch.zeros(2, **i32), torch.zeros(2, **i32)
    d_pl = torch.tensor([[0, 1]], **i32)
    d_out = torch.zeros((1, NKP, 3), **i32)
    torch.cuda.synchronize()
    engine.detect_batch_dev(d_frames, 2, W, H, d_kp, d_desc, d_counts, d_nraw, NKP)
    engine.match_batch_dev(d_desc, d_counts, NKP, 8, d_pl, 1, d_out, max_count=NKP)
    engine.check_status()
    desc = d_desc.cpu().numpy().view(np.uint32)
    exp = [_oracle_detect(frames[f], dmap, pairs, radius, NKP) for f in range(2)]
    for f in range(2):
        kept, edesc, n_raw = exp[f]
        assert len(kept) == NKP and int(d_counts[f]) == NKP and int(d_nraw[f]) == n_raw
        kp = d_kp[f].cpu().numpy()
        assert (kp[:, 0] == kept["x"]).all() and (kp[:, 1] == kept["y"]).all() and (kp[:, 2] == kept["fast_score"]).all()
        assert (desc[f] == edesc).all()
    assert _same(d_out[0].cpu().numpy(), cref.match_sorted(exp[0][1], exp[1][1]))
    engine.set_dewarp_map(None)
    engine.set_capacity(1 << 17, 1 << 20)


def test_match_three_chunks_with_wide_rounds(engine):
    """300 image pairs (3 workspace chunks of 128) over 25 descriptor sets of 2100-2500 entries (> PGX_TAIL_MAX, so every
    chunk runs whole-chip MFMA rounds on one stream while the previous chunk's per-pair finish runs on the other), with
    near-duplicate descriptors across sets (tie-heavy).  Every second pair, and the pairs at the chunk borders, against
    the oracle; run with the profiling hooks on."""
    rng = np.random.default_rng(18)
    F = 25
    sizes = [int(x) for x in rng.integers(2100, 2500, F)]
    base = rng.integers(0, 2**32, (2800, 8), dtype=np.uint32)
    descs = []
    for f in range(F):
        d = base[rng.permutation(2800)[:sizes[f]]].copy()
        d[:, 5] ^= rng.integers(0, 16, sizes[f]).astype(np.uint32)   # near-duplicates across sets
        descs.append(d)
    pl = [(a, b) for a in range(F) for b in range(a + 1, F)]
    assert len(pl) == 300
    engine.profile_reset()
    engine.profile_enable(True)
    engine.set_match_chunk(128)
    out = _match_dev(engine, descs, pl, 2560)
    engine.set_match_chunk(2048)
    engine.profile_enable(False)
    n_wide, _ = engine.profile_get("ham_argmin")
    n_fin, _ = engine.profile_get("match_finish")
    assert n_fin == 3 and n_wide >= 3              # three chunks, each with at least one wide round
    rounds, evals, evals0 = engine.match_stats()
    assert rounds >= 1 and evals0 == sum(sizes[a] * sizes[b] for a, b in pl)
    for m, (a, b) in enumerate(pl):
        if m % 2 == 0 or m % 128 in (0, 1, 126, 127):
            assert _same(out[m][:sizes[a]], cref.match_sorted(descs[a], descs[b])), (a, b)


@pytest.mark.parametrize("W,H,F", [(451, 383, 5), (333, 77, 6), (127, 129, 9)])
def test_detect_batch_odd_frame_sizes(engine, W, H, F):
    """W*H not a multiple of 4 with F > 1: frame f's base is not 16-byte aligned for f > 0."""
    assert (W * H) % 4 != 0
    CAP, radius = 2048, 12
    pairs = pg.make_brief_pairs(9, 30, 256)
    dmap = pg.build_dewarp_map(W, H, [3e-4, 1e-7, 0, 0, 0])
    engine.set_brief_pairs(pairs)
    engine.set_detect_params(T, radius)
    engine.set_capacity(1 << 16, CAP)
    engine.set_dewarp_map(dmap)
    frames = np.stack([synth.make_frame(W, H, seed=70 + i) for i in range(F)])
    d_frames = torch.from_numpy(frames).to(DEV)
    i32 = dict(dtype=torch.int32, device=DEV)
    d_kp, d_desc = torch.zeros((F, CAP, 4), **i32), torch.zeros((F, CAP, 8), **i32)
    d_counts, d_nraw = torch.zeros(F, **i32), torch.zeros(F, **i32)
    torch.cuda.synchronize()
    engine.detect_batch_dev(d_frames, F, W, H, d_kp, d_desc, d_counts, d_nraw, CAP)
    engine.check_status()
    desc = d_desc.cpu().numpy().view(np.uint32)
    for f in range(F):
        kept, edesc, n_raw = _oracle_detect(frames[f], dmap, pairs, radius, CAP)
        n = len(kept)
        assert int(d_nraw[f]) == n_raw and int(d_counts[f]) == n, f
        kp = d_kp[f].cpu().numpy()
        assert (kp[:n, 0] == kept["x"]).all() and (kp[:n, 1] == kept["y"]).all()
        assert kp[:n, 3].view(np.float32).tobytes() == kept["value"].tobytes()
        assert (desc[f, :n] == edesc).all()
    engine.set_dewarp_map(None)


def test_survivor_limit_cuts_silently_and_capacity_still_raises(engine):
    """pgx_set_capacity's max_keypoints_per_frame is a LIMIT (harness-side truncation), the per-call capacity is not."""
    W, H, radius = 640, 360, 8
    pairs = pg.make_brief_pairs(2, 30, 256)
    engine.set_brief_pairs(pairs)
    engine.set_detect_params(T, radius)
    engine.set_dewarp_map(None)
    frame = synth.make_frame(W, H, seed=33, n_shapes=1500)
    kept, edesc, _ = _oracle_detect(frame, None, pairs, radius, 1 << 20)
    assert len(kept) > 300
    engine.set_capacity(1 << 17, 200)
    kp, desc, _ = engine.detect(frame, capacity=4096)
    assert len(kp) == 200 and (desc == edesc[:200]).all() and (kp["x"] == kept["x"][:200]).all()
    engine.set_capacity(1 << 17, 1 << 20)
    with pytest.raises(pg.CapacityError):
        engine.detect(frame, capacity=200)


def test_c_abi_communicator_world1_and_sequence_step(engine):
    """pgx_comm_* at world size 1 (RCCL really initialised on this GPU) and pgx_sequence_step_dev = the four phases in ONE
    C call; its result must equal the torch-side ShardedSequence on the same frames.  (N > 1 runs on the 8-GPU node in
    bench.py's c_abi_comm leg; the sharding arithmetic is covered by the world-2/3 gloo tests.)"""
    W, H, NKP, radius, F = 640, 360, 1024, 12, 5
    pairs = pg.make_brief_pairs(4, 40, 256)
    engine.set_brief_pairs(pairs)
    engine.set_detect_params(T, radius)
    engine.set_capacity(1 << 17, NKP)
    engine.set_dewarp_map(None)
    frames = np.stack([synth.make_frame(W, H, seed=90 + i, n_shapes=900) for i in range(F)])
    d_frames = torch.from_numpy(frames).to(DEV)
    pl = pdist.all_pairs(F)
    stream = torch.cuda.Stream(device=DEV)
    assert engine.comm_info() == (0, 1)
    uid = pg.comm_unique_id()
    assert len(uid) == 128
    engine.comm_init(0, 1, uid)
    with pytest.raises(pg.ArgumentException):
        engine.comm_init(0, 1, uid)                 # a context owns ONE communicator
    assert engine.comm_info() == (0, 1)
    ja = pdist.ShardedSequence(engine, W, H, F, pl, NKP, 8, DEV, stream=stream, comm="torch")
    jb = pdist.ShardedSequence(engine, W, H, F, pl, NKP, 8, DEV, stream=stream, comm="pgx")
    torch.cuda.synchronize()
    ja.step(d_frames)
    jb.step(d_frames)
    jb.step(d_frames)
    engine.allgather_dev(jb.out_all, jb.out_all.numel() * 4)      # world 1: a no-op on the context's stream
    engine.check_status()
    assert torch.equal(ja.counts_all, jb.counts_all) and int(ja.counts_all.min()) > 50
    assert torch.equal(ja.desc_all, jb.desc_all) and torch.equal(ja.out_all, jb.out_all)
    counts = jb.counts()
    for m in (0, len(pl) - 1):
        a, b = pl[m]
        da = jb.descriptors(a).cpu().numpy().view(np.uint32)[:counts[a]]
        db = jb.descriptors(b).cpu().numpy().view(np.uint32)[:counts[b]]
        assert _same(jb.matches(m).cpu().numpy()[:counts[a]], cref.match_sorted(da, db))
    engine.comm_destroy()
    engine.comm_destroy()                           # idempotent
    engine.set_stream(0)
    engine.set_capacity(1 << 17, 1 << 20)


def test_tail_fallback_distance_256_and_oversized_residual(engine):
    """The two cases k_tail_rows leaves to the any-size form inside k_match_gs: (a) complementary descriptors (distance
    256 does not fit the byte matrix), (b) a residual that is still larger than PGX_TAIL_MAX after the planned wide rounds
    (all-identical descriptors: every round accepts exactly one edge).  Both against the oracle."""
    rng = np.random.default_rng(256)
    a = rng.integers(0, 2**32, (300, 8), dtype=np.uint32)
    b = np.concatenate([~a[:40], rng.integers(0, 2**32, (200, 8), dtype=np.uint32), a[100:130]])[rng.permutation(270)]
    a[5] = 0
    b[7] = 0xFFFFFFFF                                          # an exact (0, all-ones) pair as well
    ident1 = np.full((2200, 8), 0x5A5A5A5A, dtype=np.uint32)
    ident2 = np.full((2150, 8), 0x5A5A5A5A, dtype=np.uint32)
    ident2[::7, 3] ^= 1                                        # a few at distance 1
    sets = [a, np.ascontiguousarray(b), ident1, ident2]
    pl = [(0, 1), (1, 0), (2, 3)]
    out = _match_dev(engine, sets, pl, 2304)
    bits = lambda d: np.unpackbits(d.view(np.uint8), axis=1).astype(np.int32)
    assert ((bits(a)[:, None, :] != bits(sets[1])[None, :, :]).sum(2) == 256).any()   # the byte matrix would overflow
    for m, (x, y) in enumerate(pl):
        assert _same(out[m][:len(sets[x])], cref.match_sorted(sets[x], sets[y])), (x, y)


def _two_rank_worker(rank, world, port, q):
    import os
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)   # RCCL refuses two ranks on one GPU; gloo moves CUDA tensors
    try:
        W, H, NKP, radius, F = 640, 360, 1024, 12, 7
        e = pg.Engine(0)
        e.set_brief_pairs(pg.make_brief_pairs(4, 40, 256))
        e.set_detect_params(T, radius)
        e.set_capacity(1 << 17, NKP)
        e.set_dewarp_map(None)
        frames = np.stack([synth.make_frame(W, H, seed=90 + i, n_shapes=900) for i in range(F)])
        pl = pdist.all_pairs(F)
        job = pdist.ShardedSequence(e, W, H, F, pl, NKP, 8, DEV, stream=torch.cuda.Stream(device=DEV))
        mine = torch.from_numpy(np.stack([frames[f] for f in job.my_frames])).to(DEV)
        torch.cuda.synchronize()
        job.step(mine)
        job.step(mine)
        e.check_status()
        torch.cuda.synchronize()
        desc = np.stack([job.descriptors(f).cpu().numpy() for f in range(F)])
        out = np.stack([job.matches(p).cpu().numpy() for p in range(len(pl))])
        q.put((rank, desc, job.counts(), out))
        e.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_and_three_ranks_sharing_this_gpu_equal_one_rank(engine):
    """The N > 1 path on real pgx contexts: 2 and 3 processes (one context each, all on GPU 0, gloo carrying the CUDA
    tensors) run ShardedSequence on 7 frames / 21 image pairs (uneven shares); every rank must end with exactly the
    descriptors, counts and match lists of the one-process run."""
    import socket
    import torch.multiprocessing as mp
    W, H, NKP, radius, F = 640, 360, 1024, 12, 7
    engine.set_brief_pairs(pg.make_brief_pairs(4, 40, 256))
    engine.set_detect_params(T, radius)
    engine.set_capacity(1 << 17, NKP)
    engine.set_dewarp_map(None)
    frames = np.stack([synth.make_frame(W, H, seed=90 + i, n_shapes=900) for i in range(F)])
    pl = pdist.all_pairs(F)
    ref = pdist.ShardedSequence(engine, W, H, F, pl, NKP, 8, DEV, stream=torch.cuda.Stream(device=DEV))
    d_frames = torch.from_numpy(frames).to(DEV)
    torch.cuda.synchronize()
    ref.step(d_frames)
    engine.check_status()
    torch.cuda.synchronize()
    rdesc = np.stack([ref.descriptors(f).cpu().numpy() for f in range(F)])
    rout = np.stack([ref.matches(p).cpu().numpy() for p in range(len(pl))])
    rcounts = ref.counts()
    assert rcounts.min() > 50
    for world in (2, 3):
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        procs = [ctx.Process(target=_two_rank_worker, args=(r, world, port, q)) for r in range(world)]
        for p in procs:
            p.start()
        results = [q.get(timeout=300) for _ in range(world)]
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        for rank, d, c, o in results:
            assert (c == rcounts).all(), (world, rank)
            assert (d == rdesc).all() and (o == rout).all(), (world, rank)
    engine.set_stream(0)
    engine.set_capacity(1 << 17, 1 << 20)


@pytest.mark.parametrize("radius", [16, 12, 21])
def test_detect_batch_every_frame_repeated(engine, radius):
    """NMS order-independence under load: 8 full-size frames per call, three calls, EVERY frame of every call against the
    literal oracle (the mask rounds of k_nms.hip decide cells concurrently across the whole chip; r = 16/12 reach 2 cells,
    r = 21 reaches 3)."""
    W, H, F, CAP = 1920, 1080, 8, 8192
    pairs = pg.make_brief_pairs(0, 50, 256)
    dmap = pg.build_dewarp_map(W, H, [3e-4, 1e-7, 0, 0, 0])
    engine.set_brief_pairs(pairs)
    engine.set_detect_params(T, radius)
    engine.set_capacity(1 << 18, CAP)
    engine.set_dewarp_map(dmap)
    base = synth.make_frame(W, H, seed=4321, n_shapes=20000)
    d_base = torch.from_numpy(base).to(DEV)
    d_frames = torch.empty((F, H, W, 4), dtype=torch.uint16, device=DEV)
    for i in range(F):
        d_frames.view(torch.int64)[i] = torch.roll(d_base.view(torch.int64), shifts=(i % H, (3 * i) % W), dims=(0, 1))
    d_kp = torch.zeros((F, CAP, 4), dtype=torch.int32, device=DEV)
    d_desc = torch.zeros((F, CAP, 8), dtype=torch.int32, device=DEV)
    d_cnt = torch.zeros((F,), dtype=torch.int32, device=DEV)
    d_nraw = torch.zeros((F,), dtype=torch.int32, device=DEV)
    torch.cuda.synchronize()
    frames_h = d_frames.cpu().numpy()
    expect = []
    for f in range(F):
        g = cref.gray(cref.apply_distortion(frames_h[f], dmap))
        raw = cref.detect(g, T)
        expect.append(raw[cref.nms(raw, radius)])
    for rep in range(3):
        engine.detect_batch_dev(d_frames.data_ptr(), F, W, H, d_kp.data_ptr(), d_desc.data_ptr(), d_cnt.data_ptr(),
                                d_nraw.data_ptr(), CAP)
        engine.check_status()
        cnt = d_cnt.cpu().numpy()
        kp = d_kp.cpu().numpy()
        for f in range(F):
            kept = expect[f]
            assert cnt[f] == len(kept), (rep, f, cnt[f], len(kept))
            assert (kp[f, :cnt[f], 0] == kept["x"]).all() and (kp[f, :cnt[f], 1] == kept["y"]).all(), (rep, f)
            assert (kp[f, :cnt[f], 2] == kept["fast_score"]).all(), (rep, f)


# ---- matcher sizes beyond BASELINE's (the reference has no cap on N: KeypointMatching.cs:20-35) -------------------------

def _hub_sets(n1, n2, protos, seed):
    """the generator of test_gpu_parity.py::test_match_duplicate_and_hub_descriptors: many identical and low-popcount
    "hub" descriptors (tie chains, one acceptance per cluster and round)"""
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 2**32, (protos, 8), dtype=np.uint32)
    base[: max(1, protos // 4)] &= rng.integers(0, 2**32, (max(1, protos // 4), 8), dtype=np.uint32) & 0x11111111
    d1 = base[rng.integers(0, protos, n1)].copy()
    d2 = base[rng.integers(0, protos, n2)].copy()
    flip = rng.random(n1) < 0.3
    d1[flip, 0] ^= np.uint32(1) << rng.integers(0, 32, int(flip.sum())).astype(np.uint32)
    return d1, d2


@pytest.mark.parametrize("n1,n2,kind", [(12000, 9000, "random"), (20000, 300, "random"), (300, 20000, "random"), (9000, 9000, "hub")])
def test_match_sizes_beyond_the_bench(engine, n1, n2, kind):
    """pgx_match through the 256-bit path at sizes where a set no longer fits one column chunk of the distance kernel
    (4096 columns, 7-bit tile field), where the plan needs 4-6 whole-chip rounds, and at very unequal sides; against the
    oracle's sorted-edge-scan matcher, bit for bit (indices, distances, the (0, 0, int.MaxValue) tail of n1 > n2)."""
    if kind == "random":
        d1, d2 = synth.random_descriptors(n1, 8, 7000 + n1), synth.random_descriptors(n2, 8, 9000 + n2)
    else:
        d1, d2 = _hub_sets(n1, n2, 1500, n1 + n2)
    got = engine.match(d1, d2)
    exp = cref.match_sorted(d1, d2)
    assert len(got) == n1
    assert (got["k1"] == exp["k1"]).all() and (got["k2"] == exp["k2"]).all() and (got["dist"] == exp["dist"]).all()
    if n1 > n2:
        assert int((got["dist"] == pg.api.PGX_DIST_NONE).sum()) == n1 - n2


@pytest.mark.parametrize("gates,chunk", [((None, 2), 2048), ((None, 3), 2048), ((1, None), 2048), ((2, 3), 2048), ((0, 1), 2048), ((None, 3), 16), ((1, 2), 16)])
def test_two_jobs_in_flight_with_stage_gates_equal_one_job(engine, gates, chunk):
    """bench.py's default form ((None, 2) = --gate none,rows): two contexts, two streams, consecutive steps alternate between them and pgx_wait_stage orders
    a step's detect chain / matcher behind stages of the previous step on the OTHER context (PGX_STAGE_*: 0 detect, 1 match
    wide, 2 match rows, 3 match done).  Ordering only: every step of either context must equal the one-job result, whatever
    the gates; frames differ between steps so that a step reading the other context's buffers would show.  chunk = 16: the 21
    image pairs go through in two workspace chunks on the library's three streams (the gate then sits in front of everything)."""
    W, H, NKP, radius, F = 640, 480, 1024, 12, 7
    pairs = pg.make_brief_pairs(0, 50, 256)
    pl = pdist.all_pairs(F)
    base = synth.make_frame(W, H, seed=99, n_shapes=3000)
    d_base = torch.from_numpy(base).to(DEV)

    def frames_of(step):
        d = torch.empty((F, H, W, 4), dtype=torch.uint16, device=DEV)
        for i in range(F):
            d.view(torch.int64)[i] = torch.roll(d_base.view(torch.int64), shifts=((i + 5 * step) % H, (3 * i + 7 * step) % W), dims=(0, 1))
        return d

    def configure(e):
        e.set_brief_pairs(pairs)
        e.set_detect_params(T, radius)
        e.set_capacity(1 << 16, NKP)
        e.set_dewarp_map(None)

    nsteps = 6
    inputs = [frames_of(s) for s in range(nsteps)]
    torch.cuda.synchronize()
    configure(engine)
    ref_job = pdist.ShardedSequence(engine, W, H, F, pl, NKP, 8, DEV, stream=torch.cuda.Stream(device=DEV))
    ref = []
    for s in range(nsteps):
        ref_job.step(inputs[s])
        torch.cuda.synchronize()
        engine.check_status()
        ref.append((ref_job.out_all.clone(), ref_job.counts_all.clone(), ref_job.desc_all.clone()))
    assert int(ref[0][1].min()) > 50
    engs = [pg.Engine(0), pg.Engine(0)]
    try:
        jobs = []
        for e in engs:
            configure(e)
            e.set_match_chunk(chunk)
            jobs.append(pdist.ShardedSequence(e, W, H, F, pl, NKP, 8, DEV, stream=torch.cuda.Stream(device=DEV)))
        got = []
        for s in range(nsteps):   # nothing synchronises between the steps: results are copied on the job's own stream
            k = s % 2
            jobs[k].step(inputs[s], after=(engs[1 - k],) + tuple(gates) if s > 0 else None)
            with torch.cuda.stream(jobs[k].stream):
                got.append((jobs[k].out_all.clone(), jobs[k].counts_all.clone(), jobs[k].desc_all.clone()))
        torch.cuda.synchronize()
        for e in engs:
            e.check_status()
        for s in range(nsteps):
            for a, b in zip(got[s], ref[s]):
                assert torch.equal(a, b), "step %d differs from the one-job result" % s
        # argument checks of the new entry point
        with pytest.raises(pg.ArgumentException):
            engs[0].wait_stage(engs[1], 4)
        with pytest.raises(pg.ArgumentException):
            engs[0].wait_stage(engs[1], -1)
        engs[0].wait_stage(engs[0], 1)   # a context and itself: a stream is in order with itself
    finally:
        for e in engs:
            e.close()


def test_match_many_pairs_with_sets_above_4096(engine):
    """561 image pairs in ONE launch (>= 512: the per-pair finish runs its 512-thread form) over 34 descriptor sets of
    4200-4700 entries (above the 4096 of the counting emit: the closing sort is the bitonic one over the finish's own LDS),
    tie-heavy like the three-chunk test; pairs spread over the launch against the oracle."""
    rng = np.random.default_rng(77)
    F = 34
    sizes = [int(x) for x in rng.integers(4200, 4700, F)]
    base = rng.integers(0, 2**32, (5200, 8), dtype=np.uint32)
    descs = []
    for f in range(F):
        d = base[rng.permutation(5200)[:sizes[f]]].copy()
        d[:, 2] ^= rng.integers(0, 8, sizes[f]).astype(np.uint32)
        descs.append(d)
    pl = [(a, b) for a in range(F) for b in range(a + 1, F)]
    assert len(pl) == 561
    engine.set_match_chunk(2048)
    out = _match_dev(engine, descs, pl, 4736)
    for m in (0, 1, 255, 256, 511, 512, 560):
        a, b = pl[m]
        assert _same(out[m][:sizes[a]], cref.match_sorted(descs[a], descs[b])), (a, b)
