"""GPU tests of the device-resident batched entry points (pgx_detect_batch_dev / pgx_match_batch_dev):
several frames per launch (frame counts that are and are not multiples of 8, which exercises both branches
of the XCD-aware block mapping), ragged keypoint counts, `max_count` truncation, more image pairs than one
workspace chunk (128), and the profiling hooks.  Everything is compared with the CPU oracle."""
import numpy as np
import pytest
import torch

from oracle import cref
from photogrammetry_amd import synth
import photogrammetry_amd as pg

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def engine():
    """Own context: these tests change capacities, BRIEF pairs and the dewarp map."""
    e = pg.Engine(0)
    yield e
    e.close()


def _oracle_detect(frame, dmap, pairs, T, radius, cap):
    src = cref.apply_distortion(frame, dmap) if dmap is not None else frame
    g = cref.gray(src)
    raw = cref.detect(g, T)
    kept = raw[cref.nms(raw, radius)][:cap]
    return kept, cref.brief(g, np.stack([kept["x"], kept["y"]], 1), pairs), len(raw)


@pytest.mark.parametrize("F,radius,with_map", [(3, 9, True), (8, 16, False), (11, 20, True)])
def test_detect_batch_matches_oracle(engine, F, radius, with_map):
    W, H, CAP = 320, 240, 2048
    T = np.float32(0.1)
    frames = np.stack([synth.make_frame(W, H, seed=50 + i, n_shapes=40 + 37 * i) for i in range(F)])
    frames[F - 1][...] = synth.make_frame(W, H, seed=1, n_shapes=0)   # a frame with no keypoints at all
    pairs = pg.make_brief_pairs(3, 30, 256)
    dmap = pg.build_dewarp_map(W, H, [3e-4, 1e-7, 0, 0, 0]) if with_map else None
    engine.set_brief_pairs(pairs)
    engine.set_detect_params(T, radius)
    engine.set_capacity(1 << 16, CAP)
    engine.set_dewarp_map(dmap)
    d_frames = torch.from_numpy(frames).to(DEV)
    d_kp = torch.zeros((F, CAP, 4), dtype=torch.int32, device=DEV)
    d_desc = torch.zeros((F, CAP, 8), dtype=torch.int32, device=DEV)
    d_counts = torch.full((F,), -1, dtype=torch.int32, device=DEV)
    d_nraw = torch.full((F,), -1, dtype=torch.int32, device=DEV)
    torch.cuda.synchronize()   # torch's fills run on ITS stream; the engine's non-blocking stream does not order against it
    engine.detect_batch_dev(d_frames, F, W, H, d_kp, d_desc, d_counts, d_nraw, CAP)
    engine.check_status()
    kp = d_kp.cpu().numpy()
    desc = d_desc.cpu().numpy().view(np.uint32)
    counts, nraw = d_counts.cpu().numpy(), d_nraw.cpu().numpy()
    assert counts[F - 1] == 0 and nraw[F - 1] == 0
    for f in range(F):
        kept, edesc, n_raw = _oracle_detect(frames[f], dmap, pairs, T, radius, CAP)
        assert nraw[f] == n_raw and counts[f] == len(kept), f
        n = len(kept)
        assert (kp[f, :n, 0] == kept["x"]).all() and (kp[f, :n, 1] == kept["y"]).all()
        assert (kp[f, :n, 2] == kept["fast_score"]).all()
        assert kp[f, :n, 3].view(np.float32).tobytes() == kept["value"].tobytes()
        assert (desc[f, :n] == edesc).all()
    engine.set_dewarp_map(None)


def _match_batch(engine, descs, counts, pairlist, stride, max_count=None):
    F = len(descs)
    d = np.zeros((F, stride, 8), dtype=np.uint32)
    for f in range(F):
        d[f, :len(descs[f])] = descs[f]
    d_desc = torch.from_numpy(d.view(np.int32)).to(DEV)
    d_counts = torch.tensor(counts, dtype=torch.int32, device=DEV)
    d_pl = torch.tensor(pairlist, dtype=torch.int32, device=DEV)
    d_out = torch.full((len(pairlist), stride, 3), -7, dtype=torch.int32, device=DEV)
    torch.cuda.synchronize()   # see pgx.h: "_dev" buffers must be ready on, or ordered against, the context's stream
    engine.match_batch_dev(d_desc, d_counts, stride, 8, d_pl, len(pairlist), d_out, max_count=max_count)
    _match_batch.keepalive = (d_desc, d_counts, d_pl)   # the launch is asynchronous: inputs must outlive it
    return d_out


def test_match_batch_ragged_counts_and_order(engine):
    rng = np.random.default_rng(5)
    sizes = [1500, 40, 2300, 1024, 1025, 0, 700]
    descs = [rng.integers(0, 2**32, (n, 8), dtype=np.uint32) for n in sizes]
    descs[3] = descs[0][:1024].copy()                      # frame 3 is a prefix of frame 0: exact matches exist
    pl = [(0, 2), (2, 0), (1, 4), (4, 1), (3, 0), (0, 3), (5, 6), (6, 6), (1, 1)]
    d_out = _match_batch(engine, descs, sizes, pl, 2304)
    engine.check_status()          # synchronises the engine's stream; results are not defined before it
    out = d_out.cpu().numpy()
    for m, (a, b) in enumerate(pl):
        if sizes[a] == 0:
            continue
        exp = cref.match_sorted(descs[a], descs[b])
        got = out[m][:sizes[a]]
        assert (got[:, 0] == exp["k1"]).all() and (got[:, 1] == exp["k2"]).all() and (got[:, 2] == exp["dist"]).all(), (a, b)


def test_match_batch_host_buffers(engine):
    """pgx_match_batch: many image pairs from HOST arrays in one call (SURVEY 8b "Call sites") -- the same lists as one
    pgx_match per pair, ragged sizes, an empty first set, and the reference's exception for an empty second set."""
    rng = np.random.default_rng(11)
    sizes = [900, 37, 1300, 0, 512]
    descs = [rng.integers(0, 2**32, (n, 8), dtype=np.uint32) for n in sizes]
    descs[4] = descs[0][:512].copy()
    pl = [(0, 2), (2, 0), (1, 4), (4, 0), (3, 1), (1, 1)]
    lists = engine.match_batch(descs, pl)
    assert [len(x) for x in lists] == [sizes[a] for a, _ in pl]
    for (a, b), got in zip(pl, lists):
        if sizes[a] == 0:
            continue
        exp = cref.match_sorted(descs[a], descs[b])
        assert (got["k1"] == exp["k1"]).all() and (got["k2"] == exp["k2"]).all() and (got["dist"] == exp["dist"]).all(), (a, b)
        one = engine.match(descs[a], descs[b])
        assert (one == got).all()
    with pytest.raises(pg.ArgumentOutOfRangeException):
        engine.match_batch(descs, [(0, 2), (0, 3)])     # keypoints2 empty, keypoints1 not: KeypointMatching.cs:61
    got = engine.last_batch_lists                       # every list is still there
    assert (got[0] == lists[0]).all() and len(got[1]) == sizes[0] and (got[1]["dist"] == cref.INT_MAX).all()
    assert engine.match_batch(descs, []) == []


@pytest.mark.parametrize("words,mask", [(3, 0xFFFFFFFF), (1, 0x1F), (16, 0xFFFFFFFF)])
def test_match_batch_host_buffers_other_descriptor_lengths(engine, words, mask):
    """pgx_match_batch away from 256 bits (the matcher's generic path): 96-, 32- (tie-heavy) and 512-bit descriptors."""
    rng = np.random.default_rng(words)
    sizes = [300, 41, 520]
    descs = [rng.integers(0, 2**32, (n, words), dtype=np.uint32) & np.uint32(mask) for n in sizes]
    pl = [(0, 2), (2, 0), (1, 0), (2, 2)]
    lists = engine.match_batch(descs, pl)
    for (a, b), got in zip(pl, lists):
        exp = cref.match_sorted(descs[a], descs[b])
        assert (got["k1"] == exp["k1"]).all() and (got["k2"] == exp["k2"]).all() and (got["dist"] == exp["dist"]).all(), (a, b)


def test_match_batch_empty_second_set_raises(engine):
    rng = np.random.default_rng(6)
    descs = [rng.integers(0, 2**32, (30, 8), dtype=np.uint32), np.zeros((0, 8), np.uint32)]
    out = _match_batch(engine, descs, [30, 0], [(0, 1), (0, 0)], 64)
    with pytest.raises(pg.ArgumentOutOfRangeException):     # KeypointMatching.cs:61
        engine.check_status()
    o = out.cpu().numpy()
    assert (o[0, :30, 2] == pg.api.PGX_DIST_NONE).all()     # the failed pair is all (0,0,int.MaxValue)
    exp = cref.match(descs[0], descs[0])
    assert (o[1, :30, 1] == exp["k2"]).all()                # the other pair of the batch is unaffected


def test_match_batch_max_count_truncates(engine):
    rng = np.random.default_rng(7)
    descs = [rng.integers(0, 2**32, (900, 8), dtype=np.uint32), rng.integers(0, 2**32, (1300, 8), dtype=np.uint32)]
    d_out = _match_batch(engine, descs, [900, 1300], [(0, 1), (1, 0)], 1536, max_count=512)
    engine.check_status()
    out = d_out.cpu().numpy()
    for m, (a, b) in enumerate([(0, 1), (1, 0)]):
        exp = cref.match_sorted(descs[a][:512], descs[b][:512])   # lists are cut to their first max_count entries
        got = out[m][:512]
        assert (got[:, 0] == exp["k1"]).all() and (got[:, 1] == exp["k2"]).all() and (got[:, 2] == exp["dist"]).all()


def test_match_batch_more_pairs_than_one_chunk(engine):
    """300 image pairs > the 128-pair workspace chunk; every pair checked against the oracle."""
    rng = np.random.default_rng(8)
    F = 25
    sizes = [int(x) for x in rng.integers(50, 400, F)]
    base = rng.integers(0, 2**32, (400, 8), dtype=np.uint32)
    descs = []
    for f in range(F):
        d = base[rng.permutation(400)[:sizes[f]]].copy()
        d[:, 7] ^= rng.integers(0, 4, sizes[f]).astype(np.uint32)   # near-duplicates across frames: tie-heavy
        descs.append(d)
    pl = [(a, b) for a in range(F) for b in range(a + 1, F)]
    assert len(pl) == 300
    engine.set_match_chunk(128)
    d_out = _match_batch(engine, descs, sizes, pl, 400)
    engine.check_status()
    engine.set_match_chunk(2048)
    out = d_out.cpu().numpy()
    for m, (a, b) in enumerate(pl):
        exp = cref.match_sorted(descs[a], descs[b])
        got = out[m][:sizes[a]]
        assert (got[:, 0] == exp["k1"]).all() and (got[:, 1] == exp["k2"]).all() and (got[:, 2] == exp["dist"]).all(), (a, b)


def test_profile_hooks_and_stats(engine):
    rng = np.random.default_rng(9)
    descs = [rng.integers(0, 2**32, (3000, 8), dtype=np.uint32) for _ in range(2)]   # > PGX_TAIL_MAX: a wide round runs
    engine.profile_reset()
    engine.profile_enable(True)
    _match_batch(engine, descs, [3000, 3000], [(0, 1)], 3072)
    engine.check_status()
    engine.profile_enable(False)
    n, ms = engine.profile_get("ham_argmin")
    assert n >= 1 and ms > 0
    n2, ms2 = engine.profile_get("match_finish")
    assert n2 == 1 and ms2 > 0
    rounds, evals, ev0 = engine.match_stats()
    assert rounds >= 1 and ev0 == 3000 * 3000 and evals >= ev0
    assert engine.profile_get("tail_rows")[0] == 1
    engine.profile_reset()
    assert engine.profile_get("ham_argmin") == (0, 0.0)


@pytest.mark.parametrize("radius", [16, 20, 25, 45, 100])
def test_detect_full_size_frames_vs_oracle(engine, radius):
    """The bench workload's frame shape: 1920x1080, ~1e5 raw hits per frame, dewarp map on, two frames per launch
    (one of them a shifted copy), every stage compared bit-exactly with the oracle.  The radii cover every cell
    size / reach combination of the NMS champion rounds (8 px: reach 2 and 3; 16 px; 32 px; 64 px)."""
    W, H, CAP, RAW = 1920, 1080, 16384, 1 << 18
    T = np.float32(0.1)
    base = synth.make_frame(W, H, seed=11, n_shapes=12000)
    frames = np.stack([base, synth.shift_frame(base, 37, 11)])
    pairs = pg.make_brief_pairs(0, 50, 256)
    dmap = pg.build_dewarp_map(W, H, [3e-4, 1e-7, 0, 0, 0])
    engine.set_brief_pairs(pairs)
    engine.set_detect_params(T, radius)
    engine.set_capacity(RAW, CAP)
    engine.set_dewarp_map(dmap)
    F = 2
    d_frames = torch.from_numpy(frames).to(DEV)
    d_kp = torch.zeros((F, CAP, 4), dtype=torch.int32, device=DEV)
    d_desc = torch.zeros((F, CAP, 8), dtype=torch.int32, device=DEV)
    d_counts = torch.zeros(F, dtype=torch.int32, device=DEV)
    d_nraw = torch.zeros(F, dtype=torch.int32, device=DEV)
    torch.cuda.synchronize()
    engine.detect_batch_dev(d_frames, F, W, H, d_kp, d_desc, d_counts, d_nraw, CAP)
    engine.check_status()
    kp = d_kp.cpu().numpy()
    desc = d_desc.cpu().numpy().view(np.uint32)
    counts, nraw = d_counts.cpu().numpy(), d_nraw.cpu().numpy()
    for f in range(F):
        kept, edesc, n_raw = _oracle_detect(frames[f], dmap, pairs, T, radius, CAP)
        assert n_raw > 30000 and nraw[f] == n_raw
        assert counts[f] == len(kept), (f, counts[f], len(kept))
        n = len(kept)
        assert (kp[f, :n, 0] == kept["x"]).all() and (kp[f, :n, 1] == kept["y"]).all()
        assert (kp[f, :n, 2] == kept["fast_score"]).all()
        assert kp[f, :n, 3].view(np.float32).tobytes() == kept["value"].tobytes()
        assert (desc[f, :n] == edesc).all()
    engine.set_dewarp_map(None)


def test_detect_4k_frame_vs_oracle(engine):
    """SURVEY 8d config 4's frame shape: 3840x2160, r = 22 (16-px cells), > 8192 survivors, ~3e5 raw hits."""
    W, H, CAP, RAW, radius = 3840, 2160, 16384, 1 << 20, 22
    T = np.float32(0.1)
    frame = synth.make_frame(W, H, seed=5, n_shapes=80000)
    pairs = pg.make_brief_pairs(0, 50, 256)
    dmap = pg.build_dewarp_map(W, H, [3e-4, 1e-7, 0, 0, 0])
    engine.set_brief_pairs(pairs)
    engine.set_detect_params(T, radius)
    engine.set_capacity(RAW, CAP)
    engine.set_dewarp_map(dmap)
    d_frames = torch.from_numpy(frame[None]).to(DEV)
    d_kp = torch.zeros((1, CAP, 4), dtype=torch.int32, device=DEV)
    d_desc = torch.zeros((1, CAP, 8), dtype=torch.int32, device=DEV)
    d_counts = torch.zeros(1, dtype=torch.int32, device=DEV)
    d_nraw = torch.zeros(1, dtype=torch.int32, device=DEV)
    torch.cuda.synchronize()
    engine.detect_batch_dev(d_frames, 1, W, H, d_kp, d_desc, d_counts, d_nraw, CAP)
    engine.check_status()
    kept, edesc, n_raw = _oracle_detect(frame, dmap, pairs, T, radius, CAP)
    n = len(kept)
    assert n > 8192 and int(d_nraw[0]) == n_raw and int(d_counts[0]) == n
    kp = d_kp.cpu().numpy()[0]
    assert (kp[:n, 0] == kept["x"]).all() and (kp[:n, 1] == kept["y"]).all() and (kp[:n, 2] == kept["fast_score"]).all()
    assert kp[:n, 3].view(np.float32).tobytes() == kept["value"].tobytes()
    assert (d_desc.cpu().numpy().view(np.uint32)[0, :n] == edesc).all()
    engine.set_dewarp_map(None)


@pytest.mark.parametrize("radius", [10, 16, 21, 33])
def test_detect_dense_noise_frame_vs_oracle(engine, radius):
    """White-noise frame: a large share of the pixels are FAST hits (cells of the NMS grid hold dozens of
    records, neighbourhoods need several passes, every score level is populated)."""
    W, H, CAP, RAW = 640, 480, 8192, 1 << 19
    T = np.float32(0.1)
    rng = np.random.default_rng(77)
    frame = rng.integers(0, 65536, (H, W, 4), dtype=np.uint16)
    pairs = pg.make_brief_pairs(1, 50, 256)
    engine.set_brief_pairs(pairs)
    engine.set_detect_params(T, radius)
    engine.set_capacity(RAW, CAP)
    engine.set_dewarp_map(None)
    d_frames = torch.from_numpy(frame[None]).to(DEV)
    d_kp = torch.zeros((1, CAP, 4), dtype=torch.int32, device=DEV)
    d_desc = torch.zeros((1, CAP, 8), dtype=torch.int32, device=DEV)
    d_counts = torch.zeros(1, dtype=torch.int32, device=DEV)
    d_nraw = torch.zeros(1, dtype=torch.int32, device=DEV)
    torch.cuda.synchronize()
    engine.detect_batch_dev(d_frames, 1, W, H, d_kp, d_desc, d_counts, d_nraw, CAP)
    engine.check_status()
    kept, edesc, n_raw = _oracle_detect(frame, None, pairs, T, radius, CAP)
    n = len(kept)
    assert n_raw > 20000 and int(d_nraw[0]) == n_raw and int(d_counts[0]) == n
    kp = d_kp.cpu().numpy()[0]
    assert (kp[:n, 0] == kept["x"]).all() and (kp[:n, 1] == kept["y"]).all() and (kp[:n, 2] == kept["fast_score"]).all()
    assert (d_desc.cpu().numpy().view(np.uint32)[0, :n] == edesc).all()


def test_two_contexts_in_flight_are_independent():
    """Two contexts (two HIP streams) with different parameters enqueue detect + match back to back without any
    synchronisation in between; both results must equal the oracle (contexts share nothing but the device)."""
    W, H, CAP = 640, 360, 4096
    T = np.float32(0.1)
    frames = np.stack([synth.make_frame(W, H, seed=21 + i, n_shapes=900) for i in range(4)])
    cfg = [dict(radius=16, pairs=pg.make_brief_pairs(5, 50, 256)), dict(radius=11, pairs=pg.make_brief_pairs(6, 30, 256))]
    engs, outs = [], []
    d_frames = torch.from_numpy(frames).to(DEV)
    pl = torch.tensor([[0, 1], [2, 3], [3, 0]], dtype=torch.int32, device=DEV)
    for c in cfg:
        e = pg.Engine(0)
        e.set_brief_pairs(c["pairs"])
        e.set_detect_params(T, c["radius"])
        e.set_capacity(1 << 17, CAP)
        e.set_dewarp_map(None)
        engs.append(e)
        outs.append(dict(kp=torch.zeros((4, CAP, 4), dtype=torch.int32, device=DEV),
                         desc=torch.zeros((4, CAP, 8), dtype=torch.int32, device=DEV),
                         counts=torch.zeros(4, dtype=torch.int32, device=DEV),
                         nraw=torch.zeros(4, dtype=torch.int32, device=DEV),
                         out=torch.zeros((3, CAP, 3), dtype=torch.int32, device=DEV)))
    torch.cuda.synchronize()
    for _ in range(3):   # several rounds in flight on both streams
        for e, o in zip(engs, outs):
            e.detect_batch_dev(d_frames, 4, W, H, o["kp"], o["desc"], o["counts"], o["nraw"], CAP)
            e.match_batch_dev(o["desc"], o["counts"], CAP, 8, pl, 3, o["out"])
    for e in engs:
        e.check_status()
    for c, o in zip(cfg, outs):
        counts = o["counts"].cpu().numpy()
        desc = o["desc"].cpu().numpy().view(np.uint32)
        res = o["out"].cpu().numpy()
        exp = [_oracle_detect(frames[f], None, c["pairs"], T, c["radius"], CAP) for f in range(4)]
        for f in range(4):
            assert counts[f] == len(exp[f][0]) and (desc[f, :counts[f]] == exp[f][1]).all()
        for m, (a, b) in enumerate([(0, 1), (2, 3), (3, 0)]):
            em = cref.match_sorted(exp[a][1], exp[b][1])
            got = res[m][:counts[a]]
            assert (got[:, 0] == em["k1"]).all() and (got[:, 1] == em["k2"]).all() and (got[:, 2] == em["dist"]).all()
    for e in engs:
        e.close()


def test_match_batch_host_checks_shapes_and_ignores_unreferenced_frames(engine):
    """a narrower / 1-D descriptor array must be refused before the library copies counts[f] * words words from
    it; a large frame no pair references sizes nothing (slot size = the largest REFERENCED set) and changes no result."""
    a = synth.random_descriptors(300, 8, 1)
    b = synth.random_descriptors(280, 8, 2)
    with pytest.raises(ValueError):
        engine.match_batch([a, b[:, :4].copy()], [(0, 1)])
    with pytest.raises(ValueError):
        engine.match_batch([a, b.reshape(-1)], [(0, 1)])
    big = synth.random_descriptors(9000, 8, 3)   # never referenced
    got = engine.match_batch([a, big, b], [(0, 2), (2, 0)])
    for g, (x, y) in zip(got, ((a, b), (b, a))):
        e = cref.match_sorted(x, y)
        assert (g["k1"] == e["k1"]).all() and (g["k2"] == e["k2"]).all() and (g["dist"] == e["dist"]).all()
