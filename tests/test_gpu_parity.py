"""GPU parity tests: the HIP path (through the C ABI, libpgx.so) against the CPU oracle on the
same seeded inputs, and against the committed golden fixtures.  Bit-exact everywhere: integer
indices, integer Hamming distances, float32 grey values compared by bit pattern."""
import numpy as np
import pytest

from oracle import cref, oracle_np as onp
from photogrammetry_amd import synth
import photogrammetry_amd as pg

from conftest import check_against_dotnet_bmp, pairs_arr, star_rgba64

pytestmark = pytest.mark.gpu


def _kp_from_oracle(k):
    out = np.zeros(len(k), dtype=pg.KEYPOINT_DTYPE)
    for f in ("x", "y", "fast_score", "value"):
        out[f] = k[f]
    return out


# ---- a2/a3: dewarp + grayscale ---------------------------------------------------------------

@pytest.mark.parametrize("W,H", [(451, 383), (64, 16), (7, 5), (130, 33)])
def test_gray_bit_exact(engine, W, H):
    rng = np.random.default_rng(W * 1000 + H)
    rgba = rng.integers(0, 65536, size=(H, W, 4), dtype=np.uint16)
    got = engine.gray(rgba)
    exp = cref.gray(rgba)
    assert got.view(np.uint32).tobytes() == exp.view(np.uint32).tobytes()


def test_gray_extremes(engine):
    rgba = np.zeros((4, 8, 4), dtype=np.uint16)
    rgba[0, :, :3] = 65535
    rgba[1, :, 0] = 1
    rgba[2, :, 1] = 65535
    rgba[3, :, 3] = 65535  # alpha is ignored
    got = engine.gray(rgba)
    assert (got == cref.gray(rgba)).all()
    assert got[0, 0] == np.float32(1.0) and got[3, 0] == 0.0


@pytest.mark.parametrize("W,H", [(451, 383), (97, 61)])
def test_dewarp_matches_oracle(engine, W, H):
    rng = np.random.default_rng(5)
    rgba = rng.integers(0, 65536, size=(H, W, 4), dtype=np.uint16)
    m = pg.build_dewarp_map(W, H, [3e-4, 1e-7, 0, 0, 0])
    assert (m == cref.build_distortion_matrix(W, H, [3e-4, 1e-7, 0, 0, 0])).all()
    dw = pg.DeWarp(engine, W, H, [3e-4, 1e-7, 0, 0, 0])
    got = dw.ApplyDistortionMat(rgba, m)
    assert (got == cref.apply_distortion(rgba, m)).all()
    # random permutation map (every source in range), incl. the (ushort) wrap of negative ints
    perm = np.stack([rng.integers(0, W, (H, W)), rng.integers(0, H, (H, W))], axis=2).astype(np.int32)
    perm[0, 0] = (3 - 65536, 2 + 65536)   # wraps to (3, 2) exactly as the unchecked casts do
    got = dw.ApplyDistortionMat(rgba, perm)
    assert (got == cref.apply_distortion(rgba, perm)).all()
    engine.set_dewarp_map(None)


def test_dewarp_errors(engine):
    rgba = np.zeros((20, 30, 4), dtype=np.uint16)
    dw = pg.DeWarp(engine, 30, 20, [0, 0, 0, 0, 0])
    ident = np.stack(np.meshgrid(np.arange(30), np.arange(20)), axis=2).astype(np.int32)
    with pytest.raises(pg.ArgumentException):       # DeWarp.cs:22-23
        dw.ApplyDistortionMat(rgba, ident[:, :29])
    bad = ident.copy()
    bad[7, 7] = (-1, 3)                             # (ushort)-1 = 65535 -> out of range
    with pytest.raises(pg.IndexOutOfRangeException):
        dw.ApplyDistortionMat(rgba, bad)
    bad = ident.copy()
    bad[19, 29] = (29, 20)
    with pytest.raises(pg.IndexOutOfRangeException):
        dw.ApplyDistortionMat(rgba, bad)
    assert (dw.ApplyDistortionMat(rgba, ident) == rgba).all()
    with pytest.raises(pg.ArgumentException):       # DeWarp.cs:46-48
        pg.build_dewarp_map(30, 20, [1, 2, 3])
    engine.set_dewarp_map(None)


# ---- a4: FAST-like detector ---------------------------------------------------------------------

def _detect_cmp(engine, img, T):
    engine.set_detect_params(T, 0)
    got = engine.fast(img)
    exp = cref.detect(img, T)
    assert len(got) == len(exp)
    for f in ("x", "y", "fast_score"):
        assert (got[f] == exp[f]).all(), f
    assert got["value"].tobytes() == exp["value"].tobytes()
    return exp


def test_fast_xunit_known_answers(engine):
    """KeypointDetectionTests.cs:10-50 restated on the GPU path via the full detector."""
    m = np.zeros((7, 7), np.float32)
    m[0, 3] = m[3, 0] = m[6, 3] = m[3, 6] = 1
    engine.set_detect_params(0.5, 0)
    # centre (3,3): compass samples all differ; the ring has 12 zeros (similar) -> not a keypoint,
    # and the pre-test alone (which the xUnit test pins) is covered by the oracle test on CPU
    assert len(engine.fast(m)) == len(cref.detect(m, 0.5))
    const = np.full((7, 7), 0.5, np.float32)
    assert len(engine.fast(const)) == 0


@pytest.mark.parametrize("W,H,seed", [(83, 60, 0), (64, 16, 1), (65, 17, 2), (200, 131, 3), (451, 383, 4), (6, 6, 5),
                                      (7, 7, 6), (1, 1, 7), (70, 3, 8)])
def test_fast_random_quantised(engine, W, H, seed):
    rng = np.random.default_rng(seed)
    img = (rng.integers(0, 4, (H, W)) / 3).astype(np.float32)
    if H > 30 and W > 50:
        img[10:30, 20:50] = 1.0
    _detect_cmp(engine, img, np.float32(0.1))


def test_fast_float_threshold_edges(engine):
    """Values exactly at c-T / c+T and one ulp either side (float32 compare semantics)."""
    rng = np.random.default_rng(11)
    T = np.float32(0.1)
    base = rng.random((40, 64)).astype(np.float32)
    img = base.copy()
    # plant neighbours at the exact float32 thresholds of their centre
    for (y, x) in [(10, 10), (10, 30), (20, 20), (30, 40)]:
        c = img[y, x]
        lo, hi = np.float32(c - T), np.float32(c + T)
        img[y, x - 3] = lo
        img[y, x + 3] = hi
        img[y - 3, x] = np.nextafter(lo, np.float32(2))
        img[y + 3, x] = np.nextafter(hi, np.float32(-2))
    _detect_cmp(engine, img, T)
    _detect_cmp(engine, (base * 0.2).astype(np.float32), np.float32(0.05))


def test_fast_quirks(engine):
    """Hand-derived rings: duplicate offset (-3,+1), 4 vs 5 similar, wrap-around, score 16."""
    def ring_image(similar_idx, c=0.5, other=1.0):
        img = np.full((7, 7), c, np.float32)
        pts = onp.CIRCLE[:15]  # entry 15 duplicates entry 1
        for idx, (dx, dy) in enumerate(pts):
            img[3 + dy, 3 + dx] = c if idx in similar_idx else other
        return img
    T = np.float32(0.25)
    engine.set_detect_params(T, 0)
    cases = {
        "all different -> 16": (set(), 16),
        "entry 1 similar counts twice (idx 1 and 15)": ({1}, 13),
        "entry 14 similar only": ({14}, 15),
        "run wraps 15->0": ({7, 8}, 14),
        "4 similar ok": ({5, 6, 7, 8}, 12),
        "5 similar rejects": ({4, 5, 6, 7, 8}, None),
        "two compass similar rejects": ({0, 8}, None),
    }
    for name, (sim, score) in cases.items():
        img = ring_image(sim)
        exp = cref.intensity_if_keypoint(img, 3, 3, T)
        got = engine.fast(img)
        assert exp == score, (name, exp)
        if score is None:
            assert len(got) == 0, name
        else:
            assert len(got) == 1 and got[0]["fast_score"] == score and got[0]["x"] == 3 and got[0]["y"] == 3, name
    # pixel (-3,-1) must NOT matter: the reference never samples it
    a = ring_image({14})
    b = a.copy()
    b[2, 0] = 0.5
    assert engine.fast(a)[0]["fast_score"] == engine.fast(b)[0]["fast_score"] == 15
    # brighter and darker mixed still count as "different"
    img = ring_image(set())
    img[3 + 3, 3 + 0] = 0.0
    img[3 + 0, 3 + 3] = 0.0
    assert engine.fast(img)[0]["fast_score"] == 16


def test_fast_capacity_error(engine):
    rng = np.random.default_rng(3)
    img = (rng.integers(0, 4, (60, 83)) / 3).astype(np.float32)
    engine.set_detect_params(0.1, 0)
    n = len(cref.detect(img, np.float32(0.1)))
    assert n > 10
    with pytest.raises(pg.CapacityError):
        engine.fast(img, capacity=10)


# ---- a5: BRIEF ------------------------------------------------------------------------------------

@pytest.mark.parametrize("P", [256, 64, 70, 1, 33, 320])
def test_brief_matches_oracle(engine, P):
    rng = np.random.default_rng(P)
    H, W = 60, 83
    img = (rng.integers(0, 4, (H, W)) / 3).astype(np.float32)
    pairs = cref.gaussian_pairs(P, 10, P)
    pairs[::5] *= -1                      # negative offsets: OOB on the low side too
    pairs[1::7] = 0                       # equal points: bit 0 (v1 < v2 is false)
    engine.set_brief_pairs(pairs)
    kps = np.zeros(40, dtype=pg.KEYPOINT_DTYPE)
    kps["x"] = rng.integers(0, W, 40)
    kps["y"] = rng.integers(0, H, 40)
    kps[0]["x"], kps[0]["y"] = 0, 0
    kps[1]["x"], kps[1]["y"] = W - 1, H - 1
    got = engine.brief(img, kps)
    exp = cref.brief(img, np.stack([kps["x"], kps["y"]], 1), pairs)
    assert got.shape == exp.shape and (got == exp).all()


def test_brief_bit_order(engine):
    """Pair 0 lands on the most significant bit (bit P-1); an OOB first or second point gives 0."""
    img = np.zeros((8, 8), np.float32)
    img[4, 5] = 1.0
    pairs = np.array([[0, 0, 1, 0],      # K[4,4]=0 < K[4,5]=1 -> 1  (bit 3)
                      [1, 0, 0, 0],      # 1 < 0 false -> 0           (bit 2)
                      [-9, 0, 1, 0],     # first point OOB -> 0       (bit 1)
                      [0, 0, 1, 0]], np.int32)  # -> 1                (bit 0)
    engine.set_brief_pairs(pairs)
    kps = np.zeros(1, dtype=pg.KEYPOINT_DTYPE)
    kps["x"], kps["y"] = 4, 4
    assert engine.brief(img, kps)[0, 0] == 0b1001
    assert cref.brief(img, [[4, 4]], pairs)[0, 0] == 0b1001


# ---- a7: NMS --------------------------------------------------------------------------------------

@pytest.mark.parametrize("radius", [-1, 0, 1, 3, 5, 10, 50, 500])
def test_nms_matches_oracle(engine, radius):
    rng = np.random.default_rng(radius + 2)
    H, W = 120, 160
    img = (rng.integers(0, 4, (H, W)) / 3).astype(np.float32)
    raw = cref.detect(img, np.float32(0.1))
    assert len(raw) > 500
    elim = pg.RedundantKeypointEliminator(engine, radius)
    got = elim.EliminateRedundantKeypoints(_kp_from_oracle(raw), W, H)
    exp = cref.nms(raw, radius)
    assert len(got) == len(exp) and (got == exp).all()


def test_nms_ties_and_exact_radius(engine):
    """Score ties resolve in input order; distance == r is suppressed, just beyond is kept."""
    k = np.zeros(5, dtype=pg.KEYPOINT_DTYPE)
    k["x"] = [10, 13, 10, 20, 10]
    k["y"] = [10, 14, 15, 10, 16]
    k["fast_score"] = [12, 12, 12, 16, 12]
    elim = pg.RedundantKeypointEliminator(engine, 5)
    got = elim.EliminateRedundantKeypoints(k, 64, 64)
    # 3 (score 16) first; then 0; (13,14) is at distance exactly 5 from 0 -> dropped; (10,15) d=5 dropped;
    # (10,16) d=6 kept
    assert list(got) == [3, 0, 4]
    ok = np.zeros(5, dtype=cref.KP_DTYPE)
    for f in ("x", "y", "fast_score"):
        ok[f] = k[f]
    assert list(cref.nms(ok, 5)) == [3, 0, 4]
    assert len(elim.EliminateRedundantKeypoints(k[:0], 64, 64)) == 0


def test_nms_arbitrary_scores(engine):
    rng = np.random.default_rng(9)
    n = 700
    k = np.zeros(n, dtype=pg.KEYPOINT_DTYPE)
    flat = rng.permutation(200 * 150)[:n]
    k["x"], k["y"] = flat % 200, flat // 200
    k["fast_score"] = rng.integers(-50, 50, n)
    ok = np.zeros(n, dtype=cref.KP_DTYPE)
    for f in ("x", "y", "fast_score"):
        ok[f] = k[f]
    for r in (4, 17):
        got = pg.RedundantKeypointEliminator(engine, r).EliminateRedundantKeypoints(k, 200, 150)
        assert (got == cref.nms(ok, r)).all()


# ---- a8: matching ---------------------------------------------------------------------------------

@pytest.mark.parametrize("n1,n2,words", [(5, 5, 8), (40, 25, 8), (25, 40, 8), (60, 60, 1), (33, 47, 3), (300, 300, 8),
                                         (1, 1, 8), (1, 9, 8), (9, 1, 8), (257, 255, 8), (700, 513, 8), (130, 90, 2)])
def test_match_literal_oracle(engine, n1, n2, words):
    rng = np.random.default_rng(n1 * 7 + n2)
    d1 = rng.integers(0, 2**32, (n1, words), dtype=np.uint32)
    d2 = rng.integers(0, 2**32, (n2, words), dtype=np.uint32)
    if words == 1:
        d1 &= 0xF
        d2 &= 0xF       # tie-heavy
    got = pg.KeypointMatching(engine).MatchKeypoints(d1, d2)
    exp = cref.match(d1, d2)
    assert (pairs_arr(got) == pairs_arr(exp)).all()


def test_match_empty_sets(engine):
    d = np.zeros((4, 8), np.uint32)
    km = pg.KeypointMatching(engine)
    assert len(km.MatchKeypoints(d[:0], d)) == 0                 # N1 = 0 -> empty list
    with pytest.raises(pg.ArgumentOutOfRangeException):          # N2 = 0 < N1 -> throws (:61)
        km.MatchKeypoints(d, d[:0])
    assert len(km.MatchKeypoints(d[:0], d[:0])) == 0


def test_match_all_equal_distances(engine):
    """Every distance ties: the order is purely (k1, k2) ascending."""
    d1 = np.zeros((50, 8), np.uint32)
    d2 = np.zeros((70, 8), np.uint32)
    got = pairs_arr(pg.KeypointMatching(engine).MatchKeypoints(d1, d2))
    assert (got[:, 0] == np.arange(50)).all() and (got[:, 1] == np.arange(50)).all() and (got[:, 2] == 0).all()


@pytest.mark.parametrize("n1,n2,protos", [(1500, 1300, 40), (600, 900, 5), (3000, 3000, 700), (1025, 1024, 3)])
def test_match_duplicate_and_hub_descriptors(engine, n1, n2, protos):
    """Tie chains: many identical descriptors and low-popcount "hub" descriptors, which make the
    greedy accept one edge per cluster per round (exercises the incremental LDS tail and the
    wide-round skip logic on both sides of the 1024 boundary)."""
    rng = np.random.default_rng(n1 + protos)
    base = rng.integers(0, 2**32, (protos, 8), dtype=np.uint32)
    base[: max(1, protos // 4)] &= rng.integers(0, 2**32, (max(1, protos // 4), 8), dtype=np.uint32) & 0x11111111
    d1 = base[rng.integers(0, protos, n1)].copy()
    d2 = base[rng.integers(0, protos, n2)].copy()
    flip = rng.random(n1) < 0.3
    d1[flip, 0] ^= np.uint32(1) << rng.integers(0, 32, int(flip.sum())).astype(np.uint32)
    got = pairs_arr(pg.KeypointMatching(engine).MatchKeypoints(d1, d2))
    assert (got == pairs_arr(cref.match_sorted(d1, d2))).all()


def test_match_lego_golden(engine, lego):
    """The real descriptor sets of data/feature_matching_test (2175 x 1285, tie-heavy), both directions."""
    km = pg.KeypointMatching(engine)
    got = pairs_arr(km.MatchKeypoints(lego["left_desc"], lego["right_desc"]))
    assert (got == lego["match_lr"]).all()
    assert int((got[:, 2] == pg.api.PGX_DIST_NONE).sum()) == 890
    got = pairs_arr(km.MatchKeypoints(lego["right_desc"], lego["left_desc"]))
    assert (got == lego["match_rl"]).all()
    got = pairs_arr(km.MatchKeypoints(lego["left_desc"][:400], lego["right_desc"][:300]))
    assert (got == lego["match_lr_400x300_literal"]).all()


@pytest.mark.parametrize("n", [1024, 4096])
def test_match_large_vs_sorted_oracle(engine, n):
    d1 = synth.random_descriptors(n, 8, 100 + n)
    d2 = synth.random_descriptors(n, 8, 200 + n)
    got = pairs_arr(pg.KeypointMatching(engine).MatchKeypoints(d1, d2))
    assert (got == pairs_arr(cref.match_sorted(d1, d2))).all()
    a, b, perm = synth.true_match_descriptors(n, 8, 300 + n)
    got = pairs_arr(pg.KeypointMatching(engine).MatchKeypoints(a, b))
    assert (got == pairs_arr(cref.match_sorted(a, b))).all()
    # round trip property: every row is matched to its own noisy copy
    inv = np.empty(n, dtype=np.int64)
    inv[perm] = np.arange(n)
    assert (got[np.argsort(got[:, 0]), 1] == inv).all()


def test_match_properties_at_full_size(engine):
    """Size-independent properties at the bench size (N = 4096 random): exact Hamming distances,
    one-to-one, emission order sorted by (dist, k1), every row present."""
    n = 4096
    d1 = synth.random_descriptors(n, 8, 1)
    d2 = synth.random_descriptors(n, 8, 2)
    got = pairs_arr(pg.KeypointMatching(engine).MatchKeypoints(d1, d2))
    assert sorted(got[:, 0]) == list(range(n)) and sorted(got[:, 1]) == list(range(n))
    ham = np.array([sum(bin(int(x)).count("1") for x in (d1[a] ^ d2[b])) for a, b, _ in got[:512]])
    assert (ham == got[:512, 2]).all()
    key = got[:, 2] * (1 << 20) + got[:, 0]
    assert (np.diff(key) > 0).all()


# ---- end to end: config 1 (the star pair) --------------------------------------------------------

def test_star_pair_end_to_end(engine, star):
    W, H = 451, 383
    engine.set_dewarp_map(star["dewarp_map"].astype(np.int32))
    engine.set_brief_pairs(star["brief_pairs"])
    engine.set_detect_params(float(star["threshold"]), int(star["radius"]))
    engine.set_capacity(1 << 16, 4096)
    descs = {}
    for tag in ("a", "b"):
        kp, desc, nraw = engine.detect(star_rgba64(star, tag), capacity=4096)
        assert nraw == int(star[tag + "_n_raw"])
        exp = star[tag + "_kp"]
        assert len(kp) == len(exp)
        assert (np.stack([kp["x"], kp["y"], kp["fast_score"]], 1) == exp).all()
        assert kp["value"].tobytes() == star[tag + "_value"].tobytes()
        assert (desc == star[tag + "_desc"]).all()
        descs[tag] = desc
    got = pairs_arr(engine.match(descs["a"], descs["b"]))
    assert (got == star["match_ab"]).all()
    engine.set_dewarp_map(None)


def test_dotnet_bmp_through_the_abi(engine, star, dotnet_bmp):
    """The reference's one C#-produced detector output (tests/test_oracle.py::test_dotnet_bmp_pins_detector_locations),
    checked against the HIP path itself: pgx_gray + pgx_fast for the raw hits, the fused pgx_detect for the survivors of
    the older flow's parameters (Program.cs:116-149: T = 0.2, r = (int)(451 * 0.015) = 6, no dewarp)."""
    rgba = star_rgba64(star, "a")
    engine.set_dewarp_map(None)
    engine.set_brief_pairs(star["brief_pairs"])
    engine.set_detect_params(float(dotnet_bmp["threshold"]), int(dotnet_bmp["radius"]))
    engine.set_capacity(1 << 16, 4096)
    raw = engine.fast(engine.gray(rgba))
    kp, _, nraw = engine.detect(rgba, capacity=4096)
    assert nraw == len(raw)
    check_against_dotnet_bmp(dotnet_bmp, np.stack([raw["x"], raw["y"]], 1), np.stack([kp["x"], kp["y"]], 1))


@pytest.mark.parametrize("W,H,radius,with_map", [(320, 200, 8, True), (640, 360, 20, False), (451, 383, 50, True),
                                                 (451, 383, 16, False), (333, 77, 12, True), (70, 66, 10, False)])
def test_detect_synthetic_vs_oracle(engine, W, H, radius, with_map):
    """Fused dewarp->gray->FAST->NMS->BRIEF on a synthetic corner-rich frame vs the oracle chain."""
    frame = synth.make_frame(W, H, seed=W + H)
    pairs = pg.make_brief_pairs(7, 50, 256)
    assert (pairs == cref.gaussian_pairs(7, 50, 256)).all()
    T = np.float32(0.1)
    engine.set_brief_pairs(pairs)
    engine.set_detect_params(T, radius)
    engine.set_capacity(1 << 17, 8192)
    src = frame
    if with_map:
        m = pg.build_dewarp_map(W, H, [3e-4, 1e-7, 0, 0, 0])
        engine.set_dewarp_map(m)
        src = cref.apply_distortion(frame, m)
    else:
        engine.set_dewarp_map(None)
    kp, desc, nraw = engine.detect(frame, capacity=8192)
    g = cref.gray(src)
    raw = cref.detect(g, T)
    assert nraw == len(raw) and nraw > 100
    order = cref.nms(raw, radius)
    kept = raw[order]
    assert len(kp) == len(kept)
    for f in ("x", "y", "fast_score"):
        assert (kp[f] == kept[f]).all()
    assert kp["value"].tobytes() == kept["value"].tobytes()
    assert (desc == cref.brief(g, np.stack([kept["x"], kept["y"]], 1), pairs)).all()
    engine.set_dewarp_map(None)


@pytest.mark.parametrize("W,H,coeffs", [(451, 383, [3e-4, 1e-7, 0, 0, 0]), (1920, 1080, [3e-4, 1e-7, 0, 0, 0]),
                                        (640, 480, [1e-4, 2e-7, 1e-3, 0, 0]), (97, 61, [1e-3, 5e-7, 0, 0, 0])])
def test_dewarp_map_built_on_device(engine, W, H, coeffs):
    """pgx_set_dewarp_coeffs = DeWarp.GetDistortionMatrix on the device.  Same float64 formulas as the host
    builder; the device libm differs in the last ulp, so a truncated coordinate may differ by one at isolated
    pixels ("+-1 px, unpinned")."""
    host = pg.build_dewarp_map(W, H, coeffs)
    assert (host == cref.build_distortion_matrix(W, H, coeffs)).all()
    engine.set_dewarp_coeffs(W, H, coeffs)
    dev = engine.get_dewarp_map(W, H)
    diff = np.abs(dev.astype(np.int64) - host.astype(np.int64))
    assert diff.max() <= 1
    assert (diff != 0).any(axis=2).mean() < 2e-3
    # the device-built table drives the dewarp like an uploaded one
    frame = synth.make_frame(W, H, seed=9)
    if dev.min() >= 0 and (dev[..., 0] < W).all() and (dev[..., 1] < H).all():
        assert (engine.dewarp(frame) == cref.apply_distortion(frame, dev)).all()
    with pytest.raises(pg.ArgumentException):
        engine.set_dewarp_coeffs(W, H, [1.0, 2.0, 3.0])          # DeWarp.cs:46-48
    engine.set_dewarp_map(None)


def test_nms_stage_api_full_size_list(engine):
    """pgx_nms on the raw list of a whole 1920x1080 frame (1.5e5 points) at the reference's shipped radius: long
    dependency chains (dozens of rounds).  Exact, and finished by whole-chip rounds rather than by the serial
    tail (which took seconds here): the time bound is loose but catches that regression."""
    import time
    W, H = 1920, 1080
    g = cref.gray(synth.make_frame(W, H, seed=3, n_shapes=20000))
    raw = cref.detect(g, np.float32(0.1))
    assert len(raw) > 100000
    for radius in (50, 23):
        engine.set_detect_params(0.1, radius)
        engine.nms(raw, W, H)
        t0 = time.perf_counter()
        order = engine.nms(raw, W, H)
        dt = time.perf_counter() - t0
        assert (order == cref.nms(raw, radius)).all()
        assert dt < 0.25, dt
