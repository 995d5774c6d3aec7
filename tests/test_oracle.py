"""CPU tests (no GPU): pin the oracle.

  * the reference's own known answers (ImageProcessing.Tests/KeypointDetectionTests.cs:10-50,
    LinearAlgebra.Tests/MatrixTests.cs:41-73 addressing semantics);
  * the committed golden fixtures (tests/golden/*.npz, produced by tests/golden/make_golden.py);
  * agreement of the literal C oracle with the independently written numpy twin on seeded inputs,
    including the parallel-rounds formulations the GPU uses;
  * hand-derived quirk cases.
"""
import numpy as np
import pytest

from oracle import cref, oracle_np as onp

from conftest import check_against_dotnet_bmp, pairs_arr, star_rgba64


# ---- reference known answers ----------------------------------------------------------------------

def _candidate_matrix():
    # Matrix<Grayscale>(7x7, K=0) with [3,0]=[0,3]=[3,6]=[6,3]=1 in (x, y) order
    m = np.zeros((7, 7), np.float32)          # numpy [y][x]
    for (x, y) in [(3, 0), (0, 3), (3, 6), (6, 3)]:
        m[y, x] = 1
    return m


def test_xunit_dim_center_is_detected():
    """KeypointDetectionTests.cs:10-27: IsPotentialKeypoint(candidateMatrix, 0, 3, 3) == true, T = 0.5."""
    assert cref.is_potential_keypoint(_candidate_matrix(), 0.0, 3, 3, 0.5) is True


def test_xunit_consistent_brightness_not_detected():
    """KeypointDetectionTests.cs:29-39."""
    m = np.full((7, 7), 0.5, np.float32)
    assert cref.is_potential_keypoint(m, 0.5, 3, 3, 0.5) is False


def test_xunit_constant_brightness_no_intensity_value():
    """KeypointDetectionTests.cs:41-50."""
    m = np.full((7, 7), 0.5, np.float32)
    assert cref.intensity_if_keypoint(m, 3, 3, 0.5) is None


def test_matrix_indexer_is_x_then_y():
    """MatrixTests.cs:60-73: m[x=0, y=1] = 2 -> transposed[1, 0] = 2.  Here: the compass offsets are
    added as (dx -> x, dy -> y); an asymmetric image tells a swapped table apart."""
    m = np.zeros((7, 7), np.float32)
    m[3, 0] = 1   # (x=0, y=3) = offset (-3, 0), the first compass sample
    # only one "different" compass sample, three similar -> rejected on the second similar
    assert cref.is_potential_keypoint(m, 0.0, 3, 3, 0.5) is False
    # out-of-range access behaves like Matrix.Get (IndexOutOfRangeException)
    with pytest.raises(cref.OracleError):
        cref.is_potential_keypoint(m, 0.0, 1, 3, 0.5)


# ---- FAST quirks ------------------------------------------------------------------------------------

def _ring_image(similar_idx, c=0.5, other=1.0):
    img = np.full((7, 7), c, np.float32)
    for idx, (dx, dy) in enumerate(onp.CIRCLE[:15]):
        img[3 + dy, 3 + dx] = c if idx in similar_idx else other
    return img


@pytest.mark.parametrize("similar,score", [
    (set(), 16), ({1}, 13), ({14}, 15), ({7, 8}, 14), ({5, 6, 7, 8}, 12), ({4, 5, 6, 7, 8}, None),
    ({0, 8}, None), ({2, 3}, 14), ({13}, 15), ({2, 9}, None), ({15 - 15}, 15)])
def test_fast_hand_derived(similar, score):
    img = _ring_image(similar)
    assert cref.intensity_if_keypoint(img, 3, 3, np.float32(0.25)) == score
    tw = onp.fast_scores(img, np.float32(0.25))[3, 3]
    assert (tw or None) == score


def test_fast_duplicate_offset_bug_is_reproduced():
    """Ring entry 15 is (-3,+1) again (KeypointDetection.cs:18): pixel (-3,-1) is never sampled."""
    a = _ring_image({14})
    b = a.copy()
    b[3 - 1, 3 - 3] = 0.5          # would be "similar" for a textbook FAST ring
    assert cref.intensity_if_keypoint(a, 3, 3, np.float32(0.25)) == 15
    assert cref.intensity_if_keypoint(b, 3, 3, np.float32(0.25)) == 15


@pytest.mark.parametrize("seed", range(6))
def test_fast_c_vs_twin(seed):
    rng = np.random.default_rng(seed)
    H, W = 50 + seed * 3, 70 + seed * 5
    img = (rng.integers(0, 4, (H, W)) / 3).astype(np.float32)
    kp = cref.detect(img, np.float32(0.1))
    tw = onp.detect(img, np.float32(0.1))
    assert len(kp) == len(tw) > 0
    assert (kp["x"] == tw[:, 0]).all() and (kp["y"] == tw[:, 1]).all() and (kp["fast_score"] == tw[:, 2]).all()
    assert (np.diff(kp["y"] * 100000 + kp["x"]) > 0).all()   # raster order


def test_detect_small_images_have_no_keypoints():
    for (h, w) in [(1, 1), (6, 6), (3, 70), (70, 6)]:
        assert len(cref.detect(np.random.default_rng(0).random((h, w)).astype(np.float32), np.float32(0.01))) == 0


# ---- gray / dewarp ------------------------------------------------------------------------------------

def test_gray_formula():
    rgba = np.array([[[65535, 65535, 65535, 0], [0, 0, 0, 65535], [1, 2, 3, 9], [65535, 0, 0, 0]]], np.uint16)
    g = cref.gray(rgba)
    assert g[0, 0] == np.float32(1.0) and g[0, 1] == 0.0
    assert g[0, 2] == np.float32(6.0) / np.float32(196605.0)
    assert (g == onp.gray(rgba)).all()
    rng = np.random.default_rng(0)
    big = rng.integers(0, 65536, (64, 64, 4), dtype=np.uint16)
    assert cref.gray(big).tobytes() == onp.gray(big).tobytes()


def test_dewarp_semantics():
    rng = np.random.default_rng(1)
    rgba = rng.integers(0, 65536, (20, 30, 4), dtype=np.uint16)
    ident = np.stack(np.meshgrid(np.arange(30), np.arange(20)), axis=2).astype(np.int32)
    assert (cref.apply_distortion(rgba, ident) == rgba).all()
    with pytest.raises(cref.OracleError) as e:
        cref.apply_distortion(rgba, ident[:, :29])
    assert e.value.code == cref.ORC_E_DIM
    bad = ident.copy()
    bad[5, 5] = (-1, 0)
    with pytest.raises(cref.OracleError) as e:
        cref.apply_distortion(rgba, bad)
    assert e.value.code == cref.ORC_E_OOB
    wrap = ident.copy()
    wrap[0, 0] = (4 - 65536, 7 + 65536)      # unchecked (ushort) casts wrap mod 65536
    assert (cref.apply_distortion(rgba, wrap)[0, 0] == rgba[7, 4]).all()
    m = cref.build_distortion_matrix(451, 383, [3e-4, 1e-7, 0, 0, 0])
    assert m[..., 0].min() >= 0 and m[..., 0].max() < 451 and m[..., 1].min() >= 0 and m[..., 1].max() < 383
    assert tuple(m[191, 225]) == (225, 191)   # centre maps to itself
    with pytest.raises(cref.OracleError):
        cref.build_distortion_matrix(10, 10, [1, 2, 3])


# ---- BRIEF ----------------------------------------------------------------------------------------------

def test_brief_bit_order_and_oob():
    img = np.zeros((8, 8), np.float32)
    img[4, 5] = 1.0
    pairs = np.array([[0, 0, 1, 0], [1, 0, 0, 0], [-9, 0, 1, 0], [0, 0, 1, 0]], np.int32)
    assert cref.brief(img, [[4, 4]], pairs)[0, 0] == 0b1001
    assert onp.brief(img, [[4, 4]], pairs)[0, 0] == 0b1001
    # second point OOB -> 0 as well
    pairs2 = np.array([[0, 0, 9, 0]], np.int32)
    assert cref.brief(img, [[4, 4]], pairs2)[0, 0] == 0


@pytest.mark.parametrize("P", [1, 31, 32, 33, 64, 70, 256, 320])
def test_brief_c_vs_twin(P):
    rng = np.random.default_rng(P)
    img = rng.random((40, 50)).astype(np.float32)
    pairs = cref.gaussian_pairs(P, 12, P)
    pairs[::3] *= -1
    xy = np.stack([rng.integers(0, 50, 20), rng.integers(0, 40, 20)], 1)
    assert (cref.brief(img, xy, pairs) == onp.brief(img, xy, pairs)).all()


def test_gaussian_pairs_shape_of_reference():
    """Utils.cs:19-38 draws y1,y2 in [0,1): offsets are never negative."""
    p = cref.gaussian_pairs(0, 50, 256)
    assert p.shape == (256, 4) and (p >= 0).all() and p.max() > 50
    assert (p == cref.gaussian_pairs(0, 50, 256)).all() and (p != cref.gaussian_pairs(1, 50, 256)).any()


# ---- NMS --------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("radius", [-1, 0, 1, 2, 5, 9, 30, 1000])
def test_nms_literal_vs_rounds(radius):
    rng = np.random.default_rng(radius + 10)
    img = (rng.integers(0, 4, (70, 90)) / 3).astype(np.float32)
    raw = cref.detect(img, np.float32(0.1))
    o1 = cref.nms(raw, radius)
    o2, _ = onp.nms_rounds(np.stack([raw["x"], raw["y"]], 1), raw["fast_score"], radius)
    assert (o1 == o2).all()
    if radius < 0:
        assert len(o1) == len(raw)


def test_nms_hand_derived():
    k = np.zeros(5, dtype=cref.KP_DTYPE)
    k["x"] = [10, 13, 10, 20, 10]
    k["y"] = [10, 14, 15, 10, 16]
    k["fast_score"] = [12, 12, 12, 16, 12]
    assert list(cref.nms(k, 5)) == [3, 0, 4]     # d == r is suppressed (keep iff distance > r)
    assert list(cref.nms(k, 4)) == [3, 0, 1]     # (10,15) and (10,16) are 3.16 / 3.61 from (13,14) -> dropped
    assert list(cref.nms(k[:0], 5)) == []


# ---- matching ----------------------------------------------------------------------------------------------

@pytest.mark.parametrize("n1,n2,words", [(0, 5, 8), (5, 5, 8), (40, 25, 8), (25, 40, 8), (60, 60, 1), (33, 47, 3),
                                         (1, 1, 8), (90, 90, 2)])
def test_match_four_formulations_agree(n1, n2, words):
    rng = np.random.default_rng(n1 * 31 + n2)
    d1 = rng.integers(0, 2**32, (n1, words), dtype=np.uint32)
    d2 = rng.integers(0, 2**32, (n2, words), dtype=np.uint32)
    if words <= 2:
        d1 &= 0x1F
        d2 &= 0x1F
    a = pairs_arr(cref.match(d1, d2))
    assert (a == pairs_arr(cref.match_sorted(d1, d2))).all()
    assert (a == onp.match_rounds(d1, d2)[0]).all()
    assert (a == onp.match_literal(d1, d2)).all()


def test_match_tail_and_empty():
    d = np.arange(24, dtype=np.uint32).reshape(3, 8)
    out = pairs_arr(cref.match(d, d[:1]))
    assert out.shape == (3, 3) and (out[1:] == [0, 0, cref.INT_MAX]).all()   # KeypointMatching.cs:40-42,57-62
    with pytest.raises(cref.OracleError) as e:
        cref.match(d, d[:0])
    assert e.value.code == cref.ORC_E_EMPTY
    assert len(cref.match(d[:0], d)) == 0


def test_count_ones_is_hamming():
    rng = np.random.default_rng(0)
    d1 = rng.integers(0, 2**32, (7, 8), dtype=np.uint32)
    d2 = rng.integers(0, 2**32, (9, 8), dtype=np.uint32)
    D = onp.hamming_matrix(d1, d2)
    out = pairs_arr(cref.match(d1, d2))
    for k1, k2, dist in out:
        assert D[k1, k2] == dist


# ---- golden fixtures -----------------------------------------------------------------------------------------

def test_golden_lego(lego):
    assert lego["left_desc"].shape == (2175, 8) and lego["right_desc"].shape == (1285, 8)
    got = pairs_arr(cref.match_sorted(lego["left_desc"], lego["right_desc"]))
    assert (got == lego["match_lr"]).all()
    assert int((got[:, 2] == cref.INT_MAX).sum()) == 890
    got = pairs_arr(cref.match_sorted(lego["right_desc"], lego["left_desc"]))
    assert (got == lego["match_rl"]).all()
    got = pairs_arr(cref.match(lego["left_desc"][:400], lego["right_desc"][:300]))
    assert (got == lego["match_lr_400x300_literal"]).all()


def test_golden_star_pair(star):
    W, H = 451, 383
    dewarp = star["dewarp_map"].astype(np.int32)
    assert (dewarp == cref.build_distortion_matrix(W, H, [3e-4, 1e-7, 0, 0, 0])).all()
    pairs = star["brief_pairs"]
    assert (pairs == cref.gaussian_pairs(0, 50, 256)).all()
    descs = {}
    for tag in ("a", "b"):
        g = cref.gray(cref.apply_distortion(star_rgba64(star, tag), dewarp))
        raw = cref.detect(g, star["threshold"])
        assert len(raw) == int(star[tag + "_n_raw"])
        kept = raw[cref.nms(raw, int(star["radius"]))]
        assert (np.stack([kept["x"], kept["y"], kept["fast_score"]], 1) == star[tag + "_kp"]).all()
        descs[tag] = cref.brief(g, np.stack([kept["x"], kept["y"]], 1), pairs)
        assert (descs[tag] == star[tag + "_desc"]).all()
    assert (pairs_arr(cref.match(descs["a"], descs["b"])) == star["match_ab"]).all()
    # the second image is the first shifted right by 150 px: most matched keypoints agree
    m = star["match_ab"]
    ka, kb = star["a_kp"], star["b_kp"]
    good = [(kb[k2][0] - ka[k1][0], kb[k2][1] - ka[k1][1]) for k1, k2, d in m if d != cref.INT_MAX]
    assert len(good) == min(len(ka), len(kb))


def test_dotnet_bmp_pins_detector_locations(star, dotnet_bmp):
    """The reference holds ONE numeric output of its C# path: data/feature_detection_test/output/dotnet_keypoints_backup.bmp,
    15pt_star.png with a blue square at every keypoint of the older flow (Photogrammetry/Program.cs:116-149:
    KeypointDetection(0.2f, 50, 256), squares of half-width 5).  Its blue mask equals, pixel for pixel, the union of those
    squares at the oracle's raw FAST hits: 106 of the 126 hits own a pixel of the mask no other hit covers, and exactly one
    non-hit position could be added without changing it -- so a4 (incl. the duplicated ring offset and the either-direction
    threshold) is pinned at pixel level by a C# output, and a7's survivors at r = 6 sit one per blue component."""
    g = cref.gray(star_rgba64(star, "a"))           # K in {0, 1}: the result is the same for Grayscale.FromRgba's scale
    raw = cref.detect(g, dotnet_bmp["threshold"])
    kept = raw[cref.nms(raw, int(dotnet_bmp["radius"]))]
    check_against_dotnet_bmp(dotnet_bmp, np.stack([raw["x"], raw["y"]], 1), np.stack([kept["x"], kept["y"]], 1))
    # how tightly the mask pins the hits
    blue, half = dotnet_bmp["blue"], int(dotnet_bmp["square"])
    cover = np.zeros(blue.shape, dtype=np.int32)
    for x, y in zip(raw["x"], raw["y"]):
        cover[max(0, y - half):y + half, max(0, x - half):x + half] += 1
    own = sum(1 for x, y in zip(raw["x"], raw["y"]) if (cover[max(0, y - half):y + half, max(0, x - half):x + half] == 1).any())
    assert own == 106
