"""GPU parity for 8-bit sources (SURVEY 8f-4 "ingest"; pgx_set_source_format): an RGBA8 frame handed to the library must
give exactly what the same frame gives after the x257 widening an 8-bit file gets when LocalImageReader loads it as
Rgba64 (LocalImageReader.cs:22) -- against the oracle on the widened frame, stage by stage and end to end."""
import numpy as np
import pytest
import torch

from conftest import pairs_arr, star_rgba64
from oracle import cref
import photogrammetry_amd as pg
from photogrammetry_amd import synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture()
def eng8(engine):
    engine.set_source_format(pg.api.PGX_SRC_RGBA8)
    yield engine
    engine.set_source_format(pg.api.PGX_SRC_RGBA64)
    engine.set_dewarp_map(None)


@pytest.mark.parametrize("W,H", [(96, 64), (101, 77), (451, 383)])      # 16-byte path, odd sizes (scalar path)
def test_gray_and_dewarp_from_rgba8(eng8, W, H):
    rng = np.random.default_rng(W * 1000 + H)
    a8 = rng.integers(0, 256, (H, W, 4), dtype=np.uint8)
    wide = a8.astype(np.uint16) * 257
    assert eng8.gray(a8).tobytes() == cref.gray(wide).tobytes()
    # a permutation-like map with repeats and the far corners
    uv = np.stack([rng.integers(0, W, (H, W)), rng.integers(0, H, (H, W))], axis=2).astype(np.int32)
    uv[0, 0] = (W - 1, H - 1)
    uv[-1, -1] = (0, 0)
    eng8.set_dewarp_map(uv)
    got = eng8.dewarp(a8)
    assert got.dtype == np.uint16 and (got == cref.apply_distortion(wide, uv)).all()
    uv[3, 5] = (W, 0)                                                       # out of range: IndexOutOfRange, not a clamp
    eng8.set_dewarp_map(uv)
    with pytest.raises(IndexError):
        eng8.dewarp(a8)


def test_unknown_source_format_is_rejected(engine):
    with pytest.raises(ValueError):
        engine.set_source_format(7)


def test_star_pair_from_8bit_pixels(eng8, star):
    """Config 1 with the PNGs' own 8-bit pixels (the committed golden lists come from the widened frames)."""
    eng8.set_dewarp_map(star["dewarp_map"].astype(np.int32))
    eng8.set_brief_pairs(star["brief_pairs"])
    eng8.set_detect_params(float(star["threshold"]), int(star["radius"]))
    eng8.set_capacity(1 << 16, 4096)
    descs = {}
    for tag in ("a", "b"):
        a8 = (star_rgba64(star, tag) // 257).astype(np.uint8)
        kp, desc, nraw = eng8.detect(a8, capacity=4096)
        assert nraw == int(star[tag + "_n_raw"])
        assert (np.stack([kp["x"], kp["y"], kp["fast_score"]], 1) == star[tag + "_kp"]).all()
        assert kp["value"].tobytes() == star[tag + "_value"].tobytes()
        assert (desc == star[tag + "_desc"]).all()
        descs[tag] = desc
    assert (pairs_arr(eng8.match(descs["a"], descs["b"])) == star["match_ab"]).all()


@pytest.mark.parametrize("W,H,F", [(640, 360, 6), (333, 77, 5)])
def test_detect_batch_from_rgba8_equals_widened(engine, W, H, F):
    """Device-resident batches: 8-bit frames (4 B/px) against the same frames widened on the host (8 B/px), with the
    shipped-coefficient dewarp table; every output buffer must be identical, and frame 0 is checked against the oracle."""
    CAP = 4096
    T = np.float32(0.1)
    rng = np.random.default_rng(F * W)
    base = synth.make_frame(W, H, seed=77, n_shapes=700)
    frames8 = np.stack([np.roll(base // 257, (i, 2 * i), axis=(0, 1)) for i in range(F)]).astype(np.uint8)
    frames8[..., 3] = rng.integers(0, 256, (F, H, W), dtype=np.uint8)     # alpha is carried but never used
    wide = frames8.astype(np.uint16) * 257
    dmap = pg.build_dewarp_map(W, H, [3e-4, 1e-7, 0, 0, 0])
    pairs = pg.make_brief_pairs(11, 40, 256)
    engine.set_brief_pairs(pairs)
    engine.set_detect_params(T, 13)
    engine.set_capacity(1 << 17, CAP)
    engine.set_dewarp_map(dmap)
    out = {}
    try:
        for name, host, fmt in (("wide", wide, pg.api.PGX_SRC_RGBA64), ("8bit", frames8, pg.api.PGX_SRC_RGBA8)):
            engine.set_source_format(fmt)
            d_frames = torch.from_numpy(host).to(DEV)
            d_kp = torch.zeros((F, CAP, 4), dtype=torch.int32, device=DEV)
            d_desc = torch.zeros((F, CAP, 8), dtype=torch.int32, device=DEV)
            d_cnt = torch.zeros((F,), dtype=torch.int32, device=DEV)
            d_nraw = torch.zeros((F,), dtype=torch.int32, device=DEV)
            engine.detect_batch_dev(d_frames.data_ptr(), F, W, H, d_kp.data_ptr(), d_desc.data_ptr(), d_cnt.data_ptr(),
                                    d_nraw.data_ptr(), CAP)
            engine.check_status()
            out[name] = [t.cpu().numpy() for t in (d_kp, d_desc, d_cnt, d_nraw)]
    finally:
        engine.set_source_format(pg.api.PGX_SRC_RGBA64)
        engine.set_dewarp_map(None)
    for a, b in zip(out["wide"], out["8bit"]):
        assert (a == b).all()
    g = cref.gray(cref.apply_distortion(wide[0], dmap))
    raw = cref.detect(g, T)
    kept = raw[cref.nms(raw, 13)]
    kp, desc, cnt, nraw = out["8bit"]
    assert cnt[0] == len(kept) and nraw[0] == len(raw) and cnt[0] > 50
    assert (kp[0, :cnt[0], 0] == kept["x"]).all() and (kp[0, :cnt[0], 1] == kept["y"]).all()
    assert (desc[0, :cnt[0]].view(np.uint32) == cref.brief(g, np.stack([kept["x"], kept["y"]], 1), pairs)).all()
