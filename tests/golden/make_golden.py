#!/usr/bin/env python3
"""Generates the committed fixtures under tests/golden/ (run in the build container only).

Inputs are DATA files of the reference (read-only, /root/reference/data/feature_matching_test):
  * lego_space_1_from_{left,right}_keypoints.dat -- Python pickles of the older prototype's
    KeyPoint{coord, moment, descriptor}.  They are read with pickletools.genops (an opcode
    scan that executes nothing from the file), never pickle.load.
  * 15pt_star.png / 15pt_star_shifted_150.png -- the image pair the reference's live host
    names (Photogrammetry/TestService.cs:51-53), decoded with Pillow.

Expected outputs come from the C oracle (oracle/pgx_oracle.c), cross-checked here against
the independent numpy twin (oracle/oracle_np.py) before anything is written.  The
reference's C# cannot run in this image, so these vectors pin OUR restatement, not the
C# binary: "parity unpinned" for BRIEF / NMS / matching / dewarp.
"""
import os
import pickletools
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))
from oracle import cref, oracle_np as onp  # noqa: E402

REF = "/root/reference/data/feature_matching_test"
INT_OPS = {"BININT", "BININT1", "BININT2"}


def read_keypoint_pickle(path):
    """Opcode scan: each KeyPoint holds `MARK int int APPENDS` (coord) and one LONG1 (descriptor)."""
    ops = [(o.name, arg) for o, arg, _ in pickletools.genops(open(path, "rb").read())]
    coords, descs = [], []
    for i, (name, arg) in enumerate(ops):
        if name == "MARK" and i + 3 < len(ops) and ops[i + 1][0] in INT_OPS and ops[i + 2][0] in INT_OPS \
                and ops[i + 3][0] == "APPENDS":
            coords.append((ops[i + 1][1], ops[i + 2][1]))
        elif name in ("LONG1", "LONG4"):
            descs.append(int(arg))
    assert len(coords) == len(descs), (len(coords), len(descs))
    words = np.zeros((len(descs), 8), dtype=np.uint32)
    for k, d in enumerate(descs):
        assert 0 <= d < (1 << 256)
        for w in range(8):
            words[k, w] = (d >> (32 * w)) & 0xFFFFFFFF
    return np.array(coords, dtype=np.int32), words


def pairs_array(p):
    return np.stack([p["k1"], p["k2"], p["dist"]], axis=1).astype(np.int32)


def make_lego():
    cl, dl = read_keypoint_pickle(os.path.join(REF, "lego_space_1_from_left_keypoints.dat"))
    cr, dr = read_keypoint_pickle(os.path.join(REF, "lego_space_1_from_right_keypoints.dat"))
    print("lego sets:", dl.shape, dr.shape)
    out = {"left_desc": dl, "left_xy": cl, "right_desc": dr, "right_xy": cr}
    for tag, a, b in (("lr", dl, dr), ("rl", dr, dl)):
        exp = pairs_array(cref.match_sorted(a, b))
        twin, rounds = onp.match_rounds(a, b)
        assert (exp == twin).all(), tag
        print(tag, "rounds", rounds, "tail", int((exp[:, 2] == cref.INT_MAX).sum()))
        out["match_" + tag] = exp
    # the literal Theta(N^3) loop on a prefix small enough to finish in seconds
    a, b = dl[:400], dr[:300]
    lit = pairs_array(cref.match(a, b))
    assert (lit == pairs_array(cref.match_sorted(a, b))).all()
    out["match_lr_400x300_literal"] = lit
    np.savez_compressed(os.path.join(HERE, "lego_descriptors.npz"), **out)


def load_star(name):
    from PIL import Image
    im = Image.open(os.path.join(REF, name))
    a = np.asarray(im.convert("RGBA"), dtype=np.uint8)
    return a


def make_star():
    """Config 1: the star pair through the whole chain with the shipped parameters
    (Photogrammetry/appsettings.json:7-27) and the seeded pair table (seed 0)."""
    a = load_star("15pt_star.png")
    b = load_star("15pt_star_shifted_150.png")
    assert a.shape == b.shape == (383, 451, 4)
    # both are binary images: every channel is 0 or 255 -> keep one packed bit plane per channel
    for im in (a, b):
        assert set(np.unique(im)) <= {0, 255}
    W, H = 451, 383
    dewarp = cref.build_distortion_matrix(W, H, [3e-4, 1e-7, 0, 0, 0])
    pairs = cref.gaussian_pairs(0, 50, 256)
    T, radius = np.float32(0.1), 50
    out = {"a_bits": np.packbits(a[..., :3] > 0), "b_bits": np.packbits(b[..., :3] > 0),
           "a_alpha": np.packbits(a[..., 3] > 0), "b_alpha": np.packbits(b[..., 3] > 0),
           "dewarp_map": dewarp.astype(np.int16), "brief_pairs": pairs,
           "threshold": T, "radius": np.int32(radius)}
    res = {}
    for tag, im8 in (("a", a), ("b", b)):
        rgba64 = im8.astype(np.uint16) * 257      # ImageSharp widens 8-bit sources by x257
        dw = cref.apply_distortion(rgba64, dewarp)
        assert (dw == onp.apply_distortion(rgba64, dewarp)).all()
        g = cref.gray(dw)
        assert (g == onp.gray(dw)).all()
        raw = cref.detect(g, T)
        tw = onp.detect(g, T)
        assert len(raw) == len(tw) and (raw["x"] == tw[:, 0]).all() and (raw["y"] == tw[:, 1]).all() \
            and (raw["fast_score"] == tw[:, 2]).all()
        order = cref.nms(raw, radius)
        o2, rounds = onp.nms_rounds(np.stack([raw["x"], raw["y"]], 1), raw["fast_score"], radius)
        assert (order == o2).all()
        kept = raw[order]
        xy = np.stack([kept["x"], kept["y"]], 1)
        desc = cref.brief(g, xy, pairs)
        assert (desc == onp.brief(g, xy, pairs)).all()
        print("star", tag, "raw", len(raw), "kept", len(kept), "nms rounds", rounds)
        out[tag + "_n_raw"] = np.int32(len(raw))
        out[tag + "_raw"] = np.stack([raw["x"], raw["y"], raw["fast_score"]], 1).astype(np.int32)
        out[tag + "_kp"] = np.stack([kept["x"], kept["y"], kept["fast_score"]], 1).astype(np.int32)
        out[tag + "_value"] = kept["value"].astype(np.float32)
        out[tag + "_desc"] = desc
        res[tag] = desc
    m = pairs_array(cref.match(res["a"], res["b"]))
    tw, rounds = onp.match_rounds(res["a"], res["b"])
    assert (m == tw).all()
    print("star match rounds", rounds)
    out["match_ab"] = m
    np.savez_compressed(os.path.join(HERE, "star_pair.npz"), **out)


def make_dotnet_bmp():
    """The one numeric artefact the reference holds that its C# path produced:
    data/feature_detection_test/output/dotnet_keypoints_backup.bmp = 15pt_star.png with a blue square
    [x-5, x+5) x [y-5, y+5) drawn at every keypoint of the older flow in Photogrammetry/Program.cs:116-149
    (KeypointDetection(0.2f, 50, 256) on Grayscale.FromRgba; squares of "radius" 5, ResultBuilders.cs:41-54).
    Stored: the blue mask as packed bits (data, not the file).  Checked here before writing: every non-blue pixel equals
    the PNG; the mask equals the union of such squares at the oracle's raw FAST hits (the backup shows every hit, i.e. it
    was written before, or without, the eliminator taking effect)."""
    from PIL import Image
    bmp = np.asarray(Image.open("/root/reference/data/feature_detection_test/output/dotnet_keypoints_backup.bmp").convert("RGB"))
    png = np.asarray(Image.open("/root/reference/data/feature_detection_test/15pt_star.png").convert("RGBA"))
    assert (png == load_star("15pt_star.png")).all()          # the same file as the star pair's first image
    assert bmp.shape[:2] == png.shape[:2] == (383, 451)
    blue = (bmp[..., 0] == 0) & (bmp[..., 1] == 0) & (bmp[..., 2] == 255)
    assert (bmp[~blue] == png[..., :3][~blue]).all()
    g = cref.gray(png.astype(np.uint16) * 257)
    raw = cref.detect(g, np.float32(0.2))
    m = np.zeros_like(blue)
    for x, y in zip(raw["x"], raw["y"]):
        m[max(0, y - 5):y + 5, max(0, x - 5):x + 5] = True
    assert (m == blue).all()
    print("dotnet bmp: blue pixels", int(blue.sum()), "raw hits", len(raw))
    np.savez_compressed(os.path.join(HERE, "dotnet_keypoints_mask.npz"), blue_bits=np.packbits(blue),
                        shape=np.array(blue.shape, dtype=np.int32), threshold=np.float32(0.2),
                        radius=np.int32(int(451 * 0.015)), square=np.int32(5))


if __name__ == "__main__":
    make_lego()
    make_star()
    make_dotnet_bmp()
    for f in sorted(os.listdir(HERE)):
        print(f, os.path.getsize(os.path.join(HERE, f)))
