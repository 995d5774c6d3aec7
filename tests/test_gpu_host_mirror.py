"""GPU test: the compiled-host mirror (photogrammetry_amd/host/pgx_host.hpp, same class names and
exception behaviour as the C#) driven by host_selftest, checked line by line against the CPU oracle."""
import os
import subprocess

import numpy as np
import pytest

from oracle import cref

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
EXE = os.path.join(ROOT, "photogrammetry_amd", "host", "host_selftest")


def _image(W, H, seed, shift):
    def lcg():
        nonlocal seed
        seed = (seed * 1664525 + 1013904223) & 0xFFFFFFFF
        return seed >> 8
    cw = (W + shift) // 4 + 2
    cells = np.array([(lcg() & 3) * 21845 for _ in range(cw * (H // 4 + 2))], dtype=np.uint16).reshape(-1, cw)
    ys, xs = np.mgrid[0:H, 0:W]
    v = cells[ys // 4, (xs + shift) // 4]
    img = np.empty((H, W, 4), dtype=np.uint16)
    img[..., 0] = img[..., 1] = img[..., 2] = v
    img[..., 3] = 65535
    return img


def test_host_mirror_matches_oracle():
    if not os.path.exists(EXE):
        subprocess.check_call(["make", "-C", os.path.dirname(EXE)])
    out = subprocess.run([EXE], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.strip().splitlines()
    W, H, T, r = 240, 180, np.float32(0.1), 5
    dmap = cref.build_distortion_matrix(W, H, [3e-4, 1e-7, 0, 0, 0])
    pairs = cref.gaussian_pairs(5, 20, 256)
    exp, descs = [], []
    for k in range(2):
        g = cref.gray(cref.apply_distortion(_image(W, H, 77, 8 * k), dmap))
        raw = cref.detect(g, T)
        kept = raw[cref.nms(raw, r)]
        desc = cref.brief(g, np.stack([kept["x"], kept["y"]], 1), pairs)
        descs.append(desc)
        exp.append("image %d raw %d kept %d" % (k, len(raw), len(kept)))
        for p, d in zip(kept, desc):
            exp.append("kp %d %d %d %08x " % (p["x"], p["y"], p["fast_score"], np.float32(p["value"]).view(np.uint32))
                       + " ".join("%08x" % w for w in d))
    assert len(descs[0]) > 8 and len(descs[1]) > 8
    for m in cref.match(descs[0], descs[1]):
        exp.append("pair %d %d %d" % (m["k1"], m["k2"], m["dist"]))
    for k, (a, b) in enumerate([(0, 1), (1, 0), (1, 1)]):
        for m in cref.match(descs[a], descs[b]):
            exp.append("batch %d %d %d %d" % (k, m["k1"], m["k2"], m["dist"]))
    exp.append("exceptions 15")
    assert lines == exp
