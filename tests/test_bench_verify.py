"""bench.py's self-check (`verified` in its JSON line) must not be vacuous: given a job whose results are the
oracle's it passes, and one wrong match entry, one moved keypoint or one flipped descriptor bit makes it fail.
The job here is a stand-in holding CPU tensors (no GPU); the function under test is bench.verify_job itself."""
import numpy as np
import pytest
import torch

import bench
from oracle import cref
from photogrammetry_amd import synth


class FakeJob:
    def __init__(self, kp, desc, counts, nraw, lists):
        self.kp_l = [torch.from_numpy(k) for k in kp]
        self.nraw_l = [torch.tensor(n) for n in nraw]
        self._desc = [torch.from_numpy(d) for d in desc]
        self._counts = np.asarray(counts)
        self._lists = [torch.from_numpy(x) for x in lists]
        self.my_frames = list(range(len(kp)))

    def counts(self):
        return self._counts

    def descriptors(self, f):
        return self._desc[f]

    def matches(self, p):
        return self._lists[p]


@pytest.fixture(scope="module")
def small_job(monkeypatch_module):
    W, H, CAP = 160, 120, 256
    for k, v in (("W", W), ("H", H), ("NKP", CAP), ("RADIUS", 5)):
        monkeypatch_module.setattr(bench, k, v)
    pairs_tbl = cref.gaussian_pairs(0, 20, 256)
    base = synth.make_frame(W, H, seed=5, n_shapes=60)
    frames = [base, synth.shift_frame(base, 3, 1), synth.shift_frame(base, 6, 2)]
    kp, desc, counts, nraw = [], [], [], []
    for fr in frames:
        g = cref.gray(fr)
        raw = cref.detect(g, np.float32(bench.THRESH))
        kept = raw[cref.nms(raw, 5)][:CAP]
        k = np.zeros((CAP, 4), np.int32)
        k[:len(kept), 0], k[:len(kept), 1], k[:len(kept), 2] = kept["x"], kept["y"], kept["fast_score"]
        k[:len(kept), 3] = kept["value"].view(np.int32)
        d = np.zeros((CAP, 8), np.uint32)
        d[:len(kept)] = cref.brief(g, np.stack([kept["x"], kept["y"]], 1), pairs_tbl)
        kp.append(k), desc.append(d.view(np.int32)), counts.append(len(kept)), nraw.append(len(raw))
    assert min(counts) > 20
    pl = [(0, 1), (0, 2), (1, 2), (2, 0)]
    lists = []
    for a, b in pl:
        m = cref.match_sorted(desc[a].view(np.uint32)[:counts[a]], desc[b].view(np.uint32)[:counts[b]])
        out = np.zeros((CAP, 3), np.int32)
        out[:counts[a]] = np.stack([m["k1"], m["k2"], m["dist"]], 1)
        lists.append(out)
    return dict(kp=kp, desc=desc, counts=counts, nraw=nraw, lists=lists, pl=pl, base=base, pairs_tbl=pairs_tbl)


@pytest.fixture(scope="module")
def monkeypatch_module():
    mp = pytest.MonkeyPatch()
    yield mp
    mp.undo()


def _verify(j, **over):
    d = {k: [x.copy() if hasattr(x, "copy") else x for x in j[k]] for k in ("kp", "desc", "counts", "nraw", "lists")}
    for k, fn in over.items():
        fn(d[k])
    job = FakeJob(d["kp"], d["desc"], d["counts"], d["nraw"], d["lists"])
    return bench.verify_job(job, j["pl"], j["base"], None, j["pairs_tbl"], None)


def test_verify_accepts_the_oracles_own_results(small_job):
    res = _verify(small_job)
    assert res["ok"] and len(res["pairs"]) == 3 and len(res["frames"]) == 1
    assert [tuple(p["pair"]) for p in res["pairs"]] == [(0, 1), (1, 2), (2, 0)]      # first, middle, last


@pytest.mark.parametrize("what", ["match_k2", "match_dist", "match_last_pair", "keypoint", "grey", "descriptor", "raw_count"])
def test_verify_rejects_one_wrong_value(small_job, what):
    def bump(arr, i, j, k):
        def f(lst):
            lst[i][j, k] += 1
        return f
    n0 = small_job["counts"][0]
    over = {"match_k2": dict(lists=bump(None, 0, n0 - 1, 1)),
            "match_dist": dict(lists=bump(None, 2, 0, 2)),
            "match_last_pair": dict(lists=bump(None, 3, 3, 0)),
            "keypoint": dict(kp=bump(None, 0, n0 // 2, 0)),
            "grey": dict(kp=bump(None, 0, 1, 3)),                       # one ulp of one grey value
            "descriptor": dict(desc=bump(None, 0, n0 - 1, 7)),
            "raw_count": dict(nraw=lambda lst: lst.__setitem__(0, lst[0] + 1))}[what]
    res = _verify(small_job, **over)
    assert not res["ok"]


def test_verify_does_not_look_at_unsampled_pairs(small_job):
    """Documents the sampling: pair 1 of 4 is not one of first / middle / last, so the check is a sample, not a proof --
    the full-size parity tests are the proof (tests/test_gpu_sequence.py)."""
    res = _verify(small_job, lists=lambda lst: lst[1].__setitem__((0, 1), lst[1][0, 1] + 1))
    assert res["ok"]
