"""ctypes binding of the C oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, bench.py's cpu_baseline leg and
__graft_entry__.smoke().  Nothing under photogrammetry_amd/ imports this module.
Each wrapper cites the reference function it restates; see pgx_oracle.h for the
"parity pinned / unpinned" statement.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")

ORC_OK, ORC_E_DIM, ORC_E_OOB, ORC_E_EMPTY, ORC_E_CAPACITY, ORC_E_BADARG = 0, -1, -2, -3, -4, -5
INT_MAX = 2**31 - 1

KP_DTYPE = np.dtype([("x", "<i4"), ("y", "<i4"), ("fast_score", "<i4"), ("value", "<f4")])
PAIR_DTYPE = np.dtype([("k1", "<i4"), ("k2", "<i4"), ("dist", "<i4")])


def build(force=False):
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "pgx_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.orc_match.restype = C.c_int
        _lib.orc_match_sorted.restype = C.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class OracleError(Exception):
    def __init__(self, code):
        super().__init__({ORC_E_DIM: "ArgumentException (dimension mismatch)",
                          ORC_E_OOB: "IndexOutOfRangeException",
                          ORC_E_EMPTY: "ArgumentOutOfRangeException (empty keypoints2)",
                          ORC_E_BADARG: "ArgumentException"}.get(code, "error %d" % code))
        self.code = code


def apply_distortion(rgba, map_uv):
    """DeWarp.ApplyDistortionMat (DeWarp.cs:19-37). rgba [H][W][4] u16, map_uv [Hm][Wm][2] i32."""
    rgba = np.ascontiguousarray(rgba, dtype=np.uint16)
    map_uv = np.ascontiguousarray(map_uv, dtype=np.int32)
    H, W = rgba.shape[:2]
    out = np.zeros_like(rgba)
    rc = lib().orc_apply_distortion(_p(rgba), W, H, _p(map_uv), map_uv.shape[1], map_uv.shape[0], _p(out))
    if rc != ORC_OK:
        raise OracleError(rc)
    return out


def gray(rgba):
    """Grayscale.FromRgba64 (Grayscale.cs:19-23)."""
    rgba = np.ascontiguousarray(rgba, dtype=np.uint16)
    H, W = rgba.shape[:2]
    out = np.empty((H, W), dtype=np.float32)
    lib().orc_gray(_p(rgba), W, H, _p(out))
    return out


def is_potential_keypoint(img, intensity, x, y, T):
    """KeypointDetection.IsPotentialKeypoint (KeypointDetection.cs:116-133)."""
    img = np.ascontiguousarray(img, dtype=np.float32)
    H, W = img.shape
    rc = lib().orc_is_potential_keypoint(_p(img), W, H, C.c_float(intensity), int(x), int(y), C.c_float(T))
    if rc < 0:
        raise OracleError(rc)
    return bool(rc)


def intensity_if_keypoint(img, x, y, T):
    """KeypointDetection.GetIntensityValueIfKeypoint (:65-114): None or the score."""
    img = np.ascontiguousarray(img, dtype=np.float32)
    H, W = img.shape
    rc = lib().orc_intensity_if_keypoint(_p(img), W, H, int(x), int(y), C.c_float(T))
    if rc < 0:
        raise OracleError(rc)
    return None if rc == 0 else rc


def detect(img, T):
    """KeypointDetection.Detect (:42-63) -> structured array in raster order."""
    img = np.ascontiguousarray(img, dtype=np.float32)
    H, W = img.shape
    cap = 1 << 16
    while True:
        out = np.zeros(cap, dtype=KP_DTYPE)
        n = lib().orc_detect(_p(img), W, H, C.c_float(T), _p(out), cap)
        if n <= cap:
            return out[:n].copy()
        cap = n


def brief(img, xy, pairs):
    """Keypoint.GetBriefDescriptor (Keypoint.cs:29-57) for every (x, y) -> [N][ceil(P/32)] u32."""
    img = np.ascontiguousarray(img, dtype=np.float32)
    pairs = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1, 4)
    H, W = img.shape
    P = pairs.shape[0]
    words = (P + 31) // 32
    xy = np.asarray(xy, dtype=np.int64).reshape(-1, 2)
    out = np.zeros((xy.shape[0], words), dtype=np.uint32)
    for i, (x, y) in enumerate(xy):
        lib().orc_brief(_p(img), W, H, int(x), int(y), _p(pairs), P, C.c_void_p(out[i].ctypes.data))
    return out


def nms(kps, radius):
    """RedundantKeypointEliminator.EliminateRedundantKeypoints (:16-35) -> indices in acceptance order."""
    kps = np.ascontiguousarray(kps, dtype=KP_DTYPE)
    order = np.zeros(max(len(kps), 1), dtype=np.int32)
    n = lib().orc_nms(_p(kps), len(kps), int(radius), _p(order))
    return order[:n].copy()


def _match(fn, d1, d2):
    d1 = np.ascontiguousarray(d1, dtype=np.uint32)
    d2 = np.ascontiguousarray(d2, dtype=np.uint32)
    words = d1.shape[1] if d1.ndim == 2 and d1.shape[0] else (d2.shape[1] if d2.ndim == 2 else 8)
    n1, n2 = d1.shape[0], d2.shape[0]
    out = np.zeros(max(n1, 1), dtype=PAIR_DTYPE)
    rc = fn(_p(d1), n1, _p(d2), n2, words, _p(out))
    if rc != ORC_OK:
        raise OracleError(rc)
    return out[:n1].copy()


def match(d1, d2):
    """KeypointMatching.MatchKeypoints (KeypointMatching.cs:14-69), literal Theta(N^3) loop."""
    return _match(lib().orc_match, d1, d2)


def match_sorted(d1, d2):
    """Same result via sorted scan (cross-check; usable at N in the thousands)."""
    return _match(lib().orc_match_sorted, d1, d2)


def gaussian_pairs(seed, sigma, P):
    """Utils.NextGaussianPair (Utils.cs:14-38) on a seeded splitmix64 stream -> [P][4] i32."""
    out = np.zeros((P, 4), dtype=np.int32)
    lib().orc_gaussian_pairs(C.c_uint64(seed), int(sigma), int(P), _p(out))
    return out


def build_distortion_matrix(W, H, coeffs):
    """DeWarp.GetDistortionMatrix (DeWarp.cs:39-107) -> [H][W][2] i32 (U, V)."""
    k = np.ascontiguousarray(coeffs, dtype=np.float64)
    out = np.zeros((H, W, 2), dtype=np.int32)
    rc = lib().orc_build_distortion_matrix(int(W), int(H), _p(k), len(k), _p(out))
    if rc != ORC_OK:
        raise OracleError(rc)
    return out
