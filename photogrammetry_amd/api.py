"""Host-side mirror of the reference's ImageProcessing classes over the C ABI (include/pgx.h).

The reference's seam is "concrete class method called from a Dataflow block" (SURVEY D1):
DeWarp.ApplyDistortionMat, Grayscale.FromRgba64 (via Matrix.Convert), KeypointDetection.Detect,
RedundantKeypointEliminator.EliminateRedundantKeypoints, KeypointMatching.MatchKeypoints.
The classes below keep those names, argument meanings and error behaviour (the .NET exception
types are mirrored by the exception classes here) so that tests read like the reference's own.
All arithmetic happens in libpgx.so on the GPU; numpy arrays are only the marshalling format.
The C++ twin of this file (for a compiled host) is photogrammetry_amd/host/pgx_host.hpp.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import (PGX_DIST_NONE, PGX_SRC_RGBA64, PGX_SRC_RGBA8, PGX_E_BADARG, PGX_E_CAPACITY, PGX_E_DIM_MISMATCH, PGX_E_EMPTY_SET,
                   PGX_E_HIP, PGX_E_NOT_CONFIGURED, PGX_E_OOB_SOURCE, PGX_OK)

KEYPOINT_DTYPE = np.dtype([("x", "<i4"), ("y", "<i4"), ("fast_score", "<i4"), ("value", "<f4")])
PAIR_DTYPE = np.dtype([("k1", "<i4"), ("k2", "<i4"), ("dist", "<i4")])


class PgxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("pgx error %d: %s" % (code, msg))
        self.code = code


class ArgumentException(PgxError, ValueError):
    """System.ArgumentException (DeWarp.cs:23, :48)."""


class IndexOutOfRangeException(PgxError, IndexError):
    """System.IndexOutOfRangeException (Matrix.cs:65, :207)."""


class ArgumentOutOfRangeException(PgxError, IndexError):
    """System.ArgumentOutOfRangeException (KeypointMatching.cs:61 with an empty keypoints2)."""


class CapacityError(PgxError):
    pass


_EXC = {PGX_E_DIM_MISMATCH: ArgumentException, PGX_E_BADARG: ArgumentException,
        PGX_E_OOB_SOURCE: IndexOutOfRangeException, PGX_E_EMPTY_SET: ArgumentOutOfRangeException,
        PGX_E_CAPACITY: CapacityError}


class RcclError(PgxError):
    pass


_EXC[_lib.PGX_E_RCCL] = RcclError


def _ptr(a):
    return C.c_void_p(a.ctypes.data) if a is not None else None


def _dptr(t):
    """Device pointer of a torch tensor (or a raw int)."""
    return C.c_void_p(t if isinstance(t, int) else t.data_ptr())


class Engine:
    """One pgx context = one GPU (pgx_ctx_create).  Thin, 1:1 with the C ABI."""

    def __init__(self, device=0):
        self._L = _lib.lib()
        h = C.c_void_p()
        rc = self._L.pgx_ctx_create(int(device), C.byref(h))
        if rc != PGX_OK:
            raise PgxError(rc, "pgx_ctx_create(device=%d) failed: no usable gfx950 device "
                               "(this package has no CPU fallback)" % device)
        self._h = h
        self.device = device
        self.words = 0
        self.P = 0

    def close(self):
        if getattr(self, "_h", None):
            self._L.pgx_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != PGX_OK:
            msg = self._L.pgx_last_error(self._h).decode()
            raise _EXC.get(rc, PgxError)(rc, msg)

    # -- configuration ------------------------------------------------------------------
    def set_stream(self, stream_handle):
        self._chk(self._L.pgx_set_stream(self._h, C.c_void_p(stream_handle)))

    def check_status(self):
        self._chk(self._L.pgx_check_status(self._h))

    def set_dewarp_map(self, map_uv):
        if map_uv is None:
            self._chk(self._L.pgx_set_dewarp_map(self._h, None, 0, 0))
            return
        m = np.ascontiguousarray(map_uv, dtype=np.int32)
        assert m.ndim == 3 and m.shape[2] == 2
        self._chk(self._L.pgx_set_dewarp_map(self._h, _ptr(m), m.shape[1], m.shape[0]))

    def set_dewarp_coeffs(self, W, H, coeffs):
        """Build the table on the device from the five distortion coefficients (DeWarp.GetDistortionMatrix)."""
        k = np.ascontiguousarray(coeffs, dtype=np.float64).reshape(-1)
        self._chk(self._L.pgx_set_dewarp_coeffs(self._h, int(W), int(H), _ptr(k), int(k.size)))

    def get_dewarp_map(self, W, H):
        out = np.zeros((H, W, 2), dtype=np.int32)
        self._chk(self._L.pgx_get_dewarp_map(self._h, _ptr(out), int(W), int(H)))
        return out

    def set_brief_pairs(self, pairs):
        p = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1, 4)
        self._chk(self._L.pgx_set_brief_pairs(self._h, _ptr(p), p.shape[0]))
        self.P = p.shape[0]
        self.words = (self.P + 31) // 32

    def set_detect_params(self, threshold, suppression_radius):
        self._chk(self._L.pgx_set_detect_params(self._h, C.c_float(threshold), int(suppression_radius)))

    def set_capacity(self, max_raw_per_frame, max_keypoints_per_frame):
        self._chk(self._L.pgx_set_capacity(self._h, int(max_raw_per_frame), int(max_keypoints_per_frame)))

    def set_source_format(self, fmt):
        """PGX_SRC_RGBA64 (0, default) or PGX_SRC_RGBA8 (1): 8-bit frames, widened x257 on the device (pgx.h)."""
        self._chk(self._L.pgx_set_source_format(self._h, int(fmt)))
        self._src_dtype = np.uint8 if int(fmt) == PGX_SRC_RGBA8 else np.uint16

    def set_match_chunk(self, image_pairs_per_chunk):
        self._chk(self._L.pgx_set_match_chunk(self._h, int(image_pairs_per_chunk)))

    def wait_stage(self, other, stage):
        """This context's stream waits for a stage (PGX_STAGE_*) of `other`'s most recent call of that kind (pgx.h)."""
        self._chk(self._L.pgx_wait_stage(self._h, other._h, int(stage)))

    def gate_match(self, other, stage):
        """The next matcher call of this context waits for a stage of `other` between its init kernel and its first distance
        round (pgx_gate_match; one shot)."""
        self._chk(self._L.pgx_gate_match(self._h, other._h, int(stage)))

    # -- stage-granular host API ----------------------------------------------------------
    def dewarp(self, rgba64):
        a = np.ascontiguousarray(rgba64, dtype=getattr(self, "_src_dtype", np.uint16))
        out = np.empty(a.shape, dtype=np.uint16)
        self._chk(self._L.pgx_dewarp(self._h, _ptr(a), a.shape[1], a.shape[0], _ptr(out)))
        return out

    def gray(self, rgba64):
        a = np.ascontiguousarray(rgba64, dtype=getattr(self, "_src_dtype", np.uint16))
        out = np.empty(a.shape[:2], dtype=np.float32)
        self._chk(self._L.pgx_gray(self._h, _ptr(a), a.shape[1], a.shape[0], _ptr(out)))
        return out

    def fast(self, gray, capacity=None):
        g = np.ascontiguousarray(gray, dtype=np.float32)
        cap = int(capacity) if capacity is not None else max(1, g.size)
        out = np.zeros(cap, dtype=KEYPOINT_DTYPE)
        n = C.c_int(0)
        self._chk(self._L.pgx_fast(self._h, _ptr(g), g.shape[1], g.shape[0], _ptr(out), cap, C.byref(n)))
        return out[:n.value].copy()

    def brief(self, gray, kps):
        g = np.ascontiguousarray(gray, dtype=np.float32)
        k = np.ascontiguousarray(kps, dtype=KEYPOINT_DTYPE)
        out = np.zeros((len(k), self.words), dtype=np.uint32)
        self._chk(self._L.pgx_brief(self._h, _ptr(g), g.shape[1], g.shape[0], _ptr(k), len(k), _ptr(out)))
        return out

    def nms(self, kps, W, H):
        k = np.ascontiguousarray(kps, dtype=KEYPOINT_DTYPE)
        order = np.zeros(max(1, len(k)), dtype=np.int32)
        n = C.c_int(0)
        self._chk(self._L.pgx_nms(self._h, _ptr(k), len(k), int(W), int(H), _ptr(order), C.byref(n)))
        return order[:n.value].copy()

    def match(self, desc1, desc2):
        d1 = np.ascontiguousarray(desc1, dtype=np.uint32)
        d2 = np.ascontiguousarray(desc2, dtype=np.uint32)
        words = d1.shape[1] if d1.ndim == 2 and d1.shape[0] else (d2.shape[1] if d2.ndim == 2 and d2.shape[0] else 8)
        out = np.zeros(max(1, len(d1)), dtype=PAIR_DTYPE)
        self._chk(self._L.pgx_match(self._h, _ptr(d1), len(d1), _ptr(d2), len(d2), int(words), _ptr(out)))
        return out[:len(d1)].copy()

    def match_batch(self, descs, pair_list):
        """pgx_match_batch: descs = list of [n_f][words] arrays (one per frame), pair_list = [(a, b), ...] -> list of
        PAIR_DTYPE arrays, one per image pair, in the reference's emission order.  Raises ArgumentOutOfRangeException
        (after all lists are computed) when some pair has an empty second set."""
        F = len(descs)
        arrs = [np.ascontiguousarray(d, dtype=np.uint32) for d in descs]
        words = next((a.shape[1] for a in arrs if a.ndim == 2 and a.shape[0]), 8)
        for f, a in enumerate(arrs):   # the library copies counts[f] * words words from every descs[f]
            if len(a) and (a.ndim != 2 or a.shape[1] != words):
                raise ValueError("descs[%d] has shape %s; every non-empty set must be [n][%d]" % (f, a.shape, words))
        counts = np.array([len(a) for a in arrs], dtype=np.int32)
        ptrs = (C.c_void_p * max(1, F))(*[a.ctypes.data if len(a) else None for a in arrs])
        pl = np.ascontiguousarray(pair_list, dtype=np.int32).reshape(-1, 2)
        M = len(pl)
        total = int(sum(int(counts[a]) for a, _ in pl))
        out = np.zeros(max(1, total), dtype=PAIR_DTYPE)
        offs = np.zeros(M + 1, dtype=np.int64)
        rc = self._L.pgx_match_batch(self._h, ptrs, _ptr(counts), F, int(words), _ptr(pl), M, _ptr(out), _ptr(offs))
        lists = [out[offs[m]:offs[m + 1]].copy() for m in range(M)]
        if rc == _lib.PGX_E_EMPTY_SET:
            self.last_batch_lists = lists   # all lists are valid; the reference would have thrown at the empty pair
        self._chk(rc)
        return lists

    def detect(self, rgba64, capacity=8192):
        a = np.ascontiguousarray(rgba64, dtype=getattr(self, "_src_dtype", np.uint16))
        kp = np.zeros(capacity, dtype=KEYPOINT_DTYPE)
        desc = np.zeros((capacity, max(1, self.words)), dtype=np.uint32)
        n, nraw = C.c_int(0), C.c_int(0)
        self._chk(self._L.pgx_detect(self._h, _ptr(a), a.shape[1], a.shape[0], _ptr(kp), _ptr(desc), int(capacity),
                                     C.byref(n), C.byref(nraw)))
        return kp[:n.value].copy(), desc[:n.value].copy(), nraw.value

    # -- device-resident batched API (torch tensors or raw device pointers) -------------------
    def detect_batch_dev(self, d_rgba64, F, W, H, d_kp, d_desc, d_counts, d_nraw, capacity):
        self._chk(self._L.pgx_detect_batch_dev(self._h, _dptr(d_rgba64), int(F), int(W), int(H), _dptr(d_kp),
                                               _dptr(d_desc), _dptr(d_counts), _dptr(d_nraw), int(capacity)))

    def match_batch_dev(self, d_desc, d_counts, stride, words, d_pairlist, M, d_out, max_count=None):
        self._chk(self._L.pgx_match_batch_dev(self._h, _dptr(d_desc), _dptr(d_counts), int(stride), int(words),
                                              _dptr(d_pairlist), int(M),
                                              int(stride if max_count is None else max_count), _dptr(d_out)))

    # -- RANSAC fundamental matrix / pose (device tensors) ------------------------------------------
    def fundamental_ransac_dev(self, d_kp, d_matches, d_counts, d_pairlist, M, stride, n_samples, pairs_per_sample, threshold,
                               d_F, d_inliers, d_best_sample, rank_check=False, seed=0):
        self._chk(self._L.pgx_fundamental_ransac_dev(self._h, _dptr(d_kp), _dptr(d_matches), _dptr(d_counts), _dptr(d_pairlist),
                                                     int(M), int(stride), int(n_samples), int(pairs_per_sample),
                                                     C.c_float(threshold), 1 if rank_check else 0, C.c_uint64(seed),
                                                     _dptr(d_F), _dptr(d_inliers), _dptr(d_best_sample)))

    def pose_dev(self, d_kp, d_matches, d_counts, d_pairlist, M, stride, d_F, d_Rt, d_votes, d_best, d_points=None):
        self._chk(self._L.pgx_pose_dev(self._h, _dptr(d_kp), _dptr(d_matches), _dptr(d_counts), _dptr(d_pairlist), int(M),
                                       int(stride), _dptr(d_F), _dptr(d_Rt), _dptr(d_votes), _dptr(d_best),
                                       _dptr(d_points) if d_points is not None else None))

    # -- the track graph on the device (pgx_tracks_dev) ----------------------------------------------
    def tracks_dev(self, d_matches, d_counts, d_pairlist, M, F, stride, n_frames, max_dist, min_len, d_track_of, d_offsets,
                   d_nodes, d_summary, d_frame_ids=None):
        """Connected components over the gated match lists where they sit in HBM (pgx.h: order-independent semantics).
        d_track_of [n_frames][stride], d_offsets [n_frames * stride + 1], d_nodes [n_frames * stride][2], d_summary [8]."""
        self._chk(self._L.pgx_tracks_dev(self._h, _dptr(d_matches), _dptr(d_counts), _dptr(d_pairlist), int(M), int(F), int(stride),
                                         _dptr(d_frame_ids) if d_frame_ids is not None else None, int(n_frames), int(max_dist),
                                         int(min_len), _dptr(d_track_of), _dptr(d_offsets), _dptr(d_nodes), _dptr(d_summary)))

    # -- multi-GPU: the context's own RCCL communicator (pgx_comm_*) ------------------------------
    def comm_init(self, rank, world, unique_id):
        """Collective: every rank calls this with the 128 bytes rank 0 got from comm_unique_id()."""
        buf = (C.c_char * 128).from_buffer_copy(bytes(unique_id))
        self._chk(self._L.pgx_comm_init(self._h, int(rank), int(world), buf))

    def comm_destroy(self):
        self._chk(self._L.pgx_comm_destroy(self._h))

    def comm_info(self):
        r, w = C.c_int(0), C.c_int(0)
        self._chk(self._L.pgx_comm_info(self._h, C.byref(r), C.byref(w)))
        return r.value, w.value

    def allgather_dev(self, d_buf, bytes_per_rank):
        """In-place all-gather of fixed-size records on the context's stream (rank-major buffer)."""
        self._chk(self._L.pgx_allgather_dev(self._h, _dptr(d_buf), C.c_size_t(int(bytes_per_rank))))

    def sequence_step_dev(self, d_frames_local, n_local_frames, frame_slots, W, H, d_kp_local, d_desc_all, d_counts_all,
                          d_nraw_local, capacity, d_pairlist_local, n_local_pairs, pair_slots, d_out_all):
        """The four phases of one sharded job (detect -> all-gather -> match -> all-gather) in one C call."""
        self._chk(self._L.pgx_sequence_step_dev(self._h, _dptr(d_frames_local), int(n_local_frames), int(frame_slots), int(W), int(H),
                                                _dptr(d_kp_local), _dptr(d_desc_all), _dptr(d_counts_all), _dptr(d_nraw_local),
                                                int(capacity), _dptr(d_pairlist_local), int(n_local_pairs), int(pair_slots),
                                                _dptr(d_out_all)))

    def debug_counters(self):
        out = np.zeros(8, dtype=np.int64)
        self._chk(self._L.pgx_debug_counters(self._h, _ptr(out)))
        return out.tolist()

    def match_stats(self):
        r, ev, ev0 = C.c_int(0), C.c_int64(0), C.c_int64(0)
        self._chk(self._L.pgx_match_stats(self._h, C.byref(r), C.byref(ev), C.byref(ev0)))
        return r.value, ev.value, ev0.value

    # -- measurement ------------------------------------------------------------------------
    def profile_enable(self, on=True):
        self._chk(self._L.pgx_profile_enable(self._h, 1 if on else 0))

    def profile_filter(self, name=None):
        self._chk(self._L.pgx_profile_filter(self._h, name.encode() if name else None))

    def profile_serialize(self, on=True):
        self._chk(self._L.pgx_profile_serialize(self._h, 1 if on else 0))

    def profile_reset(self):
        self._chk(self._L.pgx_profile_reset(self._h))

    def profile_get(self, name):
        n, ms = C.c_int(0), C.c_double(0.0)
        self._chk(self._L.pgx_profile_get(self._h, name.encode(), C.byref(n), C.byref(ms)))
        return n.value, ms.value


def comm_unique_id():
    """128 bytes from ncclGetUniqueId (rank 0 makes them, the host hands them to every rank)."""
    buf = (C.c_char * 128)()
    rc = _lib.lib().pgx_comm_unique_id(buf)
    if rc != PGX_OK:
        raise PgxError(rc, "pgx_comm_unique_id: librccl could not be loaded or ncclGetUniqueId failed")
    return bytes(buf)


def tracks_host(counts, pair_list, lists, max_dist, min_len=2):
    """The host form of the track graph (pgx_tracks_*: sequential, no GPU work; same semantics as Engine.tracks_dev).
    counts [F]; pair_list [(a, b)]; lists[m] = that pair's match list ([n][3] ints or PAIR_DTYPE).
    -> (tracks, dropped_components, dropped_nodes); tracks = list of [(frame, keypoint)] lists in pgx_tracks_get's order."""
    L = _lib.lib()
    c = np.ascontiguousarray(counts, dtype=np.int32)
    h = C.c_void_p()
    if L.pgx_tracks_create(_ptr(c), len(c), C.byref(h)) != PGX_OK:
        raise PgxError(PGX_E_BADARG, "pgx_tracks_create")
    try:
        for (a, b), rows in zip(pair_list, lists):
            rows = np.asarray(rows)
            if rows.dtype == PAIR_DTYPE:
                rows = np.stack([rows["k1"], rows["k2"], rows["dist"]], axis=1)
            rows = np.ascontiguousarray(rows.reshape(-1, 3)[:int(c[a])], dtype=np.int32)
            if L.pgx_tracks_add_pair(h, int(a), int(b), _ptr(rows), len(rows), int(max_dist)) != PGX_OK:
                raise PgxError(PGX_E_BADARG, "pgx_tracks_add_pair(%d, %d)" % (a, b))
        nt, nn, nd, ndn = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0)
        if L.pgx_tracks_finish(h, int(min_len), C.byref(nt), C.byref(nn)) != PGX_OK:
            raise PgxError(PGX_E_BADARG, "pgx_tracks_finish")
        off = np.zeros(nt.value + 1, dtype=np.int32)
        nodes = np.zeros((max(nn.value, 1), 2), dtype=np.int32)
        if L.pgx_tracks_get(h, _ptr(off), _ptr(nodes)) != PGX_OK or L.pgx_tracks_dropped(h, C.byref(nd), C.byref(ndn)) != PGX_OK:
            raise PgxError(PGX_E_BADARG, "pgx_tracks_get")
        return [[(int(f), int(k)) for f, k in nodes[off[t]:off[t + 1]]] for t in range(nt.value)], nd.value, ndn.value
    finally:
        L.pgx_tracks_destroy(h)


def make_brief_pairs(seed, sigma, P):
    """Seeded table with the reference's generator formula (Utils.cs:14-38)."""
    out = np.zeros((P, 4), dtype=np.int32)
    rc = _lib.lib().pgx_make_brief_pairs(C.c_uint64(seed), int(sigma), int(P), _ptr(out))
    if rc != PGX_OK:
        raise PgxError(rc, "pgx_make_brief_pairs")
    return out


def build_dewarp_map(W, H, coeffs):
    """DeWarp.GetDistortionMatrix (DeWarp.cs:39-107) -> int32 [H][W][2]."""
    k = np.ascontiguousarray(coeffs, dtype=np.float64)
    out = np.zeros((H, W, 2), dtype=np.int32)
    rc = _lib.lib().pgx_build_dewarp_map(int(W), int(H), _ptr(k), len(k), _ptr(out))
    if rc != PGX_OK:
        raise ArgumentException(rc, "You must pass exactly 5 distortion coefficients")
    return out


# ---- the reference's classes --------------------------------------------------------------------

class DeWarp:
    """ImageProcessing/DeWarp.cs.  Options: MatrixDimensions (W, H), DistortionCoefficients[5]."""

    def __init__(self, engine, width, height, distortion_coefficients):
        self._e = engine
        self.width, self.height = int(width), int(height)
        self.coeffs = list(distortion_coefficients)

    def GetDistortionMatrix(self):
        return build_dewarp_map(self.width, self.height, self.coeffs)

    def ApplyDistortionMat(self, image_rgba64, distortion_matrix):
        """DeWarp.cs:19-37.  ArgumentException on size mismatch, IndexOutOfRangeException on an
        out-of-image source coordinate."""
        self._e.set_dewarp_map(distortion_matrix)
        return self._e.dewarp(image_rgba64)


class Grayscale:
    """Images.Abstractions/Pixels/Grayscale.cs:19-23 applied through Matrix.Convert."""

    def __init__(self, engine):
        self._e = engine

    def FromRgba64(self, image_rgba64):
        return self._e.gray(image_rgba64)


class KeypointDetection:
    """ImageProcessing/KeypointDetection.cs.  Options: Threshold, (GaussianStandardDeviation,
    NumGaussianPairs -> here the explicit pair table)."""

    def __init__(self, engine, threshold, gaussian_pairs, suppression_radius=0):
        self._e = engine
        engine.set_detect_params(threshold, suppression_radius)
        engine.set_brief_pairs(gaussian_pairs)

    def Detect(self, gray):
        """KeypointDetection.cs:42-63 -> (keypoints, descriptors) in raster order."""
        kps = self._e.fast(gray)
        return kps, self._e.brief(gray, kps)


class RedundantKeypointEliminator:
    """ImageProcessing/RedundantKeypointEliminator.cs.  Option: SuppressionRadius."""

    def __init__(self, engine, suppression_radius, threshold=0.0):
        self._e = engine
        self._r = int(suppression_radius)
        self._t = threshold

    def EliminateRedundantKeypoints(self, keypoints, width, height):
        """Returns the indices of the accepted keypoints in acceptance order (:16-35)."""
        self._e.set_detect_params(self._t, self._r)
        return self._e.nms(keypoints, width, height)


class KeypointMatching:
    """ImageProcessing/KeypointMatching.cs."""

    def __init__(self, engine):
        self._e = engine

    def MatchKeypoints(self, descriptors1, descriptors2):
        """KeypointMatching.cs:14-69 -> PAIR_DTYPE[n1] (indices + Hamming distance)."""
        return self._e.match(descriptors1, descriptors2)
