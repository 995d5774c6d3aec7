"""ctypes loader for libpgx.so -- the only compute backend of this package.

There is deliberately no fallback: if the HIP library is missing or cannot create a
gfx950 context, importing users get a loud error (PgxError / OSError), never a CPU path.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PGX_LIB", os.path.join(_HERE, "libpgx.so"))  # PGX_LIB: developer A/B builds
CSRC = os.path.join(_HERE, "csrc")

PGX_OK = 0
PGX_E_DIM_MISMATCH, PGX_E_OOB_SOURCE, PGX_E_EMPTY_SET, PGX_E_CAPACITY = 1, 2, 3, 4
PGX_E_BADARG, PGX_E_HIP, PGX_E_NOT_CONFIGURED, PGX_E_RCCL = 5, 6, 7, 8
PGX_DIST_NONE = 2**31 - 1
PGX_SRC_RGBA64, PGX_SRC_RGBA8 = 0, 1
PGX_STAGE_DETECT, PGX_STAGE_MATCH_WIDE, PGX_STAGE_MATCH_ROWS, PGX_STAGE_MATCH_DONE = 0, 1, 2, 3

# every symbol include/pgx.h declares (tests/test_abi_symbols.py checks the header against this)
EXPORTS = [
    "pgx_ctx_create", "pgx_ctx_destroy", "pgx_last_error", "pgx_version", "pgx_set_stream",
    "pgx_check_status", "pgx_set_dewarp_map", "pgx_set_dewarp_coeffs", "pgx_get_dewarp_map", "pgx_set_brief_pairs",
    "pgx_set_detect_params",
    "pgx_set_capacity", "pgx_set_source_format", "pgx_set_match_chunk", "pgx_dewarp", "pgx_gray", "pgx_fast", "pgx_brief", "pgx_nms", "pgx_match", "pgx_match_batch",
    "pgx_detect", "pgx_detect_batch_dev", "pgx_match_batch_dev", "pgx_wait_stage", "pgx_gate_match", "pgx_profile_enable",
    "pgx_profile_get", "pgx_profile_filter", "pgx_profile_reset", "pgx_profile_serialize", "pgx_match_stats", "pgx_debug_counters", "pgx_make_brief_pairs",
    "pgx_build_dewarp_map", "pgx_comm_unique_id", "pgx_comm_init", "pgx_comm_destroy", "pgx_comm_info",
    "pgx_allgather_dev", "pgx_sequence_step_dev", "pgx_tracks_create", "pgx_tracks_destroy", "pgx_tracks_add_pair",
    "pgx_tracks_finish", "pgx_tracks_get", "pgx_tracks_dropped", "pgx_tracks_dev", "pgx_fundamental_ransac_dev", "pgx_pose_dev",
]


def build(force=False):
    """Compile libpgx.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC, "-j8"]
    if force:
        subprocess.check_call(["make", "-C", CSRC, "clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(cmd, stdout=subprocess.DEVNULL)
    if not os.path.exists(LIB_PATH):
        raise OSError("libpgx.so was not produced by " + " ".join(cmd))
    return LIB_PATH


_lib = None


def lib():
    """Load libpgx.so (in-tree).  Raises OSError if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OSError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(there is no CPU fallback)" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        L.pgx_last_error.restype = C.c_char_p
        L.pgx_version.restype = C.c_char_p
        L.pgx_ctx_destroy.restype = None
        L.pgx_tracks_destroy.restype = None
        for name in EXPORTS:
            getattr(L, name)  # AttributeError if the ABI lost a symbol
        _lib = L
    return _lib
