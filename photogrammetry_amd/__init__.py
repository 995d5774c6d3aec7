"""photogrammetry_amd -- MI355X (gfx950) native engine for the hot path of
Takatsuka-Mark/Photogrammetry: dewarp -> FAST-like detect -> NMS -> BRIEF -> brute-force Hamming
match with the reference's greedy assignment.  The product is libpgx.so (hand-written HIP behind
the C ABI in include/pgx.h); this package is the thin host-side mirror used by tests and bench.py.
"""
from .api import (ArgumentException, ArgumentOutOfRangeException, CapacityError, DeWarp, Engine, Grayscale,
                  IndexOutOfRangeException, KEYPOINT_DTYPE, KeypointDetection, KeypointMatching, PAIR_DTYPE,
                  PgxError, RcclError, RedundantKeypointEliminator, build_dewarp_map, comm_unique_id, make_brief_pairs, tracks_host)

__all__ = ["ArgumentException", "ArgumentOutOfRangeException", "CapacityError", "DeWarp", "Engine", "Grayscale",
           "IndexOutOfRangeException", "KEYPOINT_DTYPE", "KeypointDetection", "KeypointMatching", "PAIR_DTYPE",
           "PgxError", "RcclError", "RedundantKeypointEliminator", "build_dewarp_map", "comm_unique_id", "make_brief_pairs", "tracks_host"]
