/*
 * pgx.h -- C ABI of libpgx.so: MI355X (gfx950) native dewarp -> FAST-like detect -> NMS ->
 * BRIEF -> all-pairs Hamming match with the reference's greedy one-to-one assignment.
 *
 * This is the drop-in boundary for the hot path of Takatsuka-Mark/Photogrammetry
 * (dotnet_src/ImageProcessing).  The reference has no FFI today; every entry
 * point below names the C# member it replaces, and INTEGRATION.md shows the [DllImport]
 * stubs a maintainer would add.  Plain pointers and sizes only; no C++ or torch types.
 *
 * Conventions
 *   - Images are row-major [H][W]; pixel (x, y) = column x, row y = the reference's
 *     Matrix<T>[x, y] (Math/LinearAlgebra/Matrix.cs:44-76).
 *   - Rgba64 pixel = 4 x uint16 {R,G,B,A} (Images.Abstractions/Pixels/Rgba64.cs:3-9).
 *   - A descriptor is ceil(P/32) little-endian uint32 words of the reference's BigInteger:
 *     bit b of the BigInteger is bit (b & 31) of word (b >> 5); BRIEF test pair p lands on
 *     bit P-1-p (ImageProcessing.Abstractions/Keypoint.cs:29-57).
 *   - Every function returns PGX_OK or a PGX_E_* code; pgx_last_error() gives the text.
 *     No C++ exception crosses this boundary.
 *   - "host" entry points take host pointers, copy in/out and return when the result is in
 *     the caller's buffer.  "_dev" entry points take DEVICE pointers, enqueue on the
 *     context's stream and return immediately; data errors surface in pgx_check_status().
 *   - The caller owns every buffer; the library keeps no caller pointer after a call returns.
 *   - One context per GPU.  Calls on one context are serialised by an internal mutex
 *     (the reference never re-enters a stage: TestService.cs:25,137-152), so a context may be called from several host
 *     threads -- the reference's pipeline runs ApplyDistortionMat on image k + 1 beside Detect on image k
 *     (TestService.cs:25,146-149) -- but its stages then run one after the other; one context per stage (each with its own
 *     configuration and workspaces) lets them overlap on the device.  tests/test_gpu_threads.py covers both.
 *   - pgx_last_error(ctx) is per calling thread: it returns the text of that thread's most recent failing call on ctx
 *     (valid until the thread's next failing call or next pgx_last_error); a thread that has not failed on ctx gets a copy
 *     of the most recent failure of any thread.
 */
#ifndef PGX_H
#define PGX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PGX_OK               0
#define PGX_E_DIM_MISMATCH   1  /* ArgumentException            DeWarp.cs:22-23           */
#define PGX_E_OOB_SOURCE     2  /* IndexOutOfRangeException     Matrix.cs:63-66,204-209   */
#define PGX_E_EMPTY_SET      3  /* ArgumentOutOfRangeException  KeypointMatching.cs:61    */
#define PGX_E_CAPACITY       4  /* an output or workspace capacity was exceeded           */
#define PGX_E_BADARG         5  /* ArgumentException / null pointer / unsupported size    */
#define PGX_E_HIP            6  /* a HIP runtime call failed                              */
#define PGX_E_NOT_CONFIGURED 7  /* stage used before its pgx_set_* call                   */
#define PGX_E_RCCL           8  /* an RCCL call failed, or librccl could not be loaded    */

#define PGX_DIST_NONE 2147483647 /* int.MaxValue: tail entries when N1 > N2 (KeypointMatching.cs:40-42) */

typedef struct pgx_ctx pgx_ctx;

/* Keypoint{Coordinate, FastScore, Value} (ImageProcessing.Abstractions/Keypoint.cs:11-15);
 * the BriefDescriptor travels in a separate [N][words] array. */
typedef struct { int32_t x, y, fast_score; float value; } pgx_keypoint;

/* KeypointPair{Keypoint1, Keypoint2, Distance} as indices into the two input lists
 * (ImageProcessing.Abstractions/KeypointPair.cs:3-8). */
typedef struct { int32_t k1, k2, dist; } pgx_pair;

/* ---- context ------------------------------------------------------------------------ */
int  pgx_ctx_create(int device, pgx_ctx **out);
void pgx_ctx_destroy(pgx_ctx *ctx);
const char *pgx_last_error(pgx_ctx *ctx);
const char *pgx_version(void);
/* Use the caller's HIP stream (hipStream_t) for all work; NULL = the context's own stream. */
int  pgx_set_stream(pgx_ctx *ctx, void *hip_stream);
/* Wait for the stream and report the first data error of the "_dev" calls since the last check. */
int  pgx_check_status(pgx_ctx *ctx);

/* ---- init-time configuration (replaces DI-bound options, Program.cs:61-69) ----------- */
/* Pixel type of every `rgba` argument below (host and device): PGX_SRC_RGBA64 (default) = 4 x uint16 as
 * Rgba64 (Rgba64.cs:3-9); PGX_SRC_RGBA8 = 4 x uint8, widened on the device to c * 257 per channel, which is what an
 * 8-bit file becomes when LocalImageReader loads it as Rgba64 (LocalImageReader.cs:22; SURVEY 8f-4 "ingest"): half the
 * upload and half the gathered bytes, identical results.  pgx_dewarp's output stays Rgba64. */
enum { PGX_SRC_RGBA64 = 0, PGX_SRC_RGBA8 = 1 };
int pgx_set_source_format(pgx_ctx *ctx, int format);
/* DeWarpTransformStepFactory.Initialize (DeWarpTransformStepFactory.cs:26-31): the Matrix<Uv>
 * built by DeWarp.GetDistortionMatrix, as host int32 [H][W][2] = (U, V).  NULL = stage off. */
int pgx_set_dewarp_map(pgx_ctx *ctx, const int32_t *uv, int W, int H);
/* The same table built ON THE DEVICE from DeWarpOptions.DistortionCoefficients (DeWarp.GetDistortionMatrix,
 * DeWarp.cs:39-107; exactly 5 coefficients or PGX_E_BADARG like DeWarp.cs:46-48): no 8-66 MB upload, no host
 * loop.  Same float64 formulas as pgx_build_dewarp_map; the device's libm differs from the host's in the last
 * ulp, so isolated entries can differ by +-1 from the host-built table ("within +-1 px, unpinned"). */
int pgx_set_dewarp_coeffs(pgx_ctx *ctx, int W, int H, const double *coeffs, int ncoeffs);
/* Copy the context's current table back to the host, int32 [H][W][2] (inspection / caching by the host). */
int pgx_get_dewarp_map(pgx_ctx *ctx, int32_t *uv_out, int W, int H);
/* KeypointDetection ctor's _gaussianKeypairs (KeypointDetection.cs:35-39): host int32 [P][4] =
 * (x1, y1, x2, y2).  The table is an INPUT because the reference draws it unseeded. */
int pgx_set_brief_pairs(pgx_ctx *ctx, const int32_t *pairs, int P);
/* KeypointDetectionOptions.Threshold, RedundantKeypointEliminationOptions.SuppressionRadius. */
int pgx_set_detect_params(pgx_ctx *ctx, float threshold, int suppression_radius);
/* Per-frame limits of the fused detect path.  max_raw_per_frame: raw FAST hits kept for NMS (more raise
 * PGX_E_CAPACITY).  max_keypoints_per_frame: survivor LIMIT -- a frame's list is cut to its first
 * max_keypoints_per_frame entries in NMS order without an error (a harness-side truncation: the reference has
 * no cap; default 2^20 = none).  Survivors beyond a call's own `capacity` still raise
 * PGX_E_CAPACITY. */
int pgx_set_capacity(pgx_ctx *ctx, int max_raw_per_frame, int max_keypoints_per_frame);
/* Image pairs per workspace chunk of pgx_match_batch_dev / pgx_sequence_step_dev (default 2048, [16, 4096]).  A job with
 * more pairs goes through in chunks; the chunk bounds the matcher's workspace (about 4.4 MiB per image pair at 4096 descriptors
 * a side -- 4 MiB of it the residual's distance matrix -- plus the allocator's 25 % slack: 11 GB at the default).  Chunks of 1024 pairs or more run one after the
 * other on the context's stream with ONE workspace; smaller chunks run their stages side by side on three streams with three
 * workspaces resident (the form for small-memory configurations: it hides nothing once a chunk fills the chip).
 * The per-pair finish is one workgroup per pair, so large chunks balance the CUs better.  Results do not depend on it. */
int pgx_set_match_chunk(pgx_ctx *ctx, int image_pairs_per_chunk);

/* ---- stage-granular host entry points (one reference function each) ------------------ */
/* DeWarp.ApplyDistortionMat<Rgba64> (DeWarp.cs:19-37) with the context's map. */
int pgx_dewarp(pgx_ctx *ctx, const uint16_t *rgba64, int W, int H, uint16_t *out_rgba64);
/* Matrix.Convert(Grayscale.FromRgba64) (Converters.cs:15-22, Grayscale.cs:19-23). */
int pgx_gray(pgx_ctx *ctx, const uint16_t *rgba64, int W, int H, float *out_gray);
/* KeypointDetection.Detect minus the BRIEF ctor work (KeypointDetection.cs:42-63): raster-order
 * hits.  *n_out = total hits; only min(*n_out, capacity) are written (PGX_E_CAPACITY if more). */
int pgx_fast(pgx_ctx *ctx, const float *gray, int W, int H,
             pgx_keypoint *out, int capacity, int *n_out);
/* Keypoint.GetBriefDescriptor (Keypoint.cs:29-57) for n keypoints -> desc [n][ceil(P/32)]. */
int pgx_brief(pgx_ctx *ctx, const float *gray, int W, int H,
              const pgx_keypoint *kps, int n, uint32_t *desc_out);
/* RedundantKeypointEliminator.EliminateRedundantKeypoints (:16-35): order_out[k] = index into
 * kps of the k-th accepted keypoint; *n_out = accepted count (order_out holds n entries). */
int pgx_nms(pgx_ctx *ctx, const pgx_keypoint *kps, int n, int W, int H,
            int32_t *order_out, int *n_out);
/* KeypointMatching.MatchKeypoints (KeypointMatching.cs:14-69): exactly n1 entries in the
 * reference's emission order; (0, 0, PGX_DIST_NONE) tail when n1 > n2; PGX_E_EMPTY_SET when
 * n2 == 0 < n1.  words = uint32 words per descriptor. */
int pgx_match(pgx_ctx *ctx, const uint32_t *desc1, int n1, const uint32_t *desc2, int n2,
              int words, pgx_pair *out);

/* KeypointMatching.MatchKeypoints for n_pairs image pairs in ONE call, host buffers ("Call sites": the batched
 * form a P/Invoke host with managed arrays calls instead of n_pairs x pgx_match -- one upload, one enqueue of the batched
 * matcher, one download).  descs[f] = frame f's descriptors [counts[f]][words] (may be NULL when counts[f] == 0);
 * pair_list [n_pairs][2] = (frame_a, frame_b).  out receives the lists back to back in pair order: list m has counts[a_m]
 * entries and starts at out_offsets[m] (out_offsets [n_pairs + 1], optional; the caller sizes out as the sum of counts[a_m]).
 * A pair with counts[b] == 0 < counts[a] gets counts[a] entries (0, 0, PGX_DIST_NONE) and the call returns PGX_E_EMPTY_SET
 * after all lists are written (the reference would have thrown at that pair, KeypointMatching.cs:61). */
int pgx_match_batch(pgx_ctx *ctx, const uint32_t *const *descs, const int32_t *counts, int n_frames, int words,
                    const int32_t *pair_list, int n_pairs, pgx_pair *out, int64_t *out_offsets);

/* ---- fused host entry point: dewarp -> gray -> detect -> NMS -> BRIEF for one image ---- */
/* The chain of TestService.BuildKeypointDetectorPipeline (TestService.cs:137-152).
 * kp_out [capacity], desc_out [capacity][ceil(P/32)]; *n_out survivors in NMS order,
 * *n_raw raw FAST hits. */
int pgx_detect(pgx_ctx *ctx, const uint16_t *rgba64, int W, int H,
               pgx_keypoint *kp_out, uint32_t *desc_out, int capacity, int *n_out, int *n_raw);

/* ---- device-resident batched entry points (asynchronous on the context's stream) ------ */
/* Stream hand-off contract: the context's own stream is NON-BLOCKING (it does not order against the null stream
 * or any other stream).  Every "_dev" input must be complete, and every output buffer free to be written, on the
 * context's stream when the call is made: either hand the library the stream that produced them
 * (pgx_set_stream) or synchronise the producer first.  Results are defined after pgx_check_status() (or after
 * the caller synchronises that stream). */
/* F frames [F][H][W][4] uint16 in HBM -> per frame up to `capacity` survivors.
 * d_kp [F][capacity], d_desc [F][capacity][words], d_counts [F], d_nraw [F] (all device). */
int pgx_detect_batch_dev(pgx_ctx *ctx, const uint16_t *d_rgba64, int F, int W, int H,
                         pgx_keypoint *d_kp, uint32_t *d_desc, int32_t *d_counts,
                         int32_t *d_nraw, int capacity);
/* M image pairs: d_pairlist [M][2] = (frame_a, frame_b) indexes descriptor sets
 * d_desc [F][stride][words] with d_counts [F].  d_out [M][stride]: the first counts[a] entries of
 * row m are the reference's match list for (a, b).  Pairs with counts[b] == 0 < counts[a] raise
 * PGX_E_EMPTY_SET in pgx_check_status and leave their row filled with (0,0,PGX_DIST_NONE).
 * max_count = an upper bound of every d_counts entry used (<= stride; pass stride if unknown):
 * it only sizes the launch grids, counts above it are clamped to it. */
int pgx_match_batch_dev(pgx_ctx *ctx, const uint32_t *d_desc, const int32_t *d_counts,
                        int stride, int words, const int32_t *d_pairlist, int M, int max_count,
                        pgx_pair *d_out);
/* Stage ordering between TWO contexts on one GPU (consecutive jobs kept in flight, each context on its own stream): `ctx`'s
 * stream waits until stage `stage` of `other`'s most recent call of that kind has finished on the device (no wait if it has
 * run none): PGX_STAGE_DETECT = the detect chain of pgx_detect_batch_dev; PGX_STAGE_MATCH_WIDE / _ROWS / _DONE = the whole-chip
 * distance rounds / the residual distance rows / everything of pgx_match_batch_dev (the stages of pgx_sequence_step_dev count
 * the same way).  What the measurements say ("Two jobs in flight"; bench.py --gate): the productive overlap is job
 * k + 1's detect chain BESIDE job k's distance kernel -- that kernel's 256-thread workgroups mix with the detect chain's -- so job
 * k + 1's pgx_detect_batch_dev needs NO wait at all, and its matcher is held back with
 * pgx_gate_match(ctx, other, PGX_STAGE_MATCH_ROWS) until job k's residual distance rows are written: the next distance kernel
 * then runs beside job k's per-pair finish (round 5: 7.13 ms per bench step; waiting for PGX_STAGE_MATCH_DONE 7.21, for
 * PGX_STAGE_MATCH_WIDE only -- the rows kernel beside the next distance kernel, both on the matrix pipe -- 7.49; no gate at all:
 * two distance kernels at once, slower still).  pgx_wait_stage remains for hosts that want another order.  The
 * reference has the same shape on the CPU: ApplyDistortionMat of image k + 1 runs beside Detect of image k
 * (TestService.cs:25,146-149).  Ordering only: results do not depend on it.
 * Memory: every context owns its workspaces.  The matcher's is about 5.5 MiB per image pair of a chunk at 4096 descriptors a
 * side (4 MiB of it the residual's byte matrix; allocations carry 25 % slack): 11 GB at the default chunk of 2048 pairs, so two
 * contexts in flight hold 22 GB plus their own output buffers -- sized for the 288 GB of an MI355X; pgx_set_match_chunk
 * lowers it. */
enum { PGX_STAGE_DETECT = 0, PGX_STAGE_MATCH_WIDE = 1, PGX_STAGE_MATCH_ROWS = 2, PGX_STAGE_MATCH_DONE = 3 };
int pgx_wait_stage(pgx_ctx *ctx, pgx_ctx *other, int stage);
/* The same wait, placed INSIDE ctx's next matcher call (pgx_match_batch_dev / pgx_sequence_step_dev; one shot): between its
 * init kernel -- which touches ctx's own workspace only and so runs ahead, beside whatever `other` still has on the chip -- and
 * its first whole-chip distance round, which then starts the moment the gate opens (with pgx_wait_stage in front of the call the
 * init kernel sits on the critical path: 0.2 ms per step of the bench job).  `other` must outlive that call. */
int pgx_gate_match(pgx_ctx *ctx, pgx_ctx *other, int stage);

/* ---- RANSAC fundamental matrix and camera pose, batched over image pairs (asynchronous, device pointers) --- */
/* CameraPoseEstimation.GetFundamentalMatrix (CameraPoseEstimation.cs:26-94) for M image pairs at once: `keypointPairs` of
 * image pair m = the first counts[a] entries of d_matches[m] (indices into d_kp[a] / d_kp[b], (a, b) = d_pairlist[m]); per
 * sample a subset of pairs_per_sample DISTINCT list positions, the normalised 8-point estimate (EstimateFundamentalMatrix,
 * :204-250, incl. its always-1 scale and column-major fill), the signed score (F * p2) . p1 <= threshold over the whole list
 * (:67-77); the first sample with the most inliers wins.  rank_check != 0 keeps only matrices of numerical rank 2 like
 * :46-51 (with noisy pairs that rejects nearly every sample -- the reference then throws; here d_inliers[m] = -1).
 * The reference draws subsets from an unseeded System.Random and singular vectors from MathNet's SVD: `seed` replaces the
 * former, a Jacobi eigen-solver with a fixed sign rule the latter (parity unpinned).  d_F [M][9] row-major;
 * d_inliers[m] = -1 when the list is shorter than pairs_per_sample (:31-32) or no sample qualified (:88-89).
 * pairs_per_sample < 8 -> PGX_E_BADARG (:28-29). */
int pgx_fundamental_ransac_dev(pgx_ctx *ctx, const pgx_keypoint *d_kp /* [F][stride] */, const pgx_pair *d_matches /* [M][stride] */,
                               const int32_t *d_counts /* [F] */, const int32_t *d_pairlist /* [M][2] */, int M, int stride,
                               int n_samples, int pairs_per_sample, float threshold, int rank_check, uint64_t seed,
                               float *d_F, int32_t *d_inliers, int32_t *d_best_sample);
/* CameraPoseEstimation.EstimateCameraPose (:96-202): E = K^T F K with the reference's hard-coded K, the four (R, t)
 * candidates, linear triangulation of every keypoint pair, vote on z >= 0.  d_Rt [M][12] = R row-major then t of the
 * winning candidate, d_votes [M][4], d_best [M]; d_points [M][stride][3] (or NULL) = the winner's point cloud (the input of
 * Utils.CreatePointCloud, :199). */
int pgx_pose_dev(pgx_ctx *ctx, const pgx_keypoint *d_kp, const pgx_pair *d_matches, const int32_t *d_counts,
                 const int32_t *d_pairlist, int M, int stride, const float *d_F, float *d_Rt, int32_t *d_votes,
                 int32_t *d_best, float *d_points);

/* ---- multi-GPU: one process and one context per GPU; the context owns the RCCL communicator -------- */
/* The reference handles one image pair in one process (TestService.cs:80-96) and nothing couples image pairs, so the
 * path shards with no collective inside detect or match: frame f -> rank f mod G, image pair p -> rank p mod G, and two
 * exchanges of FIXED-SIZE records (per-frame {count, descriptors}; per-pair match lists -- always exactly N1 entries,
 * KeypointMatching.cs:38), both in-place all-gathers over a rank-major buffer [G][slots][...]: global item k sits in
 * block k mod G at place k / G.  librccl is loaded on first use; without it these calls return PGX_E_RCCL. */
#define PGX_COMM_ID_BYTES 128
/* Rank 0 makes the id (ncclGetUniqueId); the host hands the 128 bytes to every rank (socket, file, MPI, ...). */
int pgx_comm_unique_id(void *id_out /* [PGX_COMM_ID_BYTES] */);
/* Collective over all ranks (ncclCommInitRank on the context's device). */
int pgx_comm_init(pgx_ctx *ctx, int rank, int world, const void *id /* [PGX_COMM_ID_BYTES] */);
int pgx_comm_destroy(pgx_ctx *ctx);
int pgx_comm_info(pgx_ctx *ctx, int *rank, int *world);   /* (0, 1) without a communicator */
/* In-place all-gather on the context's stream: this rank's record is bytes [rank * bytes_per_rank, +bytes_per_rank) of
 * d_buf [world * bytes_per_rank].  world == 1: no-op. */
int pgx_allgather_dev(pgx_ctx *ctx, void *d_buf, size_t bytes_per_rank);
/* The four phases of one job, enqueued on the context's stream: detect this rank's n_local_frames frames into its block
 * of d_desc_all [world * frame_slots][capacity][words] / d_counts_all [world * frame_slots]; all-gather both; match this
 * rank's n_local_pairs image pairs (d_pairlist_local [n][2] = SLOT indices into the gathered buffers) into its block of
 * d_out_all [world * pair_slots][capacity]; all-gather the lists.  Lists are cut to `capacity` by pgx_set_capacity's
 * survivor limit (set it <= capacity).  world == 1: the same without the collectives.
 * Errors at world > 1: every rank-local check and allocation happens before the first collective, and the first call with a
 * given set of arguments ends that part with a status exchange among the ranks: if any rank fails there, EVERY rank returns
 * (the failing one its own code, the others PGX_E_RCCL naming it) and nothing has been exchanged.  A HIP / RCCL failure later
 * in the step aborts this rank's communicator (best effort: peers blocked in a collective may not notice on one node -- the
 * host must put a time limit on its ranks) and leaves the context without one. */
int pgx_sequence_step_dev(pgx_ctx *ctx, const uint16_t *d_frames_local, int n_local_frames, int frame_slots, int W, int H,
                          pgx_keypoint *d_kp_local, uint32_t *d_desc_all, int32_t *d_counts_all, int32_t *d_nraw_local,
                          int capacity, const int32_t *d_pairlist_local, int n_local_pairs, int pair_slots,
                          pgx_pair *d_out_all);

/* ---- the track graph over the gathered match lists -------------------------------------------------- */
/* Not in the reference (TestService.cs:80-96 handles one image pair); north_star names it as the consumer of the
 * gathered lists.  What the reference does hold is the distance gate: python_src/scripts/match_keypoints.py:23,127
 * (`--match-threshold`), `new KeypointMatching(100)` in the commented code of Photogrammetry/Program.cs:165,224.
 * Semantics -- order-independent, so the device build and the sequential host build below give the same result bit for bit:
 *   nodes   (frame, keypoint) with keypoint < counts[frame]
 *   edges   entry e < counts[a] of image pair (a, b)'s list links (a, k1) with (b, k2) when dist <= max_dist; the
 *           (0, 0, PGX_DIST_NONE) tail entries (KeypointMatching.cs:40-42) never link, whatever max_dist is
 *   tracks  connected components with at least min_len nodes; a component that holds two keypoints of ONE frame is
 *           inconsistent and dropped as a whole (counted, its nodes marked -2)
 *   order   tracks by their first (frame, keypoint), nodes inside a track ascending.
 *
 * Device form (asynchronous on the context's stream, device pointers): the lists are used where the matcher / the all-gather
 * left them.  d_matches [M][stride], d_counts [F], d_pairlist [M][2] exactly as for pgx_match_batch_dev (pair m's frames are
 * SLOT indices into d_counts).  d_frame_ids [F] (or NULL = identity, n_frames = F) maps a slot to the frame NUMBER the graph
 * uses, in [0, n_frames), distinct; -1 = this slot is not part of the graph (padding slots of the rank-major gathered buffers;
 * frames of sequences another rank builds the graph for): image pairs that touch such a slot are skipped.  Outputs:
 *   d_track_of [n_frames][stride]   track index of every node; -1 = no track (beyond counts, or a component below min_len),
 *                                   -2 = dropped with its inconsistent component
 *   d_offsets  [n_frames*stride+1]  the first n_tracks + 1 entries: track t's nodes are d_nodes[d_offsets[t] .. d_offsets[t+1])
 *   d_nodes    [n_frames*stride][2] (frame, keypoint)
 *   d_summary  [8]                  n_tracks, n_nodes, dropped components, nodes in them, edges used, longest track,
 *                                   largest dropped component, 0
 * n_frames * stride <= 2^30.  min_len < 1 counts as 1. */
int pgx_tracks_dev(pgx_ctx *ctx, const pgx_pair *d_matches, const int32_t *d_counts, const int32_t *d_pairlist, int M, int F,
                   int stride, const int32_t *d_frame_ids, int n_frames, int max_dist, int min_len,
                   int32_t *d_track_of, int32_t *d_offsets, int32_t *d_nodes, int32_t *d_summary);

/* Host form, no GPU work (small inputs; a host that holds the lists in managed memory): the same semantics, sequential.
 * counts [n_frames] = keypoints per frame. */
typedef struct pgx_tracks pgx_tracks;
int  pgx_tracks_create(const int32_t *counts, int n_frames, pgx_tracks **out);
void pgx_tracks_destroy(pgx_tracks *t);
/* matches: the first n entries of one image pair's list (n = counts[frame_a]). */
int  pgx_tracks_add_pair(pgx_tracks *t, int frame_a, int frame_b, const pgx_pair *matches, int n, int max_dist);
/* Closes the graph: *n_tracks consistent components of at least min_len nodes with *n_nodes nodes in all. */
int  pgx_tracks_finish(pgx_tracks *t, int min_len, int *n_tracks, int *n_nodes);
int  pgx_tracks_get(pgx_tracks *t, int32_t *track_offsets /* [n_tracks + 1] */, int32_t *nodes /* [n_nodes][2] = (frame, keypoint) */);
/* After pgx_tracks_finish: inconsistent components and the nodes in them (what d_summary[2], [3] report on the device). */
int  pgx_tracks_dropped(pgx_tracks *t, int *n_components, int *n_nodes);

/* ---- measurement hooks (bench.py) ---------------------------------------------------- */
/* When on, the named hot kernels are bracketed by HIP events on the launch stream. */
int pgx_profile_enable(pgx_ctx *ctx, int on);
/* Bracket only the kernel group `name` (NULL or "": every group again).  Two event records per launch cost about 10 us
 * of device time each way on a busy stream (0.6 ms per step of the bench job with every launch bracketed): the timed
 * region of bench.py brackets the dominant kernel only, the untimed stand-alone pass brackets everything. */
int pgx_profile_filter(pgx_ctx *ctx, const char *name);
/* Sums since the last reset for kernel `name` ("dewarp_gray", "fast", "ham_argmin", ...):
 * launches and total milliseconds.  Synchronises the stream. */
int pgx_profile_get(pgx_ctx *ctx, const char *name, int *launches, double *total_ms);
int pgx_profile_reset(pgx_ctx *ctx);
/* When on, a multi-chunk pgx_match_batch_dev runs its stages in order on the context's stream instead of side by side on
 * the library's own streams: the event times above are then stand-alone kernel times (side by side they overlap and
 * stretch each other). */
int pgx_profile_serialize(pgx_ctx *ctx, int on);
/* Counters of the last pgx_match* call: rounds run on the all-CU distance kernel, and the
 * descriptor-pair distance evaluations those launches issued (sum over rounds and image pairs of
 * n1*n2; evaluations_round0 = the first launch alone = sum of N1*N2).  Synchronises the stream. */
int pgx_match_stats(pgx_ctx *ctx, int *rounds_wide, int64_t *evaluations, int64_t *evaluations_round0);

/* Diagnostic counters of the match tail since the last call, 64-bit (they count per image pair and step: a long run
 * overflows 32 bits): [3] queue entries, [4] matrix-row scans, [6] proposals of the per-pair finish; the others are
 * used by developer builds only.  Cleared on read.  Synchronises the stream. */
int pgx_debug_counters(pgx_ctx *ctx, int64_t *out8);

/* ---- host-side helpers (no GPU work) -------------------------------------------------- */
/* Utils.NextGaussianPair (Utils.cs:14-38) on a seeded splitmix64 stream; out [P][4]. */
int pgx_make_brief_pairs(uint64_t seed, int sigma, int P, int32_t *out);
/* DeWarp.GetDistortionMatrix (DeWarp.cs:39-107) in float64 on the host; out [H][W][2].
 * MathNet's Cubic.RealRoots is restated from its published algorithm (parity unpinned). */
int pgx_build_dewarp_map(int W, int H, const double *coeffs, int ncoeffs, int32_t *out);

#ifdef __cplusplus
}
#endif
#endif /* PGX_H */
