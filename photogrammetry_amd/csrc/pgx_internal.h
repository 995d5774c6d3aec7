// pgx_internal.h -- shared between the HIP translation units of libpgx.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <map>
#include <mutex>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "../../include/pgx.h"

// status bits set by kernels (device word 0), decoded by pgx_check_status
#define PGX_ST_OOB_SOURCE 1u
#define PGX_ST_RAW_CAP    2u
#define PGX_ST_KP_CAP     4u
#define PGX_ST_EMPTY_SET  8u
#define PGX_ST_INTERNAL   16u  /* a device loop made no progress (a bug, never an input property) */

// key = (distance << PGX_IDX_BITS) | index ; limits: index < 2^20, distance < 2^12
#define PGX_IDX_BITS 20
#define PGX_IDX_MASK ((1u << PGX_IDX_BITS) - 1u)
#define PGX_KEY_NONE 0xFFFFFFFFu
// d_status layout (ints): [0] sticky error bits, [4..4+2*PGX_MAX_WIDE_ROUNDS) u64 evaluation counters per wide round
#define PGX_MAX_WIDE_ROUNDS 8
// ... and [PGX_DBG_OFF .. +16) eight u64 diagnostic counters of the match tail (pgx_debug_counters)
#define PGX_DBG_OFF 32
// residual size (rows and columns) from which one workgroup finishes an image pair out of LDS
#define PGX_TAIL_MAX 2048
// ... and the limit when the tail workgroup has to fill its distance cache itself (descriptors staged in LDS)
#define PGX_TAIL_FILL_MAX 1024
// chunks of at least this many image pairs run in order on one stream; smaller ones through the three-stream pipeline
#define PGX_PIPELINE_BELOW 1024

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        if (p) { hipError_t e = hipFree(p); p = nullptr; cap = 0; if (e != hipSuccess) return e; }
        size_t want = bytes + bytes / 4 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

// pinned host staging (hipHostMalloc), grown on demand
struct PinBuf {
    void *p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; }
        const size_t want = bytes + bytes / 4 + 4096;
        hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
    template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

struct ProfEntry {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
    int launches = 0;
    double total_ms = 0.0;
};

struct pgx_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    std::mutex mu;
    std::string err;     // text of the last failing call of any thread (written through pgx_note_error, under err_mu)
    std::mutex err_mu;

    // configuration
    float threshold = 0.f;
    int radius = 0;
    bool params_set = false;
    int P = 0, words = 0;
    bool pairs_set = false;
    DevBuf d_pairs;
    int mapW = 0, mapH = 0;
    bool map_set = false;
    DevBuf d_map;
    int raw_cap = 1 << 17;
    int kp_cap = 1 << PGX_IDX_BITS; // soft survivor limit of the fused path (pgx_set_capacity); default: none
    // image pairs per matcher workspace chunk (pgx_set_match_chunk).  The per-pair finish is one workgroup per image pair, and
    // the pairs of a sequence differ 3:1 in how long they take: the more of them one launch holds, the better the CUs are
    // balanced (stand-alone finish of the bench job's 2016 pairs: 2.15 ms at 256 pairs per chunk, 1.70 at 512, 1.24 in one
    // chunk; whole matcher 6.58 / 6.43 / 6.00 ms).  4.4 MiB of workspace per pair (+ 25 % slack of DevBuf::ensure): 11 GB at the default.
    int match_chunk = 2048;
    int src8 = 0; // pgx_set_source_format: 1 = the rgba arguments are 8-bit RGBA

    // status words: [0] sticky error bits
    int *d_status = nullptr;
    int *h_status = nullptr; // pinned

    // detect workspaces
    DevBuf ws_gray, ws_seg, ws_segoff, ws_nraw, ws_rawxy, ws_rawscore, ws_nms, ws_order, ws_nkept;
    // host-API staging
    DevBuf st_a, st_b, st_c, st_d, st_e, st_f;
    PinBuf pin_in, pin_out; // pgx_match_batch
    // match workspaces: four, so that with several chunks of image pairs the stages of consecutive chunks run side by side
    DevBuf ws_matchn[4];
    DevBuf ws_pose, ws_tracks;
    hipStream_t mstream[4] = {nullptr, nullptr, nullptr, nullptr}; // [0] wide rounds, [1] residual distance rows, [2], [3] per-pair finishes (alternating)
    hipEvent_t ev_in = nullptr, ev_wide[4] = {nullptr, nullptr, nullptr, nullptr}, ev_rows[4] = {nullptr, nullptr, nullptr, nullptr},
               ev_fin[4] = {nullptr, nullptr, nullptr, nullptr}, ev_join[4] = {nullptr, nullptr, nullptr, nullptr};
    // recorded behind the stages of the most recent calls (pgx_wait_stage: another context's work is held back until a stage
    // of this one is done -- e.g. its detect chain until the distance rounds here are over, so that it runs beside the
    // residual rows and the per-pair finish instead); [PGX_STAGE_*]
    hipEvent_t ev_stage[4] = {nullptr, nullptr, nullptr, nullptr};
    // pgx_gate_match: the next matcher call waits for this event between its init kernel and its first distance round (one shot)
    hipEvent_t match_gate = nullptr;

    // multi-GPU: the RCCL communicator of this context's process (pgx_comm.hip); world 1 = none
    void *comm = nullptr;
    int comm_rank = 0, comm_world = 1;
    std::string comm_agreed;   // rank-symmetric arguments + cfg_epoch of the last pgx_sequence_step_dev whose local part every rank confirmed
    unsigned cfg_epoch = 0;    // bumped by every pgx_set_* call (the status exchange of pgx_sequence_step_dev runs again after one)
    unsigned long long id = 0; // unique per created context (pgx_last_error tells a new context from a dead one at the same address)
    DevBuf ws_agree;

    // profiling
    bool prof_on = false;
    std::string prof_only; // when not empty: only this kernel group is bracketed (pgx_profile_filter)
    bool prof_serial = false; // matcher stages in order on one stream (stand-alone kernel times)
    std::map<std::string, ProfEntry> prof;
    std::vector<hipEvent_t> ev_pool; // recycled timing events (creating two per launch costs more than the record itself)

    // last match stats
    int last_rounds_mfma = 0;
    long long last_evals = 0;
};

// XCD-aware block -> (frame, block-in-frame) map for batched kernels whose blocks of one frame share data.
// Workgroup ids are dealt round-robin to the 8 XCDs (observed, not contractual), so id % 8 selects the frame
// inside a group of 8 frames and one frame's working set lives in ONE XCD's 4 MiB L2.  Speed only: any
// mapping is correct.  Launch with a 1-D grid of nblk * F blocks.
#ifdef __HIPCC__
__host__ __device__ __forceinline__ unsigned long long *pgx_dbg(int *status) { return reinterpret_cast<unsigned long long *>(status + PGX_DBG_OFF); }
__device__ __forceinline__ void pgx_xcd_map(int lin, int nblk, int F, int &f, int &b)
{
    const int F8 = (F >> 3) << 3;
    if (lin < nblk * F8) {
        const int j = lin >> 3;
        f = (j / nblk) * 8 + (lin & 7);
        b = j % nblk;
    } else {
        const int r = lin - nblk * F8;
        f = F8 + r / nblk;
        b = r % nblk;
    }
}
#endif

// error text of a failing call: kept per calling thread (what pgx_last_error returns to that thread) and on the context
void pgx_note_error(pgx_ctx *c, const std::string &msg);

// RAII event bracket used by the launchers' callers
// attach = true: the scope records nothing itself; the ONE kernel launched inside it takes the two events with
// hipExtLaunchKernelGGL, which stamps the dispatch's own begin and end -- no barrier packets between back-to-back kernels
// (two hipEventRecord per launch of the distance kernel cost 2.4 % of the bench step)
struct ProfScope {
    pgx_ctx *c;
    ProfEntry *e = nullptr;
    hipEvent_t a = nullptr, b = nullptr;
    hipStream_t st;
    bool attach;
    ProfScope(pgx_ctx *ctx, const char *name, hipStream_t s = nullptr, bool attach_to_kernel = false)
        : c(ctx), st(s ? s : ctx->stream), attach(attach_to_kernel)
    {
        if (!c->prof_on || (!c->prof_only.empty() && c->prof_only != name)) return;
        e = &c->prof[name];
        auto take = [&](hipEvent_t &ev) {
            if (!c->ev_pool.empty()) { ev = c->ev_pool.back(); c->ev_pool.pop_back(); return true; }
            return hipEventCreate(&ev) == hipSuccess;
        };
        if (!take(a) || !take(b)) { e = nullptr; a = b = nullptr; return; }
        if (!attach) (void)hipEventRecord(a, st);
    }
    ~ProfScope()
    {
        if (!e) return;
        if (!attach) (void)hipEventRecord(b, st);
        e->pending.emplace_back(a, b);
    }
};

// ---------------------------------------------------------------------------------------
// kernel launchers (each .hip file owns its kernels)
// ---------------------------------------------------------------------------------------

// k_image.hip
// rgba: [F][H][W] pixels of 4 x uint16 (src8 = 0) or 4 x uint8 (src8 = 1, widened x257 in registers)
void pgx_launch_dewarp_gray(hipStream_t s, const void *rgba, int src8, const int32_t *map_uv, int F, int W, int H,
                            float *gray, uint16_t *dewarped, int *status);
void pgx_launch_dewarp_map(hipStream_t s, int W, int H, const double *k /*[5]*/, int32_t *map_uv, int *status);

// k_fast.hip
size_t pgx_fast_seg_count(int W, int H);   // segments per frame
void pgx_launch_fast(hipStream_t s, const float *gray, int F, int W, int H, float T,
                     unsigned long long *seg /*[F][nseg][4]*/, uint32_t *segoff /*[F][nseg]*/,
                     int32_t *n_raw /*[F]*/, uint32_t *raw_xy /*[F][raw_cap]*/,
                     int32_t *raw_score /*[F][raw_cap]*/, int raw_cap, int *status,
                     bool compact = true /* false: planes and ranks only, raw_xy/raw_score are not written */);

// k_nms.hip
size_t pgx_nms_ws_bytes(int W, int H, int radius, int n_cap, bool planes);
// true when pgx_launch_nms with planes never reads raw_xy/raw_score and fills in the kept points' entries itself
bool pgx_nms_fills_raw_lists(int W, int H, int radius, int n_cap);  // per frame; planes: the fused detect path
void pgx_launch_nms(hipStream_t s, uint32_t *raw_xy, int32_t *raw_score, const int32_t *n_raw,
                    int F, int n_cap, int W, int H, int radius, void *ws, size_t ws_stride,
                    uint32_t *order /*[F][kp_cap]*/, int32_t *n_kept /*[F]*/, int kp_cap, int *status,
                    const unsigned long long *seg = nullptr /* FAST planes: enables atomic-free binning */,
                    const uint32_t *segoff = nullptr,
                    bool kp_soft = false /* true: kp_cap is the caller's soft limit, lists are cut without PGX_ST_KP_CAP */);
// stage API (one list, host waits): whole-chip rounds until nothing is left undecided, see k_nms.hip
hipError_t pgx_launch_nms_sync(hipStream_t s, const uint32_t *raw_xy, const int32_t *raw_score, const int32_t *n_raw,
                               int n_cap, int W, int H, int radius, void *ws, size_t ws_stride, uint32_t *order,
                               int32_t *n_kept, int kp_cap, int *status);

// k_brief.hip
void pgx_launch_brief(hipStream_t s, const float *gray, int F, int W, int H,
                      const uint32_t *raw_xy, const int32_t *raw_score, int raw_cap,
                      const uint32_t *order, const int32_t *n_kept, int kp_cap /* order stride and list bound */,
                      const int32_t *pairs, int P,
                      pgx_keypoint *kp_out, uint32_t *desc_out, int32_t *counts_out, int out_stride /* slots per frame in kp_out/desc_out */);
// descriptors for an explicit keypoint list (stage API)
void pgx_launch_brief_list(hipStream_t s, const float *gray, int W, int H, const pgx_keypoint *kps, int n,
                           const int32_t *pairs, int P, uint32_t *desc_out);

// k_pose.hip
size_t pgx_pose_ws_bytes(int M, int n_samples);
void pgx_launch_fundamental(hipStream_t s, const pgx_keypoint *kp, const pgx_pair *matches, const int32_t *counts,
                            const int32_t *pairlist, int M, int stride, int n_samples, int P, float threshold, int rank_check,
                            uint64_t seed, void *ws, float *F_out, int32_t *inliers, int32_t *best_sample);
void pgx_launch_pose(hipStream_t s, const pgx_keypoint *kp, const pgx_pair *matches, const int32_t *counts,
                     const int32_t *pairlist, int M, int stride, const float *F_in, float *Rt_out, int32_t *votes,
                     int32_t *best, float *points);

// k_tracks.hip
size_t pgx_tracks_ws_bytes(int n_frames, int stride);
void pgx_launch_tracks(hipStream_t s, const pgx_pair *d_matches, const int32_t *d_counts, const int32_t *d_pairlist, int M, int F,
                       int stride, const int32_t *d_frame_ids, int n_frames, int max_dist, int min_len, void *ws,
                       int32_t *d_track_of, int32_t *d_offsets, int32_t *d_nodes, int32_t *d_summary);

// k_match.hip
struct MatchPlan {
    int M;        // image pairs
    int stride;   // descriptor slots per frame
    int words;
    int max_n;    // upper bound of any count (<= stride)
    int rounds_mfma;
    int skip_below = PGX_TAIL_FILL_MAX; // a wide round leaves image pairs alone whose residual (rows and columns) is at most this
};
size_t pgx_match_ws_bytes(int M, int stride);
// the three stages of one chunk of image pairs -- wide part (init, whole-chip mutual-nearest rounds), residual distance
// rows (256-bit descriptors only), per-pair finish; they may run on different streams (the caller orders them with events)
// gate: an event the stream waits for between the init kernel and the first distance round (or nullptr)
void pgx_launch_match_wide(pgx_ctx *ctx, hipStream_t s, const uint32_t *d_desc, const int32_t *d_counts,
                           const int32_t *d_pairlist, const MatchPlan &plan, void *ws, int *status, hipEvent_t gate = nullptr);
void pgx_launch_match_rows(pgx_ctx *ctx, hipStream_t s, const uint32_t *d_desc, const int32_t *d_pairlist,
                           const MatchPlan &plan, void *ws, int *status);
void pgx_launch_match_finish(pgx_ctx *ctx, hipStream_t s, const uint32_t *d_desc, const int32_t *d_pairlist,
                             const MatchPlan &plan, void *ws, pgx_pair *d_out, int *status);
