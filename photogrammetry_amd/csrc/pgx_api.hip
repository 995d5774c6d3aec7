// pgx_api.hip -- the C ABI of libpgx.so (see include/pgx.h): context, configuration,
// host-buffer entry points (copy in -> kernels -> copy out) and the device-resident batched
// entry points.  All compute is in the k_*.hip kernels; there is no CPU fallback anywhere:
// every entry point needs a working gfx950 device and fails with PGX_E_HIP without one.
#include "pgx_internal.h"

#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <new>
#include <cstring>
#include <cstdlib>

#define PGX_VERSION_STR "pgx 0.1 (gfx950)"

static thread_local std::string tl_err;
static thread_local const pgx_ctx *tl_err_ctx = nullptr;
static thread_local unsigned long long tl_err_id = 0;   // the context's id: a new context at a dead one's address is not it
static std::atomic<unsigned long long> g_ctx_ids{0};

void pgx_note_error(pgx_ctx *c, const std::string &msg)
{
    tl_err = msg;
    tl_err_ctx = c;
    tl_err_id = c ? c->id : 0;
    if (c) {
        std::lock_guard<std::mutex> g(c->err_mu);
        c->err = msg;
    }
}

namespace {

int fail(pgx_ctx *c, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    pgx_note_error(c, buf);
    return code;
}

#define HIPCHK(c, expr)                                                                           \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) return fail((c), PGX_E_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

struct Lock {
    std::lock_guard<std::mutex> g;
    explicit Lock(pgx_ctx *c) : g(c->mu) { (void)hipSetDevice(c->device); }
};

int decode_status(pgx_ctx *c, int bits)
{
    if (bits & PGX_ST_OOB_SOURCE)
        return fail(c, PGX_E_OOB_SOURCE, "dewarp map points outside the source image (IndexOutOfRangeException)");
    if (bits & PGX_ST_EMPTY_SET)
        return fail(c, PGX_E_EMPTY_SET, "keypoints2 is empty while keypoints1 is not (ArgumentOutOfRangeException)");
    if (bits & PGX_ST_RAW_CAP)
        return fail(c, PGX_E_CAPACITY, "raw FAST hits exceed max_raw_per_frame (pgx_set_capacity)");
    if (bits & PGX_ST_KP_CAP) return fail(c, PGX_E_CAPACITY, "NMS survivors exceed the output capacity");
    if (bits & PGX_ST_INTERNAL) return fail(c, PGX_E_HIP, "internal error: a device-side loop stopped without progress");
    return PGX_OK;
}

// wait for the stream, read and clear the sticky status word
int sync_status(pgx_ctx *c)
{
    HIPCHK(c, hipMemcpyAsync(c->h_status, c->d_status, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_status, 0, sizeof(int), c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipGetLastError());
    return decode_status(c, c->h_status[0]);
}

// every check and workspace allocation of the detect chain (no kernel launch): pgx_sequence_step_dev calls it before its
// first collective so that no rank can fail locally between two collectives
int prepare_detect(pgx_ctx *c, int F, int W, int H, int cap)
{
    if (!c->params_set) return fail(c, PGX_E_NOT_CONFIGURED, "pgx_set_detect_params not called");
    if (!c->pairs_set) return fail(c, PGX_E_NOT_CONFIGURED, "pgx_set_brief_pairs not called");
    if (W <= 0 || H <= 0 || W > 65535 || H > 65535) return fail(c, PGX_E_BADARG, "dimensions must fit ushort");
    if (c->map_set && (c->mapW != W || c->mapH != H))
        return fail(c, PGX_E_DIM_MISMATCH, "image %dx%d vs dewarp map %dx%d (ArgumentException)", W, H, c->mapW, c->mapH);
    if (F <= 0) return PGX_OK;
    const size_t npix = (size_t)W * H;
    const size_t nseg = pgx_fast_seg_count(W, H);
    const int raw_cap = c->raw_cap;
    HIPCHK(c, c->ws_gray.ensure((size_t)F * npix * 4));
    HIPCHK(c, c->ws_seg.ensure((size_t)F * nseg * 32));
    HIPCHK(c, c->ws_segoff.ensure((size_t)F * nseg * 4));
    HIPCHK(c, c->ws_rawxy.ensure((size_t)F * raw_cap * 4));
    HIPCHK(c, c->ws_rawscore.ensure((size_t)F * raw_cap * 4));
    const size_t nms_stride = pgx_nms_ws_bytes(W, H, c->radius, raw_cap, true);
    HIPCHK(c, c->ws_nms.ensure((size_t)F * nms_stride));
    const int kp_eff = c->kp_cap <= cap ? c->kp_cap : cap;
    HIPCHK(c, c->ws_order.ensure((size_t)F * kp_eff * 4));
    HIPCHK(c, c->ws_nkept.ensure((size_t)F * 4));
    return PGX_OK;
}

// detect chain on device-resident frames (enqueue only)
int enqueue_detect(pgx_ctx *c, const uint16_t *d_rgba, int F, int W, int H, pgx_keypoint *d_kp, uint32_t *d_desc,
                   int32_t *d_counts, int32_t *d_nraw, int cap)
{
    const int rcp = prepare_detect(c, F, W, H, cap);
    if (rcp != PGX_OK) return rcp;
    if (F <= 0) return PGX_OK;
    const int raw_cap = c->raw_cap;
    const size_t nms_stride = pgx_nms_ws_bytes(W, H, c->radius, raw_cap, true);
    // pgx_set_capacity's survivor limit: lists are cut to their first kp_cap entries (NMS order) without an error;
    // only an overflow of the caller's own `cap` raises PGX_E_CAPACITY
    const bool kp_soft = c->kp_cap <= cap;
    const int kp_eff = kp_soft ? c->kp_cap : cap;

    float *gray = c->ws_gray.as<float>();
    {
        ProfScope ps(c, "dewarp_gray");
        pgx_launch_dewarp_gray(c->stream, d_rgba, c->src8, c->map_set ? c->d_map.as<int32_t>() : nullptr, F, W, H, gray,
                               nullptr, c->d_status);
    }
    {
        ProfScope ps(c, "fast");
        pgx_launch_fast(c->stream, gray, F, W, H, c->threshold, c->ws_seg.as<unsigned long long>(),
                        c->ws_segoff.as<uint32_t>(), d_nraw, c->ws_rawxy.as<uint32_t>(),
                        c->ws_rawscore.as<int32_t>(), raw_cap, c->d_status,
                        !pgx_nms_fills_raw_lists(W, H, c->radius, raw_cap));
    }
    {
        ProfScope ps(c, "nms");
        pgx_launch_nms(c->stream, c->ws_rawxy.as<uint32_t>(), c->ws_rawscore.as<int32_t>(), d_nraw, F, raw_cap, W, H,
                       c->radius, c->ws_nms.p, nms_stride, c->ws_order.as<uint32_t>(), c->ws_nkept.as<int32_t>(),
                       kp_eff, c->d_status, c->ws_seg.as<unsigned long long>(), c->ws_segoff.as<uint32_t>(), kp_soft);
    }
    {
        ProfScope ps(c, "brief");
        pgx_launch_brief(c->stream, gray, F, W, H, c->ws_rawxy.as<uint32_t>(), c->ws_rawscore.as<int32_t>(), raw_cap,
                         c->ws_order.as<uint32_t>(), c->ws_nkept.as<int32_t>(), kp_eff, c->d_pairs.as<int32_t>(), c->P,
                         d_kp, d_desc, d_counts, cap);
    }
    HIPCHK(c, hipEventRecord(c->ev_stage[PGX_STAGE_DETECT], c->stream));
    HIPCHK(c, hipGetLastError());
    return PGX_OK;
}

// checks and workspace allocation of the matcher for M image pairs of `stride` descriptor slots (no kernel launch)
int prepare_match(pgx_ctx *c, int stride, int words, int M)
{
    if (M <= 0) return PGX_OK;
    if (stride <= 0 || stride > (1 << PGX_IDX_BITS)) return fail(c, PGX_E_BADARG, "stride must be in [1, 2^20]");
    if (words <= 0 || words > 127) return fail(c, PGX_E_BADARG, "words must be in [1, 127] (P <= 4064)");
    const int CHUNK = c->match_chunk;
    const int mc = M < CHUNK ? M : CHUNK;
    const int nws = (M <= CHUNK || c->prof_serial || CHUNK >= PGX_PIPELINE_BELOW) ? 1 : 3;
    for (int k = 0; k < nws; k++) HIPCHK(c, c->ws_matchn[k].ensure(pgx_match_ws_bytes(mc, stride)));
    return PGX_OK;
}

int enqueue_match(pgx_ctx *c, const uint32_t *d_desc, const int32_t *d_counts, int stride, int words,
                  const int32_t *d_pairlist, int M, int max_n, pgx_pair *d_out)
{
    // pgx_gate_match is one shot on the NEXT matcher call, whatever becomes of that call: taken and cleared before any
    // early return, so that a gate never stays armed for some later call (its event belongs to another context)
    hipEvent_t gate = c->match_gate;
    c->match_gate = nullptr;
    if (M <= 0) return PGX_OK;
    const int rcp = prepare_match(c, stride, words, M);
    if (rcp != PGX_OK) return rcp;
    // image pairs go through in chunks so the per-pair workspace stays bounded
    const int CHUNK = c->match_chunk;
    MatchPlan plan;
    plan.stride = stride; plan.words = words;
    plan.max_n = max_n > stride ? stride : (max_n < 1 ? 1 : max_n);
    // all-CU rounds until the residual fits the per-pair tail (random data halves per round; each launch
    // skips image pairs that already fit, so extra rounds only cost their launch)
    plan.rounds_mfma = 0;
    // 256-bit descriptors: k_tail_rows caches the residual's distance rows up to PGX_TAIL_MAX, so wide rounds stop there;
    // other lengths: the tail workgroup stages the descriptors itself (PGX_TAIL_FILL_MAX)
    plan.skip_below = words == 8 ? PGX_TAIL_MAX : PGX_TAIL_FILL_MAX;
    // One or two image pairs (a pgx_match call): the per-pair finish is one workgroup per pair and leaves the chip idle, so the
    // whole-chip rounds go on to half that size first.  Measured at N = 4096 (tools/call_latency.py): random descriptor sets
    // 0.33 -> 0.27 ms per call, detect-chain sets of translated frames 0.29 -> 0.30 (their finish is bound by the number of
    // its rounds, not by the residual's size), true-match sets unchanged.  From 8 pairs per call on the lower threshold costs
    // more than it saves on the frame sets (+6 %), and at full chunks +0.6 ms per bench step: those keep PGX_TAIL_MAX.
    if (words == 8 && M <= 2) plan.skip_below = PGX_TAIL_MAX / 2;
    for (int n = plan.max_n; n > plan.skip_below && plan.rounds_mfma < PGX_MAX_WIDE_ROUNDS; n = (n + 1) / 2) plan.rounds_mfma++;
    if (plan.rounds_mfma > 0 && plan.rounds_mfma < PGX_MAX_WIDE_ROUNDS) plan.rounds_mfma++;
    HIPCHK(c, hipMemsetAsync(c->d_status + 4, 0, PGX_MAX_WIDE_ROUNDS * 8, c->stream));
    // Chunks of PGX_PIPELINE_BELOW image pairs or more go through in order on the context's stream, one workspace: every
    // stage of the matcher is bound by vector-instruction issue, so running the stages of consecutive chunks side by side
    // buys nothing (round 4: 6.43 ms side by side against 6.25 in order at 1024 pairs per chunk), while a large chunk gives
    // the per-pair finish several workgroups per CU to balance (2016 pairs in one chunk: 6.0 ms).
    if (M <= CHUNK || c->prof_serial || CHUNK >= PGX_PIPELINE_BELOW) { // everything in order on the context's stream
        for (int m0 = 0; m0 < M; m0 += CHUNK) {
            plan.M = (M - m0 < CHUNK) ? M - m0 : CHUNK;
            const int32_t *pl = d_pairlist + 2 * (size_t)m0;
            pgx_launch_match_wide(c, c->stream, d_desc, d_counts, pl, plan, c->ws_matchn[0].p, c->d_status, m0 == 0 ? gate : nullptr);
            const bool last = m0 + CHUNK >= M; // pgx_wait_stage: the stages of the last chunk stand for the call
            if (last) HIPCHK(c, hipEventRecord(c->ev_stage[PGX_STAGE_MATCH_WIDE], c->stream));
            pgx_launch_match_rows(c, c->stream, d_desc, pl, plan, c->ws_matchn[0].p, c->d_status);
            if (last) HIPCHK(c, hipEventRecord(c->ev_stage[PGX_STAGE_MATCH_ROWS], c->stream));
            pgx_launch_match_finish(c, c->stream, d_desc, pl, plan, c->ws_matchn[0].p, d_out + (size_t)m0 * stride, c->d_status);
        }
    } else {
        // Several chunks: a three-stage pipeline over chunks on three streams -- the whole-chip mutual-nearest rounds
        // (matrix pipe) of chunk i + 2 beside the residual distance rows (matrix pipe + stores) of chunk i + 1 beside the
        // per-pair finish of chunk i (latency-bound, one small workgroup per pair on half of the CUs).  Three workspaces
        // rotate; events order "inputs ready -> wide(i) -> rows(i) -> finish(i) -> wide(i + 3)".  The streams sit in
        // different stream-priority classes where the device has three: streams of one class share a small pool of hardware
        // queues round-robin, and two streams that land on one queue run strictly in order.  (Two finishes in flight on
        // alternating streams were measured too: 15.3 ms per step of config 3 against 12.6 -- their registers and LDS
        // crowd the matrix kernels out of every CU.)
        const int NS = 3;
        int lo = 0, hi = 0;
        HIPCHK(c, hipDeviceGetStreamPriorityRange(&lo, &hi)); // lo = least priority (numerically greatest)
        for (int k = 0; k < NS; k++) {
            if (!c->mstream[k]) {
                const int prio = k == 0 ? lo : (k == 2 ? hi : (lo + hi) / 2);
                HIPCHK(c, hipStreamCreateWithPriority(&c->mstream[k], hipStreamNonBlocking, prio));
            }
            if (!c->ev_wide[k]) HIPCHK(c, hipEventCreateWithFlags(&c->ev_wide[k], hipEventDisableTiming));
            if (!c->ev_rows[k]) HIPCHK(c, hipEventCreateWithFlags(&c->ev_rows[k], hipEventDisableTiming));
            if (!c->ev_fin[k]) HIPCHK(c, hipEventCreateWithFlags(&c->ev_fin[k], hipEventDisableTiming));
            if (!c->ev_join[k]) HIPCHK(c, hipEventCreateWithFlags(&c->ev_join[k], hipEventDisableTiming));
        }
        if (!c->ev_in) HIPCHK(c, hipEventCreateWithFlags(&c->ev_in, hipEventDisableTiming));
        hipStream_t sw = c->mstream[0], sr = c->mstream[1];
        if (gate) HIPCHK(c, hipStreamWaitEvent(c->stream, gate, 0)); // the three-stream form takes the gate in front of everything
        HIPCHK(c, hipEventRecord(c->ev_in, c->stream));
        for (int k = 0; k < NS; k++) HIPCHK(c, hipStreamWaitEvent(c->mstream[k], c->ev_in, 0));
        int i = 0;
        for (int m0 = 0; m0 < M; m0 += CHUNK, i++) {
            const int b = i % NS;
            hipStream_t sf = c->mstream[2];
            void *ws = c->ws_matchn[b].p;
            const int32_t *pl = d_pairlist + 2 * (size_t)m0;
            plan.M = (M - m0 < CHUNK) ? M - m0 : CHUNK;
            if (i >= NS) HIPCHK(c, hipStreamWaitEvent(sw, c->ev_fin[b], 0)); // this workspace's previous chunk is finished
            pgx_launch_match_wide(c, sw, d_desc, d_counts, pl, plan, ws, c->d_status);
            HIPCHK(c, hipEventRecord(c->ev_wide[b], sw));
            if (m0 + CHUNK >= M) HIPCHK(c, hipEventRecord(c->ev_stage[PGX_STAGE_MATCH_WIDE], sw));
            HIPCHK(c, hipStreamWaitEvent(sr, c->ev_wide[b], 0));
            pgx_launch_match_rows(c, sr, d_desc, pl, plan, ws, c->d_status);
            HIPCHK(c, hipEventRecord(c->ev_rows[b], sr));
            if (m0 + CHUNK >= M) HIPCHK(c, hipEventRecord(c->ev_stage[PGX_STAGE_MATCH_ROWS], sr));
            HIPCHK(c, hipStreamWaitEvent(sf, c->ev_rows[b], 0));
            pgx_launch_match_finish(c, sf, d_desc, pl, plan, ws, d_out + (size_t)m0 * stride, c->d_status);
            HIPCHK(c, hipEventRecord(c->ev_fin[b], sf));
        }
        for (int k = 0; k < NS; k++) {
            HIPCHK(c, hipEventRecord(c->ev_join[k], c->mstream[k]));
            HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_join[k], 0));
        }
    }
    c->last_rounds_mfma = plan.rounds_mfma;
    HIPCHK(c, hipEventRecord(c->ev_stage[PGX_STAGE_MATCH_DONE], c->stream));
    HIPCHK(c, hipGetLastError());
    return PGX_OK;
}

} // namespace

// used by pgx_comm.hip (the caller holds the context's mutex)
int pgx_enqueue_detect(pgx_ctx *c, const uint16_t *d_rgba, int F, int W, int H, pgx_keypoint *d_kp, uint32_t *d_desc,
                       int32_t *d_counts, int32_t *d_nraw, int cap)
{
    return enqueue_detect(c, d_rgba, F, W, H, d_kp, d_desc, d_counts, d_nraw, cap);
}
int pgx_enqueue_match(pgx_ctx *c, const uint32_t *d_desc, const int32_t *d_counts, int stride, int words,
                      const int32_t *d_pairlist, int M, int max_n, pgx_pair *d_out)
{
    return enqueue_match(c, d_desc, d_counts, stride, words, d_pairlist, M, max_n, d_out);
}
int pgx_prepare_detect(pgx_ctx *c, int F, int W, int H, int cap) { return prepare_detect(c, F, W, H, cap); }
int pgx_prepare_match(pgx_ctx *c, int stride, int words, int M) { return prepare_match(c, stride, words, M); }

extern "C" {

const char *pgx_version(void) { return PGX_VERSION_STR; }

int pgx_ctx_create(int device, pgx_ctx **out)
{
    if (!out) return PGX_E_BADARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return PGX_E_HIP;
    if (hipSetDevice(device) != hipSuccess) return PGX_E_HIP;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return PGX_E_HIP;
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        fprintf(stderr, "pgx: device %d is %s; this library is built for gfx950 only\n", device, prop.gcnArchName);
        return PGX_E_HIP;
    }
    pgx_ctx *c = new (std::nothrow) pgx_ctx();
    if (!c) return PGX_E_HIP;
    c->device = device;
    c->id = ++g_ctx_ids;
    if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_stage[0], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_stage[1], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_stage[2], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_stage[3], hipEventDisableTiming) != hipSuccess ||
        hipMalloc(reinterpret_cast<void **>(&c->d_status), 256) != hipSuccess ||
        hipHostMalloc(reinterpret_cast<void **>(&c->h_status), 64, hipHostMallocDefault) != hipSuccess ||
        hipMemset(c->d_status, 0, 256) != hipSuccess) {
        pgx_ctx_destroy(c);
        return PGX_E_HIP;
    }
    c->stream = c->own_stream;
    *out = c;
    return PGX_OK;
}

void pgx_ctx_destroy(pgx_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->comm) (void)pgx_comm_destroy(c);
    for (auto &kv : c->prof)
        for (auto &ev : kv.second.pending) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
    for (hipEvent_t ev : c->ev_pool) (void)hipEventDestroy(ev);
    DevBuf *bufs[] = {&c->d_pairs, &c->d_map, &c->ws_gray, &c->ws_seg, &c->ws_segoff, &c->ws_nraw, &c->ws_rawxy,
                      &c->ws_rawscore, &c->ws_nms, &c->ws_order, &c->ws_nkept, &c->st_a, &c->st_b, &c->st_c,
                      &c->st_d, &c->st_e, &c->st_f, &c->ws_pose, &c->ws_tracks, &c->ws_agree, &c->ws_matchn[0], &c->ws_matchn[1], &c->ws_matchn[2], &c->ws_matchn[3]};
    for (DevBuf *b : bufs) b->release();
    c->pin_in.release();
    c->pin_out.release();
    if (c->d_status) (void)hipFree(c->d_status);
    if (c->h_status) (void)hipHostFree(c->h_status);
    for (int k = 0; k < 4; k++) {
        if (c->mstream[k]) (void)hipStreamDestroy(c->mstream[k]);
        if (c->ev_wide[k]) (void)hipEventDestroy(c->ev_wide[k]);
        if (c->ev_rows[k]) (void)hipEventDestroy(c->ev_rows[k]);
        if (c->ev_fin[k]) (void)hipEventDestroy(c->ev_fin[k]);
        if (c->ev_join[k]) (void)hipEventDestroy(c->ev_join[k]);
    }
    if (c->ev_in) (void)hipEventDestroy(c->ev_in);
    for (int k = 0; k < 4; k++) if (c->ev_stage[k]) (void)hipEventDestroy(c->ev_stage[k]);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

const char *pgx_last_error(pgx_ctx *c)
{
    if (!c) return "null context";
    if (tl_err_ctx == c && tl_err_id == c->id) return tl_err.c_str();   // this thread's own last failure on this context
    static thread_local std::string copy;         // another thread's: a private copy, valid until this thread asks again
    {
        std::lock_guard<std::mutex> g(c->err_mu);
        copy = c->err;
    }
    return copy.c_str();
}

int pgx_set_stream(pgx_ctx *c, void *hip_stream)
{
    if (!c) return PGX_E_BADARG;
    Lock l(c);
    c->cfg_epoch++;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->stream = hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : c->own_stream;
    return PGX_OK;
}

int pgx_check_status(pgx_ctx *c)
{
    if (!c) return PGX_E_BADARG;
    Lock l(c);
    return sync_status(c);
}

int pgx_set_dewarp_map(pgx_ctx *c, const int32_t *uv, int W, int H)
{
    if (!c) return PGX_E_BADARG;
    Lock l(c);
    c->cfg_epoch++;
    if (!uv) { c->map_set = false; c->mapW = c->mapH = 0; return PGX_OK; }
    if (W <= 0 || H <= 0 || W > 65535 || H > 65535) return fail(c, PGX_E_BADARG, "map dimensions must fit ushort");
    const size_t bytes = (size_t)W * H * 8;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, c->d_map.ensure(bytes + 32));
    HIPCHK(c, hipMemcpy(c->d_map.p, uv, bytes, hipMemcpyHostToDevice));
    c->mapW = W; c->mapH = H; c->map_set = true;
    return PGX_OK;
}

int pgx_set_dewarp_coeffs(pgx_ctx *c, int W, int H, const double *coeffs, int ncoeffs)
{
    if (!c || !coeffs) return c ? fail(c, PGX_E_BADARG, "null pointer") : PGX_E_BADARG;
    Lock l(c);
    c->cfg_epoch++;
    if (ncoeffs != 5) return fail(c, PGX_E_BADARG, "You must pass exactly 5 distortion coefficients (ArgumentException)"); // DeWarp.cs:46-48
    if (W <= 0 || H <= 0 || W > 65535 || H > 65535) return fail(c, PGX_E_BADARG, "map dimensions must fit ushort");
    const size_t bytes = (size_t)W * H * 8;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, c->d_map.ensure(bytes + 32));
    pgx_launch_dewarp_map(c->stream, W, H, coeffs, c->d_map.as<int32_t>(), c->d_status);
    HIPCHK(c, hipGetLastError());
    c->mapW = W; c->mapH = H; c->map_set = true;
    return sync_status(c);
}

int pgx_get_dewarp_map(pgx_ctx *c, int32_t *uv_out, int W, int H)
{
    if (!c || !uv_out) return c ? fail(c, PGX_E_BADARG, "null pointer") : PGX_E_BADARG;
    Lock l(c);
    if (!c->map_set) return fail(c, PGX_E_NOT_CONFIGURED, "no dewarp map is set");
    if (W != c->mapW || H != c->mapH) return fail(c, PGX_E_DIM_MISMATCH, "map is %dx%d, asked for %dx%d", c->mapW, c->mapH, W, H);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(uv_out, c->d_map.p, (size_t)W * H * 8, hipMemcpyDeviceToHost));
    return PGX_OK;
}

int pgx_set_brief_pairs(pgx_ctx *c, const int32_t *pairs, int P)
{
    if (!c) return PGX_E_BADARG;
    Lock l(c);
    c->cfg_epoch++;
    if (!pairs || P <= 0 || P > 4064) return fail(c, PGX_E_BADARG, "P must be in [1, 4064]");
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, c->d_pairs.ensure((size_t)P * 16));
    HIPCHK(c, hipMemcpy(c->d_pairs.p, pairs, (size_t)P * 16, hipMemcpyHostToDevice));
    c->P = P; c->words = (P + 31) / 32; c->pairs_set = true;
    return PGX_OK;
}

int pgx_set_detect_params(pgx_ctx *c, float threshold, int suppression_radius)
{
    if (!c) return PGX_E_BADARG;
    Lock l(c);
    c->cfg_epoch++;
    c->threshold = threshold; c->radius = suppression_radius; c->params_set = true;
    return PGX_OK;
}

int pgx_set_match_chunk(pgx_ctx *c, int pairs)
{
    if (!c) return PGX_E_BADARG;
    Lock l(c);
    c->cfg_epoch++;
    if (pairs < 16 || pairs > 4096) return fail(c, PGX_E_BADARG, "image pairs per chunk must be in [16, 4096]");
    c->match_chunk = pairs;
    return PGX_OK;
}

int pgx_wait_stage(pgx_ctx *c, pgx_ctx *other, int stage)
{
    if (!c || !other) return c ? fail(c, PGX_E_BADARG, "null context") : PGX_E_BADARG;
    Lock l(c);
    if (stage < 0 || stage > PGX_STAGE_MATCH_DONE) return fail(c, PGX_E_BADARG, "unknown stage %d", stage);
    if (c == other) return PGX_OK; // a stream is in order with itself
    if (c->device != other->device) return fail(c, PGX_E_BADARG, "contexts on different devices (%d, %d)", c->device, other->device);
    // other->ev_stage[] are created with the context and never replaced: no lock on `other` (taking two context mutexes
    // here could deadlock against a thread that calls the two contexts the other way round); an event that has not been
    // recorded yet does not hold the stream
    HIPCHK(c, hipStreamWaitEvent(c->stream, other->ev_stage[stage], 0));
    return PGX_OK;
}

int pgx_gate_match(pgx_ctx *c, pgx_ctx *other, int stage)
{
    if (!c || !other) return c ? fail(c, PGX_E_BADARG, "null context") : PGX_E_BADARG;
    Lock l(c);
    if (stage < 0 || stage > PGX_STAGE_MATCH_DONE) return fail(c, PGX_E_BADARG, "unknown stage %d", stage);
    if (c->device != other->device) return fail(c, PGX_E_BADARG, "contexts on different devices (%d, %d)", c->device, other->device);
    c->match_gate = c == other ? nullptr : other->ev_stage[stage]; // events live as long as their context (see pgx_wait_stage)
    return PGX_OK;
}

int pgx_set_source_format(pgx_ctx *c, int format)
{
    if (!c) return PGX_E_BADARG;
    Lock l(c);
    c->cfg_epoch++;
    if (format != PGX_SRC_RGBA64 && format != PGX_SRC_RGBA8) return fail(c, PGX_E_BADARG, "unknown source format %d", format);
    c->src8 = format == PGX_SRC_RGBA8;
    return PGX_OK;
}

int pgx_set_capacity(pgx_ctx *c, int max_raw, int max_kp)
{
    if (!c) return PGX_E_BADARG;
    Lock l(c);
    c->cfg_epoch++;
    if (max_raw <= 0 || max_kp <= 0 || max_kp > (1 << PGX_IDX_BITS)) return fail(c, PGX_E_BADARG, "bad capacity");
    c->raw_cap = max_raw; c->kp_cap = max_kp;
    return PGX_OK;
}

// ---- stage-granular host entry points ---------------------------------------------------

int pgx_dewarp(pgx_ctx *c, const uint16_t *rgba, int W, int H, uint16_t *out)
{
    if (!c || !rgba || !out) return c ? fail(c, PGX_E_BADARG, "null pointer") : PGX_E_BADARG;
    Lock l(c);
    if (!c->map_set) return fail(c, PGX_E_NOT_CONFIGURED, "pgx_set_dewarp_map not called");
    if (W != c->mapW || H != c->mapH)   // DeWarp.cs:22-23
        return fail(c, PGX_E_DIM_MISMATCH, "image %dx%d vs dewarp map %dx%d (ArgumentException)", W, H, c->mapW, c->mapH);
    const size_t bytes = (size_t)W * H * 8, in_bytes = (size_t)W * H * (c->src8 ? 4 : 8);
    HIPCHK(c, c->st_a.ensure(bytes + 32));
    HIPCHK(c, c->st_b.ensure(bytes + 32));
    HIPCHK(c, hipMemcpyAsync(c->st_a.p, rgba, in_bytes, hipMemcpyHostToDevice, c->stream));
    pgx_launch_dewarp_gray(c->stream, c->st_a.p, c->src8, c->d_map.as<int32_t>(), 1, W, H, nullptr,
                           c->st_b.as<uint16_t>(), c->d_status);
    HIPCHK(c, hipMemcpyAsync(out, c->st_b.p, bytes, hipMemcpyDeviceToHost, c->stream));
    return sync_status(c);
}

int pgx_gray(pgx_ctx *c, const uint16_t *rgba, int W, int H, float *out)
{
    if (!c || !rgba || !out) return c ? fail(c, PGX_E_BADARG, "null pointer") : PGX_E_BADARG;
    Lock l(c);
    if (W <= 0 || H <= 0 || W > 65535 || H > 65535) return fail(c, PGX_E_BADARG, "dimensions must fit ushort");
    const size_t npix = (size_t)W * H;
    HIPCHK(c, c->st_a.ensure(npix * 8 + 32));
    HIPCHK(c, c->st_b.ensure(npix * 4 + 32));
    HIPCHK(c, hipMemcpyAsync(c->st_a.p, rgba, npix * (c->src8 ? 4 : 8), hipMemcpyHostToDevice, c->stream));
    pgx_launch_dewarp_gray(c->stream, c->st_a.p, c->src8, nullptr, 1, W, H, c->st_b.as<float>(), nullptr,
                           c->d_status);
    HIPCHK(c, hipMemcpyAsync(out, c->st_b.p, npix * 4, hipMemcpyDeviceToHost, c->stream));
    return sync_status(c);
}

int pgx_fast(pgx_ctx *c, const float *gray, int W, int H, pgx_keypoint *out, int capacity, int *n_out)
{
    if (!c || !gray || !n_out || (capacity > 0 && !out) || capacity < 0)
        return c ? fail(c, PGX_E_BADARG, "null pointer") : PGX_E_BADARG;
    Lock l(c);
    if (!c->params_set) return fail(c, PGX_E_NOT_CONFIGURED, "pgx_set_detect_params not called");
    if (W <= 0 || H <= 0 || W > 65535 || H > 65535) return fail(c, PGX_E_BADARG, "dimensions must fit ushort");
    const size_t npix = (size_t)W * H, nseg = pgx_fast_seg_count(W, H);
    const int cap = capacity > 0 ? capacity : 1;
    HIPCHK(c, c->st_a.ensure(npix * 4 + 32));
    HIPCHK(c, c->ws_seg.ensure(nseg * 32));
    HIPCHK(c, c->ws_segoff.ensure(nseg * 4));
    HIPCHK(c, c->ws_nraw.ensure(64));
    HIPCHK(c, c->st_b.ensure((size_t)cap * 4));
    HIPCHK(c, c->st_c.ensure((size_t)cap * 4));
    HIPCHK(c, hipMemcpyAsync(c->st_a.p, gray, npix * 4, hipMemcpyHostToDevice, c->stream));
    pgx_launch_fast(c->stream, c->st_a.as<float>(), 1, W, H, c->threshold, c->ws_seg.as<unsigned long long>(),
                    c->ws_segoff.as<uint32_t>(), c->ws_nraw.as<int32_t>(), c->st_b.as<uint32_t>(),
                    c->st_c.as<int32_t>(), cap, c->d_status);
    int n = 0;
    HIPCHK(c, hipMemcpyAsync(&n, c->ws_nraw.p, 4, hipMemcpyDeviceToHost, c->stream));
    int rc = sync_status(c);
    *n_out = n;
    const int nw = n < capacity ? n : capacity;
    if (nw > 0) {
        std::vector<uint32_t> xy(nw);
        std::vector<int32_t> sc(nw);
        HIPCHK(c, hipMemcpy(xy.data(), c->st_b.p, (size_t)nw * 4, hipMemcpyDeviceToHost));
        HIPCHK(c, hipMemcpy(sc.data(), c->st_c.p, (size_t)nw * 4, hipMemcpyDeviceToHost));
        for (int i = 0; i < nw; i++) {
            out[i].x = (int32_t)(xy[i] & 0xFFFFu);
            out[i].y = (int32_t)(xy[i] >> 16);
            out[i].fast_score = sc[i];
            out[i].value = gray[(size_t)out[i].y * W + out[i].x]; // Keypoint.cs:26 (a copy of the caller's pixel)
        }
    }
    if (rc == PGX_OK && n > capacity) return fail(c, PGX_E_CAPACITY, "%d hits, capacity %d", n, capacity);
    return rc;
}

int pgx_brief(pgx_ctx *c, const float *gray, int W, int H, const pgx_keypoint *kps, int n, uint32_t *desc_out)
{
    if (!c || !gray || n < 0 || (n > 0 && (!kps || !desc_out)))
        return c ? fail(c, PGX_E_BADARG, "null pointer") : PGX_E_BADARG;
    Lock l(c);
    if (!c->pairs_set) return fail(c, PGX_E_NOT_CONFIGURED, "pgx_set_brief_pairs not called");
    if (W <= 0 || H <= 0 || W > 65535 || H > 65535) return fail(c, PGX_E_BADARG, "dimensions must fit ushort");
    if (n == 0) return PGX_OK;
    const size_t npix = (size_t)W * H;
    HIPCHK(c, c->st_a.ensure(npix * 4 + 32));
    HIPCHK(c, c->st_b.ensure((size_t)n * sizeof(pgx_keypoint)));
    HIPCHK(c, c->st_c.ensure((size_t)n * c->words * 4));
    HIPCHK(c, hipMemcpyAsync(c->st_a.p, gray, npix * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->st_b.p, kps, (size_t)n * sizeof(pgx_keypoint), hipMemcpyHostToDevice, c->stream));
    pgx_launch_brief_list(c->stream, c->st_a.as<float>(), W, H, c->st_b.as<pgx_keypoint>(), n,
                          c->d_pairs.as<int32_t>(), c->P, c->st_c.as<uint32_t>());
    HIPCHK(c, hipMemcpyAsync(desc_out, c->st_c.p, (size_t)n * c->words * 4, hipMemcpyDeviceToHost, c->stream));
    return sync_status(c);
}

int pgx_nms(pgx_ctx *c, const pgx_keypoint *kps, int n, int W, int H, int32_t *order_out, int *n_out)
{
    if (!c || !n_out || n < 0 || (n > 0 && (!kps || !order_out)))
        return c ? fail(c, PGX_E_BADARG, "null pointer") : PGX_E_BADARG;
    Lock l(c);
    if (!c->params_set) return fail(c, PGX_E_NOT_CONFIGURED, "pgx_set_detect_params not called");
    if (W <= 0 || H <= 0 || W > 65535 || H > 65535) return fail(c, PGX_E_BADARG, "dimensions must fit ushort");
    *n_out = 0;
    if (n == 0) return PGX_OK;
    std::vector<uint32_t> xy(n);
    std::vector<int32_t> sc(n);
    for (int i = 0; i < n; i++) {
        if (kps[i].x < 0 || kps[i].x >= W || kps[i].y < 0 || kps[i].y >= H)
            return fail(c, PGX_E_BADARG, "keypoint %d (%d,%d) outside %dx%d", i, kps[i].x, kps[i].y, W, H);
        xy[i] = ((uint32_t)kps[i].y << 16) | (uint32_t)kps[i].x;
        sc[i] = kps[i].fast_score;
    }
    const size_t wsb = pgx_nms_ws_bytes(W, H, c->radius, n, false);
    HIPCHK(c, c->st_a.ensure((size_t)n * 4));
    HIPCHK(c, c->st_b.ensure((size_t)n * 4));
    HIPCHK(c, c->st_c.ensure((size_t)n * 4));
    HIPCHK(c, c->st_d.ensure(64));
    HIPCHK(c, c->ws_nms.ensure(wsb));
    HIPCHK(c, hipMemcpyAsync(c->st_a.p, xy.data(), (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->st_b.p, sc.data(), (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->st_d.p, &n, 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, pgx_launch_nms_sync(c->stream, c->st_a.as<uint32_t>(), c->st_b.as<int32_t>(), c->st_d.as<int32_t>(), n, W, H,
                                  c->radius, c->ws_nms.p, wsb, c->st_c.as<uint32_t>(), c->st_d.as<int32_t>() + 1, n,
                                  c->d_status));
    int nk = 0;
    HIPCHK(c, hipMemcpyAsync(&nk, c->st_d.as<int32_t>() + 1, 4, hipMemcpyDeviceToHost, c->stream));
    int rc = sync_status(c);
    if (rc != PGX_OK) return rc;
    *n_out = nk;
    if (nk > 0) HIPCHK(c, hipMemcpy(order_out, c->st_c.p, (size_t)nk * 4, hipMemcpyDeviceToHost));
    return PGX_OK;
}

int pgx_match(pgx_ctx *c, const uint32_t *desc1, int n1, const uint32_t *desc2, int n2, int words, pgx_pair *out)
{
    if (!c || n1 < 0 || n2 < 0 || (n1 > 0 && (!desc1 || !out)) || (n2 > 0 && !desc2))
        return c ? fail(c, PGX_E_BADARG, "null pointer") : PGX_E_BADARG;
    Lock l(c);
    if (n1 == 0) return PGX_OK;                                  // KeypointMatching.cs:38: loop never runs
    if (n2 == 0) return fail(c, PGX_E_EMPTY_SET, "keypoints2 is empty (ArgumentOutOfRangeException)"); // :61
    if (words <= 0 || words > 127) return fail(c, PGX_E_BADARG, "words must be in [1, 127]");
    const int S = n1 > n2 ? n1 : n2;
    if (S > (1 << PGX_IDX_BITS)) return fail(c, PGX_E_BADARG, "more than 2^20 keypoints");
    // two "frames" of S slots
    HIPCHK(c, c->st_a.ensure((size_t)2 * S * words * 4));
    HIPCHK(c, c->st_b.ensure(64));
    HIPCHK(c, c->st_c.ensure((size_t)S * sizeof(pgx_pair)));
    const int32_t meta[4] = {n1, n2, 0, 1}; // counts[2], pairlist[1][2]
    HIPCHK(c, hipMemcpyAsync(c->st_a.p, desc1, (size_t)n1 * words * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->st_a.as<uint32_t>() + (size_t)S * words, desc2, (size_t)n2 * words * 4,
                             hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->st_b.p, meta, sizeof meta, hipMemcpyHostToDevice, c->stream));
    int rc = enqueue_match(c, c->st_a.as<uint32_t>(), c->st_b.as<int32_t>(), S, words, c->st_b.as<int32_t>() + 2, 1,
                           S, c->st_c.as<pgx_pair>());
    if (rc != PGX_OK) return rc;
    HIPCHK(c, hipMemcpyAsync(out, c->st_c.p, (size_t)n1 * sizeof(pgx_pair), hipMemcpyDeviceToHost, c->stream));
    return sync_status(c);
}

int pgx_match_batch(pgx_ctx *c, const uint32_t *const *descs, const int32_t *counts, int n_frames, int words,
                    const int32_t *pair_list, int n_pairs, pgx_pair *out, int64_t *out_offsets)
{
    if (!c || n_frames < 0 || n_pairs < 0 || (n_frames > 0 && (!descs || !counts)) || (n_pairs > 0 && !pair_list))
        return c ? fail(c, PGX_E_BADARG, "null pointer") : PGX_E_BADARG;
    Lock l(c);
    if (words <= 0 || words > 127) return fail(c, PGX_E_BADARG, "words must be in [1, 127]");
    for (int f = 0; f < n_frames; f++) {
        if (counts[f] < 0 || counts[f] > (1 << PGX_IDX_BITS)) return fail(c, PGX_E_BADARG, "counts[%d] = %d", f, counts[f]);
        if (counts[f] > 0 && !descs[f]) return fail(c, PGX_E_BADARG, "descs[%d] is null", f);
    }
    // the slot size S (descriptor slots per frame, entries per list) is the largest set that some image pair REFERENCES:
    // a large frame nobody matches costs neither upload nor workspace nor download
    int S = 1;
    std::vector<char> used((size_t)(n_frames > 0 ? n_frames : 1), 0);
    int64_t total = 0;
    bool empty_set = false;
    for (int m = 0; m < n_pairs; m++) {
        const int fa = pair_list[2 * m], fb = pair_list[2 * m + 1];
        if (fa < 0 || fa >= n_frames || fb < 0 || fb >= n_frames) return fail(c, PGX_E_BADARG, "pair %d names frame %d / %d of %d", m, fa, fb, n_frames);
        if (out_offsets) out_offsets[m] = total;
        total += counts[fa];
        if (counts[fa] > 0 && counts[fb] == 0) empty_set = true;
        used[(size_t)fa] = used[(size_t)fb] = 1;
        if (counts[fa] > S) S = counts[fa];
        if (counts[fb] > S) S = counts[fb];
    }
    if (out_offsets) out_offsets[n_pairs] = total;
    if (n_pairs == 0 || total == 0) return PGX_OK;
    if (!out) return fail(c, PGX_E_BADARG, "null pointer");
    // stage: [F][S][words] descriptors + counts + pair list in one pinned block, one upload
    const size_t desc_bytes = (size_t)n_frames * S * words * 4, cnt_bytes = ((size_t)n_frames * 4 + 15) & ~(size_t)15;
    const size_t pl_bytes = (size_t)n_pairs * 8, in_bytes = desc_bytes + cnt_bytes + pl_bytes;
    const size_t out_bytes = (size_t)n_pairs * S * sizeof(pgx_pair);
    HIPCHK(c, c->pin_in.ensure(in_bytes));
    HIPCHK(c, c->pin_out.ensure(out_bytes));
    HIPCHK(c, c->st_a.ensure(in_bytes));
    HIPCHK(c, c->st_c.ensure(out_bytes));
    uint8_t *hin = c->pin_in.as<uint8_t>();
    for (int f = 0; f < n_frames; f++)
        if (counts[f] > 0 && used[(size_t)f]) memcpy(hin + (size_t)f * S * words * 4, descs[f], (size_t)counts[f] * words * 4);
    memcpy(hin + desc_bytes, counts, (size_t)n_frames * 4); // an unreferenced frame's count is never read on the device
    memcpy(hin + desc_bytes + cnt_bytes, pair_list, pl_bytes);
    HIPCHK(c, hipMemcpyAsync(c->st_a.p, hin, in_bytes, hipMemcpyHostToDevice, c->stream));
    uint8_t *din = c->st_a.as<uint8_t>();
    int rc = enqueue_match(c, reinterpret_cast<const uint32_t *>(din), reinterpret_cast<const int32_t *>(din + desc_bytes), S, words,
                           reinterpret_cast<const int32_t *>(din + desc_bytes + cnt_bytes), n_pairs, S, c->st_c.as<pgx_pair>());
    if (rc != PGX_OK) return rc;
    HIPCHK(c, hipMemcpyAsync(c->pin_out.p, c->st_c.p, out_bytes, hipMemcpyDeviceToHost, c->stream));
    rc = sync_status(c);
    if (rc != PGX_OK && rc != PGX_E_EMPTY_SET) return rc;
    const pgx_pair *hout = c->pin_out.as<pgx_pair>();
    int64_t o = 0;
    for (int m = 0; m < n_pairs; m++) {
        const int n = counts[pair_list[2 * m]];
        if (n > 0) memcpy(out + o, hout + (size_t)m * S, (size_t)n * sizeof(pgx_pair));
        o += n;
    }
    if (empty_set || rc == PGX_E_EMPTY_SET) return fail(c, PGX_E_EMPTY_SET, "keypoints2 is empty while keypoints1 is not (ArgumentOutOfRangeException)");
    return PGX_OK;
}

int pgx_detect(pgx_ctx *c, const uint16_t *rgba, int W, int H, pgx_keypoint *kp_out, uint32_t *desc_out,
               int capacity, int *n_out, int *n_raw)
{
    if (!c || !rgba || !kp_out || !desc_out || !n_out || capacity <= 0)
        return c ? fail(c, PGX_E_BADARG, "null pointer") : PGX_E_BADARG;
    Lock l(c);
    if (W <= 0 || H <= 0 || W > 65535 || H > 65535) return fail(c, PGX_E_BADARG, "dimensions must fit ushort");
    const size_t npix = (size_t)W * H;
    HIPCHK(c, c->st_e.ensure(npix * 8 + 32));
    HIPCHK(c, c->st_f.ensure((size_t)capacity * (sizeof(pgx_keypoint) + (size_t)(c->words ? c->words : 1) * 4) + 64));
    pgx_keypoint *d_kp = c->st_f.as<pgx_keypoint>();
    uint32_t *d_desc = reinterpret_cast<uint32_t *>(d_kp + capacity);
    int32_t *d_cnt = reinterpret_cast<int32_t *>(d_desc + (size_t)capacity * (c->words ? c->words : 1));
    HIPCHK(c, hipMemcpyAsync(c->st_e.p, rgba, npix * (c->src8 ? 4 : 8), hipMemcpyHostToDevice, c->stream));
    int rc = enqueue_detect(c, c->st_e.as<uint16_t>(), 1, W, H, d_kp, d_desc, d_cnt, d_cnt + 1, capacity);
    if (rc != PGX_OK) return rc;
    int cnt[2] = {0, 0};
    HIPCHK(c, hipMemcpyAsync(cnt, d_cnt, 8, hipMemcpyDeviceToHost, c->stream));
    rc = sync_status(c);
    *n_out = cnt[0];
    if (n_raw) *n_raw = cnt[1];
    if (cnt[0] > 0) {
        HIPCHK(c, hipMemcpy(kp_out, d_kp, (size_t)cnt[0] * sizeof(pgx_keypoint), hipMemcpyDeviceToHost));
        HIPCHK(c, hipMemcpy(desc_out, d_desc, (size_t)cnt[0] * c->words * 4, hipMemcpyDeviceToHost));
    }
    return rc;
}

// ---- device-resident batched entry points -------------------------------------------------

int pgx_detect_batch_dev(pgx_ctx *c, const uint16_t *d_rgba, int F, int W, int H, pgx_keypoint *d_kp,
                         uint32_t *d_desc, int32_t *d_counts, int32_t *d_nraw, int capacity)
{
    if (!c || !d_rgba || !d_kp || !d_desc || !d_counts || !d_nraw || capacity <= 0 || F < 0)
        return c ? fail(c, PGX_E_BADARG, "null pointer") : PGX_E_BADARG;
    Lock l(c);
    if (W <= 0 || H <= 0 || W > 65535 || H > 65535) return fail(c, PGX_E_BADARG, "dimensions must fit ushort");
    return enqueue_detect(c, d_rgba, F, W, H, d_kp, d_desc, d_counts, d_nraw, capacity);
}

int pgx_match_batch_dev(pgx_ctx *c, const uint32_t *d_desc, const int32_t *d_counts, int stride, int words,
                        const int32_t *d_pairlist, int M, int max_count, pgx_pair *d_out)
{
    if (!c || !d_desc || !d_counts || !d_pairlist || !d_out || M < 0)
        return c ? fail(c, PGX_E_BADARG, "null pointer") : PGX_E_BADARG;
    Lock l(c);
    return enqueue_match(c, d_desc, d_counts, stride, words, d_pairlist, M, max_count, d_out);
}

// ---- RANSAC fundamental matrix and pose ----------------------------------------------------------

int pgx_fundamental_ransac_dev(pgx_ctx *c, const pgx_keypoint *d_kp, const pgx_pair *d_matches, const int32_t *d_counts,
                               const int32_t *d_pairlist, int M, int stride, int n_samples, int pairs_per_sample,
                               float threshold, int rank_check, uint64_t seed, float *d_F, int32_t *d_inliers,
                               int32_t *d_best_sample)
{
    if (!c || !d_kp || !d_matches || !d_counts || !d_pairlist || !d_F || !d_inliers || !d_best_sample || M < 0 || stride <= 0)
        return c ? fail(c, PGX_E_BADARG, "null pointer") : PGX_E_BADARG;
    Lock l(c);
    if (pairs_per_sample < 8) // CameraPoseEstimation.cs:28-29
        return fail(c, PGX_E_BADARG, "At least 8 keypoint pairs must be included per sample (InvalidOperationException)");
    if (pairs_per_sample > 64 || n_samples <= 0 || n_samples > 65535 * 64)
        return fail(c, PGX_E_BADARG, "pairs_per_sample must be <= 64 and n_samples in [1, 4194240]");
    if (M == 0) return PGX_OK;
    HIPCHK(c, c->ws_pose.ensure(pgx_pose_ws_bytes(M, n_samples)));
    pgx_launch_fundamental(c->stream, d_kp, d_matches, d_counts, d_pairlist, M, stride, n_samples, pairs_per_sample, threshold,
                           rank_check, seed, c->ws_pose.p, d_F, d_inliers, d_best_sample);
    HIPCHK(c, hipGetLastError());
    return PGX_OK;
}

int pgx_pose_dev(pgx_ctx *c, const pgx_keypoint *d_kp, const pgx_pair *d_matches, const int32_t *d_counts,
                 const int32_t *d_pairlist, int M, int stride, const float *d_F, float *d_Rt, int32_t *d_votes,
                 int32_t *d_best, float *d_points)
{
    if (!c || !d_kp || !d_matches || !d_counts || !d_pairlist || !d_F || !d_Rt || !d_votes || !d_best || M < 0 || stride <= 0)
        return c ? fail(c, PGX_E_BADARG, "null pointer") : PGX_E_BADARG;
    Lock l(c);
    pgx_launch_pose(c->stream, d_kp, d_matches, d_counts, d_pairlist, M, stride, d_F, d_Rt, d_votes, d_best, d_points);
    HIPCHK(c, hipGetLastError());
    return PGX_OK;
}

// ---- the track graph on the device ---------------------------------------------------------------

int pgx_tracks_dev(pgx_ctx *c, const pgx_pair *d_matches, const int32_t *d_counts, const int32_t *d_pairlist, int M, int F,
                   int stride, const int32_t *d_frame_ids, int n_frames, int max_dist, int min_len, int32_t *d_track_of,
                   int32_t *d_offsets, int32_t *d_nodes, int32_t *d_summary)
{
    if (!c || !d_counts || !d_track_of || !d_offsets || !d_nodes || !d_summary || M < 0 || (M > 0 && (!d_matches || !d_pairlist)))
        return c ? fail(c, PGX_E_BADARG, "null pointer") : PGX_E_BADARG;
    Lock l(c);
    if (F <= 0 || stride <= 0 || n_frames <= 0) return fail(c, PGX_E_BADARG, "F, stride and n_frames must be positive");
    if (!d_frame_ids && n_frames != F) return fail(c, PGX_E_BADARG, "without d_frame_ids, n_frames must equal F");
    if ((long long)n_frames * stride > (1ll << 30)) return fail(c, PGX_E_BADARG, "n_frames * stride must be <= 2^30");
    if ((long long)M * ((stride + 255) / 256) > 0x7FFFFFFFll || (long long)F * ((stride + 255) / 256) > 0x7FFFFFFFll)
        return fail(c, PGX_E_BADARG, "too many image pairs for one call");
    HIPCHK(c, c->ws_tracks.ensure(pgx_tracks_ws_bytes(n_frames, stride)));
    {
        ProfScope ps(c, "tracks");
        pgx_launch_tracks(c->stream, d_matches, d_counts, d_pairlist, M, F, stride, d_frame_ids, n_frames, max_dist, min_len,
                          c->ws_tracks.p, d_track_of, d_offsets, d_nodes, d_summary);
    }
    HIPCHK(c, hipGetLastError());
    return PGX_OK;
}

// ---- measurement hooks ---------------------------------------------------------------------

int pgx_profile_enable(pgx_ctx *c, int on)
{
    if (!c) return PGX_E_BADARG;
    Lock l(c);
    c->prof_on = on != 0;
    return PGX_OK;
}

int pgx_profile_filter(pgx_ctx *c, const char *name)
{
    if (!c) return PGX_E_BADARG;
    Lock l(c);
    c->prof_only = name ? name : "";
    return PGX_OK;
}

static void prof_drain(pgx_ctx *c)
{
    for (auto &kv : c->prof) {
        for (auto &ev : kv.second.pending) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, ev.first, ev.second) == hipSuccess) {
                kv.second.total_ms += ms;
                kv.second.launches += 1;
            }
            c->ev_pool.push_back(ev.first);
            c->ev_pool.push_back(ev.second);
        }
        kv.second.pending.clear();
    }
}

int pgx_profile_get(pgx_ctx *c, const char *name, int *launches, double *total_ms)
{
    if (!c || !name) return PGX_E_BADARG;
    Lock l(c);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    prof_drain(c);
    auto it = c->prof.find(name);
    if (launches) *launches = it == c->prof.end() ? 0 : it->second.launches;
    if (total_ms) *total_ms = it == c->prof.end() ? 0.0 : it->second.total_ms;
    return PGX_OK;
}

int pgx_profile_serialize(pgx_ctx *c, int on)
{
    if (!c) return PGX_E_BADARG;
    Lock l(c);
    c->prof_serial = on != 0;
    return PGX_OK;
}

int pgx_profile_reset(pgx_ctx *c)
{
    if (!c) return PGX_E_BADARG;
    Lock l(c);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    prof_drain(c);
    c->prof.clear();
    return PGX_OK;
}

int pgx_debug_counters(pgx_ctx *c, int64_t *out8)
{
    if (!c || !out8) return PGX_E_BADARG;
    Lock l(c);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(out8, c->d_status + PGX_DBG_OFF, 64, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemset(c->d_status + PGX_DBG_OFF, 0, 64));
    return PGX_OK;
}

int pgx_match_stats(pgx_ctx *c, int *rounds_wide, int64_t *evaluations, int64_t *evaluations_round0)
{
    if (!c) return PGX_E_BADARG;
    Lock l(c);
    unsigned long long ev[PGX_MAX_WIDE_ROUNDS] = {0};
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(ev, c->d_status + 4, sizeof ev, hipMemcpyDeviceToHost));
    long long tot = 0;
    for (int r = 0; r < c->last_rounds_mfma && r < PGX_MAX_WIDE_ROUNDS; r++) tot += (long long)ev[r];
    if (rounds_wide) *rounds_wide = c->last_rounds_mfma;
    if (evaluations) *evaluations = tot;
    if (evaluations_round0) *evaluations_round0 = c->last_rounds_mfma > 0 ? (long long)ev[0] : 0;
    return PGX_OK;
}

} // extern "C"
