// pgx_comm.hip -- multi-GPU part of the C ABI ("ctx owns the RCCL comm", PGX_E_RCCL).
//
// One process per GPU, one context per process.  The path needs exactly two exchanges per job, both all-gathers of
// fixed-size records (per-frame {count, descriptors}; per-image-pair match lists: the reference always emits exactly N1
// entries, KeypointMatching.cs:38), so the ABI offers an in-place all-gather on the context's stream and the four-phase
// step built from it.  RCCL is loaded at run time (dlopen) so that libpgx.so itself has no link-time dependency: a
// single-GPU host never touches it, and a missing library surfaces as PGX_E_RCCL from pgx_comm_unique_id / pgx_comm_init,
// nowhere else.  PGX_RCCL_LIB names the library file to load instead of the default search (deployment knob: a host that
// ships its own RCCL build; the CPU test of the missing-library path points it at a file that does not exist).
//
// Failure on ONE rank: a collective is a rendezvous, so a rank that returned early would leave its peers waiting in the
// all-gather for ever.  pgx_sequence_step_dev therefore does every rank-local check and every workspace allocation BEFORE
// the first collective and -- on the first call, and again whenever the rank-symmetric arguments or the configuration
// epoch change -- ends that part with a one-int status exchange that the failing rank joins too: every rank returns an
// error, nobody waits.  The decision to exchange never depends on rank-local state (see the function).  A HIP or RCCL error in the middle
// of a step (after that point) makes the rank abort its communicator (ncclCommAbort, a required symbol); that is best
// effort: on one node the peers' collective kernels may spin on shared flags without noticing, so the host must time
// its ranks out.  N > 1 on RCCL has only ever run under the driver's multi-GPU bench (the development box has one GPU);
// the agreement logic is rehearsed on CPU through its torch twin (dist.ShardedSequence, tests/test_dist_gloo.py).
#include "pgx_internal.h"

#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace {

// the slice of rccl.h this file needs (ABI-stable NCCL entry points)
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;
enum { ncclSuccessV = 0, ncclUint8V = 1 };

struct Rccl {
    void *h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string err;
};

Rccl *rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        const char *only = getenv("PGX_RCCL_LIB");
        std::string why;
        for (const char *n : names) {
            if (only && only[0]) n = only;
            r.h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (r.h) break;
            const char *e = dlerror();   // ONE call: dlerror() clears the message it returns
            if (why.empty()) why = e ? e : "?";
            if (only && only[0]) break;
        }
        if (!r.h) { r.err = "dlopen(librccl) failed: " + why; return; }
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(r.h, "ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(r.h, "ncclCommInitRank"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.h, "ncclCommDestroy"));
        r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(r.h, "ncclAllGather"));
        r.CommAbort = reinterpret_cast<decltype(r.CommAbort)>(dlsym(r.h, "ncclCommAbort"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.h, "ncclGetErrorString"));
        if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllGather || !r.CommAbort) r.err = "librccl lacks an expected nccl* symbol";
    });
    return &r;
}

int comm_fail(pgx_ctx *c, const char *what, ncclResult_t rc)
{
    Rccl *r = rccl();
    char buf[256];
    snprintf(buf, sizeof buf, "%s: %s", what, (r->GetErrorString && rc) ? r->GetErrorString(rc) : (r->err.empty() ? "RCCL error" : r->err.c_str()));
    pgx_note_error(c, buf);
    return PGX_E_RCCL;
}

} // namespace

int pgx_enqueue_detect(pgx_ctx *c, const uint16_t *d_rgba, int F, int W, int H, pgx_keypoint *d_kp, uint32_t *d_desc,
                       int32_t *d_counts, int32_t *d_nraw, int cap);
int pgx_enqueue_match(pgx_ctx *c, const uint32_t *d_desc, const int32_t *d_counts, int stride, int words,
                      const int32_t *d_pairlist, int M, int max_n, pgx_pair *d_out);
int pgx_prepare_detect(pgx_ctx *c, int F, int W, int H, int cap);
int pgx_prepare_match(pgx_ctx *c, int stride, int words, int M);

extern "C" {

int pgx_comm_unique_id(void *id_out)
{
    if (!id_out) return PGX_E_BADARG;
    Rccl *r = rccl();
    if (!r->err.empty()) return PGX_E_RCCL;
    ncclUniqueId id;
    if (r->GetUniqueId(&id) != ncclSuccessV) return PGX_E_RCCL;
    static_assert(sizeof id == PGX_COMM_ID_BYTES, "pgx.h promises the size of ncclUniqueId");
    memcpy(id_out, &id, sizeof id);
    return PGX_OK;
}

int pgx_comm_init(pgx_ctx *c, int rank, int world, const void *id)
{
    if (!c || !id || world < 1 || rank < 0 || rank >= world) return c ? (pgx_note_error(c, "bad rank/world"), PGX_E_BADARG) : PGX_E_BADARG;
    std::lock_guard<std::mutex> g(c->mu);
    (void)hipSetDevice(c->device);
    if (c->comm) { pgx_note_error(c, "communicator already initialised"); return PGX_E_BADARG; }
    Rccl *r = rccl();
    if (!r->err.empty()) return comm_fail(c, "pgx_comm_init", 0);
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof uid);
    ncclComm_t comm = nullptr;
    const ncclResult_t rc = r->CommInitRank(&comm, world, uid, rank);
    if (rc != ncclSuccessV) return comm_fail(c, "ncclCommInitRank", rc);
    c->comm = comm;
    c->comm_rank = rank;
    c->comm_world = world;
    c->comm_agreed.clear();
    return PGX_OK;
}

int pgx_comm_destroy(pgx_ctx *c)
{
    if (!c) return PGX_E_BADARG;
    std::lock_guard<std::mutex> g(c->mu);
    if (!c->comm) return PGX_OK;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    const ncclResult_t rc = rccl()->CommDestroy(reinterpret_cast<ncclComm_t>(c->comm));
    c->comm = nullptr;
    c->comm_rank = 0;
    c->comm_world = 1;
    c->comm_agreed.clear();
    return rc == ncclSuccessV ? PGX_OK : comm_fail(c, "ncclCommDestroy", rc);
}

int pgx_comm_info(pgx_ctx *c, int *rank, int *world)
{
    if (!c) return PGX_E_BADARG;
    if (rank) *rank = c->comm_rank;
    if (world) *world = c->comm_world;
    return PGX_OK;
}

static int allgather_locked(pgx_ctx *c, void *d_buf, size_t bytes_per_rank)
{
    if (c->comm_world <= 1 || bytes_per_rank == 0) return PGX_OK; // one rank: the buffer already is the gathered one
    if (!c->comm) { pgx_note_error(c, "pgx_comm_init not called"); return PGX_E_NOT_CONFIGURED; }
    const char *send = reinterpret_cast<const char *>(d_buf) + (size_t)c->comm_rank * bytes_per_rank; // in place
    const ncclResult_t rc = rccl()->AllGather(send, d_buf, bytes_per_rank, ncclUint8V, reinterpret_cast<ncclComm_t>(c->comm), c->stream);
    return rc == ncclSuccessV ? PGX_OK : comm_fail(c, "ncclAllGather", rc);
}

int pgx_allgather_dev(pgx_ctx *c, void *d_buf, size_t bytes_per_rank)
{
    if (!c || !d_buf) return PGX_E_BADARG;
    std::lock_guard<std::mutex> g(c->mu);
    (void)hipSetDevice(c->device);
    return allgather_locked(c, d_buf, bytes_per_rank);
}

int pgx_sequence_step_dev(pgx_ctx *c, const uint16_t *d_frames_local, int n_local_frames, int frame_slots, int W, int H,
                          pgx_keypoint *d_kp_local, uint32_t *d_desc_all, int32_t *d_counts_all, int32_t *d_nraw_local,
                          int capacity, const int32_t *d_pairlist_local, int n_local_pairs, int pair_slots,
                          pgx_pair *d_out_all)
{
    if (!c) return PGX_E_BADARG;
    std::lock_guard<std::mutex> g(c->mu);
    (void)hipSetDevice(c->device);
    if (c->comm_world > 1 && !c->comm) { pgx_note_error(c, "pgx_comm_init not called"); return PGX_E_NOT_CONFIGURED; }
    const int words = c->words, r = c->comm_rank;
    // Every rank-local check and allocation first: nothing below this block fails for a local reason in normal operation.
    // A rank that does fail here (a bad argument, not configured, a size mismatch, no memory) must not leave its peers waiting
    // in the first all-gather, so this block ends with a status exchange -- one int per rank through the same all-gather,
    // which the failing rank joins too -- and every rank returns an error if any rank failed.
    // WHETHER the exchange runs must be the same decision on every rank, or the ranks issue different collectives.  It is
    // therefore taken from nothing but (a) the arguments that the contract requires to be equal on all ranks -- slot counts,
    // image size, capacity -- and (b) the context's configuration epoch, which every pgx_set_* call bumps (configuration
    // calls are init-time and must be made on every rank alike): the exchange runs on the first call and whenever that key
    // changes, NEVER because of this rank's own rc or its own share of the work (n_local_*: the last batch may be short on
    // one rank only).  A rank that fails locally under an agreed key (e.g. a larger local share that no longer fits) cannot
    // tell its peers through a collective they are not going to issue: it aborts its communicator like any mid-step failure.
    int rc = PGX_OK;
    if (!d_desc_all || !d_counts_all || !d_out_all || capacity <= 0 || frame_slots < 0 || pair_slots < 0 ||
        n_local_frames < 0 || n_local_frames > frame_slots || n_local_pairs < 0 || n_local_pairs > pair_slots ||
        (n_local_frames > 0 && (!d_frames_local || !d_kp_local || !d_nraw_local)) || (n_local_pairs > 0 && !d_pairlist_local)) {
        pgx_note_error(c, "bad argument");
        rc = PGX_E_BADARG;
    }
    if (rc == PGX_OK && !c->pairs_set) { pgx_note_error(c, "pgx_set_brief_pairs not called"); rc = PGX_E_NOT_CONFIGURED; }
    if (rc == PGX_OK) rc = pgx_prepare_detect(c, n_local_frames, W, H, capacity);
    if (rc == PGX_OK) rc = pgx_prepare_match(c, capacity, words, n_local_pairs);
    if (c->comm_world > 1) {
        char key[160];
        snprintf(key, sizeof key, "%d/%dx%d/%d/%d/e%u", frame_slots, W, H, capacity, pair_slots, c->cfg_epoch);
        if (c->comm_agreed != key) {
            const std::string own = rc != PGX_OK ? std::string(pgx_last_error(c)) : std::string();
            const int G = c->comm_world;
            std::vector<int> st((size_t)G, 0);
            st[(size_t)r] = rc;
            int xrc = PGX_OK;
            if (c->ws_agree.ensure((size_t)G * sizeof(int)) != hipSuccess ||
                hipMemcpyAsync(c->ws_agree.p, st.data(), (size_t)G * sizeof(int), hipMemcpyHostToDevice, c->stream) != hipSuccess)
                xrc = PGX_E_HIP;
            if (xrc == PGX_OK) xrc = allgather_locked(c, c->ws_agree.p, sizeof(int));
            if (xrc == PGX_OK && (hipMemcpyAsync(st.data(), c->ws_agree.p, (size_t)G * sizeof(int), hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
                                  hipStreamSynchronize(c->stream) != hipSuccess))
                xrc = PGX_E_HIP;
            if (rc != PGX_OK) { pgx_note_error(c, own); return rc; }   // this rank's own error, text restored
            if (xrc != PGX_OK) { if (xrc == PGX_E_HIP) pgx_note_error(c, "status exchange before the first collective failed"); return xrc; }
            for (int k = 0; k < G; k++)
                if (st[(size_t)k] != PGX_OK) {
                    char msg[160];
                    snprintf(msg, sizeof msg, "rank %d failed before the first collective (code %d); nothing was exchanged", k, st[(size_t)k]);
                    pgx_note_error(c, msg);
                    return PGX_E_RCCL;
                }
            c->comm_agreed = key;
        } else if (rc != PGX_OK) {
            // a local failure under an agreed key: the peers are about to issue the step's all-gathers, not a status exchange
            const std::string own(pgx_last_error(c));
            (void)rccl()->CommAbort(reinterpret_cast<ncclComm_t>(c->comm));
            c->comm = nullptr; c->comm_rank = 0; c->comm_world = 1;
            c->comm_agreed.clear();
            pgx_note_error(c, own + " (after the ranks had agreed on these arguments: communicator aborted)");
            return rc;
        }
    } else if (rc != PGX_OK) return rc;
    // From here on a local failure is a HIP or RCCL error in the middle of the step.  The communicator is aborted so that
    // the peers' collectives have a chance to fail instead of waiting -- best effort: on one node the peers' collective
    // kernels may not notice a remote abort, and the host is expected to time its ranks out (include/pgx.h).
    auto bail = [&](int code) {
        if (c->comm_world > 1 && c->comm) {
            (void)rccl()->CommAbort(reinterpret_cast<ncclComm_t>(c->comm));
            c->comm = nullptr; c->comm_rank = 0; c->comm_world = 1;
            c->comm_agreed.clear();
            pgx_note_error(c, std::string(pgx_last_error(c)) + " (communicator aborted)");
        }
        return code;
    };
    // phase 1: detect the frames this rank owns, straight into its block of the gathered buffers
    uint32_t *desc_l = d_desc_all + (size_t)r * frame_slots * capacity * words;
    int32_t *counts_l = d_counts_all + (size_t)r * frame_slots;
    rc = pgx_enqueue_detect(c, d_frames_local, n_local_frames, W, H, d_kp_local, desc_l, counts_l, d_nraw_local, capacity);
    if (rc != PGX_OK) return bail(rc);
    // phase 2: the fixed-size per-frame records
    if ((rc = allgather_locked(c, d_desc_all, (size_t)frame_slots * capacity * words * 4)) != PGX_OK) return bail(rc);
    if ((rc = allgather_locked(c, d_counts_all, (size_t)frame_slots * 4)) != PGX_OK) return bail(rc);
    // phase 3: the image pairs this rank owns (pair list pre-mapped to slots of the gathered buffer)
    pgx_pair *out_l = d_out_all + (size_t)r * pair_slots * capacity;
    rc = pgx_enqueue_match(c, d_desc_all, d_counts_all, capacity, words, d_pairlist_local, n_local_pairs, capacity, out_l);
    if (rc != PGX_OK) return bail(rc);
    // phase 4: the fixed-size match lists
    rc = allgather_locked(c, d_out_all, (size_t)pair_slots * capacity * sizeof(pgx_pair));
    return rc == PGX_OK ? rc : bail(rc);
}

} // extern "C"
