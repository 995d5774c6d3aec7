// k_nms.hip -- RedundantKeypointEliminator.EliminateRedundantKeypoints on gfx950.
//
// Reference: ImageProcessing/RedundantKeypointEliminator.cs:16-39 -- stable sort by FastScore
// descending, then repeatedly keep the head and drop everything with
// sqrt(dx^2+dy^2) > r FALSE, i.e. dx^2+dy^2 <= r^2 (integers, so the test is exact).
//
// Parallel formulation (proven equal in tests against the literal oracle):
// priority = (score desc, input index asc).  Round: every undecided point that has no
// higher-priority undecided point within r becomes accepted; every undecided point within r
// of a newly accepted one becomes suppressed.  Output = accepted points in priority order.
//
// One 1024-thread workgroup per frame (frames of a batch run on different CUs); points are
// binned into a uniform grid of cells >= r so a round only visits 3x3 cells; all per-point
// arrays are re-ordered into cell order so neighbour scans are contiguous.  Integer work on
// a few hundred KB of L2-resident data: not HBM-relevant; latency-bound on the round count.
#include "pgx_internal.h"

namespace {

constexpr int NT = 1024;

struct NmsLayout {
    int gw, gh, cs, ncell;
    size_t off_cellstart, off_cellfill, off_sxy, off_sscore, off_sidx, off_state, off_listA, off_listB, off_accflag,
        total;
};

__host__ __device__ inline NmsLayout nms_layout(int W, int H, int radius, int n_cap)
{
    NmsLayout L;
    int cs = radius > 16 ? radius : 16;
    L.cs = cs;
    L.gw = (W + cs - 1) / cs; if (L.gw < 1) L.gw = 1;
    L.gh = (H + cs - 1) / cs; if (L.gh < 1) L.gh = 1;
    L.ncell = L.gw * L.gh;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o += (bytes + 255) & ~(size_t)255; return r; };
    L.off_cellstart = take((size_t)(L.ncell + 1) * 4);
    L.off_cellfill = take((size_t)(L.ncell + 1) * 4);
    L.off_sxy = take((size_t)n_cap * 4);
    L.off_sscore = take((size_t)n_cap * 4);
    L.off_sidx = take((size_t)n_cap * 4);
    L.off_state = take((size_t)n_cap);
    L.off_listA = take((size_t)n_cap * 4);
    L.off_listB = take((size_t)n_cap * 4);
    L.off_accflag = take((size_t)n_cap);
    L.total = o;
    return L;
}

// exclusive scan of one value per thread over the 1024-thread block; returns the block total
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t *excl, uint32_t *wsum /*[17]*/)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = __shfl_up(incl, d);
        if (lane >= d) incl += t;
    }
    __syncthreads(); // wsum free to overwrite
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    uint32_t woff = 0, total = 0;
#pragma unroll
    for (int w = 0; w < NT / 64; w++) {
        uint32_t s = wsum[w];
        if (w < wv) woff += s;
        total += s;
    }
    *excl = woff + incl - v;
    return total;
}

enum : uint8_t { ST_UNDECIDED = 0, ST_ACCEPTED = 1, ST_SUPPRESSED = 2, ST_CAND = 3 };

__device__ __forceinline__ bool better(int sq, uint32_t iq, int si, uint32_t ii)
{
    return sq > si || (sq == si && iq < ii);
}

__global__ __launch_bounds__(NT) void k_nms(const uint32_t *__restrict__ raw_xy_all,
                                            const int32_t *__restrict__ raw_score_all,
                                            const int32_t *__restrict__ n_raw_all, int n_cap, int W, int H, int radius,
                                            unsigned char *ws_all, size_t ws_stride, uint32_t *__restrict__ order_all,
                                            int32_t *__restrict__ n_kept_all, int kp_cap, int *status)
{
    __shared__ uint32_t wsum[NT / 64 + 1];
    __shared__ uint32_t sh_cnt;
    __shared__ int sh_max;
    const int f = blockIdx.x, tid = threadIdx.x;
    const uint32_t *raw_xy = raw_xy_all + (size_t)f * n_cap;
    const int32_t *raw_score = raw_score_all + (size_t)f * n_cap;
    int n = n_raw_all[f];
    if (n > n_cap) n = n_cap;
    if (n < 0) n = 0;
    uint32_t *order = order_all + (size_t)f * kp_cap;
    unsigned char *ws = ws_all + (size_t)f * ws_stride;
    const NmsLayout L = nms_layout(W, H, radius, n_cap);
    uint32_t *cell_start = reinterpret_cast<uint32_t *>(ws + L.off_cellstart);
    uint32_t *cell_fill = reinterpret_cast<uint32_t *>(ws + L.off_cellfill);
    uint32_t *s_xy = reinterpret_cast<uint32_t *>(ws + L.off_sxy);
    int32_t *s_score = reinterpret_cast<int32_t *>(ws + L.off_sscore);
    uint32_t *s_idx = reinterpret_cast<uint32_t *>(ws + L.off_sidx);
    uint8_t *state = reinterpret_cast<uint8_t *>(ws + L.off_state);
    uint32_t *listA = reinterpret_cast<uint32_t *>(ws + L.off_listA);
    uint32_t *listB = reinterpret_cast<uint32_t *>(ws + L.off_listB);
    uint8_t *accflag = reinterpret_cast<uint8_t *>(ws + L.off_accflag);

    if (n == 0) {
        if (tid == 0) n_kept_all[f] = 0;
        return;
    }

    const bool suppress_any = radius >= 0; // distance > r is always true for r < 0
    const long long r2 = (long long)radius * (long long)radius;

    if (suppress_any) {
        // ---- bin into cells (counting sort) ----
        for (int c = tid; c <= L.ncell; c += NT) cell_fill[c] = 0;
        __syncthreads();
        for (int i = tid; i < n; i += NT) {
            const uint32_t xy = raw_xy[i];
            const int cx = (int)(xy & 0xFFFFu) / L.cs, cy = (int)(xy >> 16) / L.cs;
            atomicAdd(&cell_fill[cy * L.gw + cx], 1u);
        }
        __syncthreads();
        // exclusive scan over cells, chunks of NT
        {
            uint32_t carry = 0;
            for (int base = 0; base <= L.ncell; base += NT) {
                const int c = base + tid;
                const uint32_t v = (c < L.ncell) ? cell_fill[c] : 0u;
                uint32_t ex;
                const uint32_t tot = block_excl_scan(v, &ex, wsum);
                if (c <= L.ncell) cell_start[c] = carry + ex;
                carry += tot;
            }
        }
        __syncthreads();
        for (int c = tid; c < L.ncell; c += NT) cell_fill[c] = cell_start[c];
        __syncthreads();
        for (int i = tid; i < n; i += NT) {
            const uint32_t xy = raw_xy[i];
            const int cx = (int)(xy & 0xFFFFu) / L.cs, cy = (int)(xy >> 16) / L.cs;
            const uint32_t pos = atomicAdd(&cell_fill[cy * L.gw + cx], 1u);
            s_xy[pos] = xy;
            s_score[pos] = raw_score[i];
            s_idx[pos] = (uint32_t)i;
            state[pos] = ST_UNDECIDED;
            listA[pos] = pos;
        }
        __syncthreads();

        // ---- rounds ----
        uint32_t *cur = listA, *nxt = listB;
        int n_act = n;
        while (n_act > 0) {
            // phase A: local-best undecided points become candidates
            for (int a = tid; a < n_act; a += NT) {
                const uint32_t p = cur[a];
                const uint32_t xy = s_xy[p];
                const int x = (int)(xy & 0xFFFFu), y = (int)(xy >> 16);
                const int sc = s_score[p];
                const uint32_t id = s_idx[p];
                const int cx = x / L.cs, cy = y / L.cs;
                const int cx0 = cx > 0 ? cx - 1 : 0, cx1 = cx + 1 < L.gw ? cx + 1 : L.gw - 1;
                const int cy0 = cy > 0 ? cy - 1 : 0, cy1 = cy + 1 < L.gh ? cy + 1 : L.gh - 1;
                bool beaten = false;
                for (int yy = cy0; yy <= cy1 && !beaten; yy++) {
                    const uint32_t q0 = cell_start[yy * L.gw + cx0], q1 = cell_start[yy * L.gw + cx1 + 1];
                    for (uint32_t q = q0; q < q1; q++) {
                        const uint8_t stq = state[q];
                        if (stq != ST_UNDECIDED && stq != ST_CAND) continue; // CAND written this phase = still undecided
                        if (q == p) continue;
                        const uint32_t qxy = s_xy[q];
                        const long long dx = (int)(qxy & 0xFFFFu) - x, dy = (int)(qxy >> 16) - y;
                        if (dx * dx + dy * dy > r2) continue;
                        if (better(s_score[q], s_idx[q], sc, id)) { beaten = true; break; }
                    }
                }
                if (!beaten) state[p] = ST_CAND;
            }
            __syncthreads();
            // phase B: candidates are accepted; undecided points near a candidate are suppressed
            if (tid == 0) sh_cnt = 0;
            __syncthreads();
            for (int a = tid; a < n_act; a += NT) {
                const uint32_t p = cur[a];
                if (state[p] == ST_CAND) continue;
                const uint32_t xy = s_xy[p];
                const int x = (int)(xy & 0xFFFFu), y = (int)(xy >> 16);
                const int cx = x / L.cs, cy = y / L.cs;
                const int cx0 = cx > 0 ? cx - 1 : 0, cx1 = cx + 1 < L.gw ? cx + 1 : L.gw - 1;
                const int cy0 = cy > 0 ? cy - 1 : 0, cy1 = cy + 1 < L.gh ? cy + 1 : L.gh - 1;
                bool sup = false;
                for (int yy = cy0; yy <= cy1 && !sup; yy++) {
                    const uint32_t q0 = cell_start[yy * L.gw + cx0], q1 = cell_start[yy * L.gw + cx1 + 1];
                    for (uint32_t q = q0; q < q1; q++) {
                        if (state[q] != ST_CAND) continue;
                        const uint32_t qxy = s_xy[q];
                        const long long dx = (int)(qxy & 0xFFFFu) - x, dy = (int)(qxy >> 16) - y;
                        if (dx * dx + dy * dy <= r2) { sup = true; break; }
                    }
                }
                if (sup) {
                    state[p] = ST_SUPPRESSED;
                } else {
                    nxt[atomicAdd(&sh_cnt, 1u)] = p; // order inside the list is irrelevant
                }
            }
            __syncthreads();
            for (int a = tid; a < n_act; a += NT) {
                const uint32_t p = cur[a];
                if (state[p] == ST_CAND) state[p] = ST_ACCEPTED;
            }
            n_act = (int)sh_cnt;
            uint32_t *t = cur; cur = nxt; nxt = t;
            __syncthreads();
        }
        // scatter acceptance to input order
        for (int p = tid; p < n; p += NT) accflag[s_idx[p]] = (state[p] == ST_ACCEPTED) ? 1 : 0;
        __syncthreads();
    } else {
        for (int i = tid; i < n; i += NT) accflag[i] = 1;
        __syncthreads();
    }

    // ---- emit accepted points in (score desc, index asc) order ----
    // one stable compaction pass per distinct score, highest first (FAST has <= 5 of them)
    const int per = (n + NT - 1) / NT;
    const int i0 = tid * per, i1 = (i0 + per < n) ? i0 + per : n;
    uint32_t base = 0;
    long long bound = (long long)INT32_MAX + 1; // scores strictly below this remain
    while (true) {
        if (tid == 0) sh_max = INT32_MIN;
        __syncthreads();
        int lm = INT32_MIN;
        bool have = false;
        for (int i = i0; i < i1; i++)
            if (accflag[i] && (long long)raw_score[i] < bound) { int s = raw_score[i]; if (!have || s > lm) lm = s; have = true; }
        if (have) atomicMax(&sh_max, lm);
        __syncthreads();
        // INT32_MIN doubles as "none": a real INT32_MIN score is still handled because `have` below re-tests
        const int m = sh_max;
        uint32_t cnt = 0;
        for (int i = i0; i < i1; i++) cnt += (accflag[i] && raw_score[i] == m && (long long)m < bound) ? 1u : 0u;
        uint32_t ex;
        const uint32_t tot = block_excl_scan(cnt, &ex, wsum);
        if (tot == 0) break; // uniform
        uint32_t o = base + ex;
        for (int i = i0; i < i1; i++)
            if (accflag[i] && raw_score[i] == m) { if (o < (uint32_t)kp_cap) order[o] = (uint32_t)i; o++; }
        base += tot;
        bound = m;
        __syncthreads();
    }
    if (tid == 0) {
        n_kept_all[f] = (int32_t)(base < (uint32_t)kp_cap ? base : (uint32_t)kp_cap);
        if (base > (uint32_t)kp_cap) atomicOr(status, (int)PGX_ST_KP_CAP);
    }
}

} // namespace

size_t pgx_nms_ws_bytes(int W, int H, int radius, int n_cap) { return nms_layout(W, H, radius, n_cap).total; }

void pgx_launch_nms(hipStream_t s, const uint32_t *raw_xy, const int32_t *raw_score, const int32_t *n_raw, int F,
                    int n_cap, int W, int H, int radius, void *ws, size_t ws_stride, uint32_t *order,
                    int32_t *n_kept, int kp_cap, int *status)
{
    if (F <= 0) return;
    hipLaunchKernelGGL(k_nms, dim3(F), dim3(NT), 0, s, raw_xy, raw_score, n_raw, n_cap, W, H, radius,
                       reinterpret_cast<unsigned char *>(ws), ws_stride, order, n_kept, kp_cap, status);
}
