// k_nms.hip -- RedundantKeypointEliminator.EliminateRedundantKeypoints on gfx950.
//
// Reference: ImageProcessing/RedundantKeypointEliminator.cs:16-39 -- stable sort by FastScore
// descending, then repeatedly keep the head and drop everything with
// sqrt(dx^2+dy^2) > r FALSE, i.e. dx^2+dy^2 <= r^2 (integers, so the test is exact).
//
// Parallel formulation (checked against the literal oracle): priority =
// (score desc, input index asc).  Round: every undecided point with no higher-priority
// undecided point within r is accepted; every undecided point within r of an accepted one is
// suppressed.  Output = accepted points in priority order.  On a 1080p frame with 1.3e5 raw
// hits the undecided set shrinks ~2.6x per round (9 rounds).
//
// Launch structure (all frames of a batch in every launch, blockIdx.y = frame):
//   k_nms_count / k_nms_cellscan / k_nms_scatter   counting sort of the points into a uniform
//        grid of cells >= r, all per-point arrays re-ordered into cell order so a neighbour
//        scan reads three contiguous runs;
//   WIDE_ROUNDS x k_nms_phase_cell<0>, <1>   one wavefront per cell, whole chip;
//   k_nms_tail   one 1024-thread workgroup per frame: finishes the few points still undecided
//        (rounds with an active list), then sorts the accepted points (bitonic, LDS) and
//        writes the order.
// Integer work on L2-resident data (a few MB per frame); latency-bound, not HBM-bound.
#include "pgx_internal.h"

namespace {

constexpr int NT = 1024;
constexpr int WIDE_ROUNDS = 10;
constexpr uint32_t SORT_LDS_MAX = 16384; // u64 keys -> 128 KiB

struct NmsLayout {
    int gw, gh, cs, ncell;
    size_t off_cellstart, off_cellfill, off_sxy, off_sscore, off_sidx, off_state, off_listA, off_listB, off_accflag,
        off_sortkeys, total;
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
    L.off_sortkeys = take((size_t)n_cap * 8);
    L.total = o;
    return L;
}

struct NmsPtrs {
    uint32_t *cell_start, *cell_fill, *s_xy, *s_idx, *listA, *listB;
    int32_t *s_score;
    uint8_t *state, *accflag;
    unsigned long long *sortkeys;
};

__device__ __forceinline__ NmsPtrs nms_ptrs(unsigned char *ws, const NmsLayout &L)
{
    NmsPtrs p;
    p.cell_start = reinterpret_cast<uint32_t *>(ws + L.off_cellstart);
    p.cell_fill = reinterpret_cast<uint32_t *>(ws + L.off_cellfill);
    p.s_xy = reinterpret_cast<uint32_t *>(ws + L.off_sxy);
    p.s_score = reinterpret_cast<int32_t *>(ws + L.off_sscore);
    p.s_idx = reinterpret_cast<uint32_t *>(ws + L.off_sidx);
    p.state = reinterpret_cast<uint8_t *>(ws + L.off_state);
    p.listA = reinterpret_cast<uint32_t *>(ws + L.off_listA);
    p.listB = reinterpret_cast<uint32_t *>(ws + L.off_listB);
    p.accflag = reinterpret_cast<uint8_t *>(ws + L.off_accflag);
    p.sortkeys = reinterpret_cast<unsigned long long *>(ws + L.off_sortkeys);
    return p;
}

// 0 undecided, 1 accepted in the running round, 2 suppressed, 3 accepted earlier
enum : uint8_t { ST_UNDECIDED = 0, ST_NEW = 1, ST_SUPPRESSED = 2, ST_ACCEPTED = 3 };

__device__ __forceinline__ bool better(int sq, uint32_t iq, int si, uint32_t ii)
{
    return sq > si || (sq == si && iq < ii);
}

__device__ __forceinline__ int clamp_n(const int32_t *n_raw_all, int f, int n_cap)
{
    int n = n_raw_all[f];
    return n < 0 ? 0 : (n > n_cap ? n_cap : n);
}

// exclusive scan of one value per thread over the 1024-thread block; returns the block total
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t *excl, uint32_t *wsum /*[16]*/)
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

__global__ __launch_bounds__(256) void k_nms_zero(int n_cap, int W, int H, int radius, unsigned char *ws_all, size_t ws_stride)
{
    const NmsLayout L = nms_layout(W, H, radius, n_cap);
    NmsPtrs P = nms_ptrs(ws_all + (size_t)blockIdx.y * ws_stride, L);
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c <= L.ncell) P.cell_fill[c] = 0;
}

__global__ __launch_bounds__(256) void k_nms_count(const uint32_t *__restrict__ raw_xy_all,
                                                   const int32_t *__restrict__ n_raw_all, int n_cap, int W, int H,
                                                   int radius, unsigned char *ws_all, size_t ws_stride)
{
    const int f = blockIdx.y;
    const int n = clamp_n(n_raw_all, f, n_cap);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const NmsLayout L = nms_layout(W, H, radius, n_cap);
    NmsPtrs P = nms_ptrs(ws_all + (size_t)f * ws_stride, L);
    const uint32_t xy = raw_xy_all[(size_t)f * n_cap + i];
    const int cx = (int)(xy & 0xFFFFu) / L.cs, cy = (int)(xy >> 16) / L.cs;
    atomicAdd(&P.cell_fill[cy * L.gw + cx], 1u);
}

__global__ __launch_bounds__(NT) void k_nms_cellscan(int n_cap, int W, int H, int radius, unsigned char *ws_all,
                                                     size_t ws_stride)
{
    __shared__ uint32_t wsum[NT / 64];
    const int f = blockIdx.x, tid = threadIdx.x;
    const NmsLayout L = nms_layout(W, H, radius, n_cap);
    NmsPtrs P = nms_ptrs(ws_all + (size_t)f * ws_stride, L);
    uint32_t carry = 0;
    for (int base = 0; base <= L.ncell; base += NT) {
        const int c = base + tid;
        const uint32_t v = (c < L.ncell) ? P.cell_fill[c] : 0u;
        uint32_t ex;
        const uint32_t tot = block_excl_scan(v, &ex, wsum);
        if (c <= L.ncell) P.cell_start[c] = carry + ex;
        if (c < L.ncell) P.cell_fill[c] = carry + ex; // becomes the scatter cursor
        carry += tot;
    }
}

__global__ __launch_bounds__(256) void k_nms_scatter(const uint32_t *__restrict__ raw_xy_all,
                                                     const int32_t *__restrict__ raw_score_all,
                                                     const int32_t *__restrict__ n_raw_all, int n_cap, int W, int H,
                                                     int radius, unsigned char *ws_all, size_t ws_stride)
{
    const int f = blockIdx.y;
    const int n = clamp_n(n_raw_all, f, n_cap);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const NmsLayout L = nms_layout(W, H, radius, n_cap);
    NmsPtrs P = nms_ptrs(ws_all + (size_t)f * ws_stride, L);
    const uint32_t xy = raw_xy_all[(size_t)f * n_cap + i];
    const int cx = (int)(xy & 0xFFFFu) / L.cs, cy = (int)(xy >> 16) / L.cs;
    const uint32_t pos = atomicAdd(&P.cell_fill[cy * L.gw + cx], 1u);
    P.s_xy[pos] = xy;
    P.s_score[pos] = raw_score_all[(size_t)f * n_cap + i];
    P.s_idx[pos] = (uint32_t)i;
    P.state[pos] = ST_UNDECIDED;
}

// phase A for one point: is there an undecided (at round start) better point within r?
__device__ __forceinline__ bool point_beaten(const NmsPtrs &P, const NmsLayout &L, uint32_t p, long long r2)
{
    const uint32_t xy = P.s_xy[p];
    const int x = (int)(xy & 0xFFFFu), y = (int)(xy >> 16);
    const int sc = P.s_score[p];
    const uint32_t id = P.s_idx[p];
    const int cx = x / L.cs, cy = y / L.cs;
    const int cx0 = cx > 0 ? cx - 1 : 0, cx1 = cx + 1 < L.gw ? cx + 1 : L.gw - 1;
    const int cy0 = cy > 0 ? cy - 1 : 0, cy1 = cy + 1 < L.gh ? cy + 1 : L.gh - 1;
    for (int yy = cy0; yy <= cy1; yy++) {
        const uint32_t q0 = P.cell_start[yy * L.gw + cx0], q1 = P.cell_start[yy * L.gw + cx1 + 1];
        for (uint32_t q = q0; q < q1; q++) {
            const uint8_t stq = P.state[q];
            if (stq != ST_UNDECIDED && stq != ST_NEW) continue; // NEW was undecided when the round began
            if (q == p) continue;
            const uint32_t qxy = P.s_xy[q];
            const long long dx = (int)(qxy & 0xFFFFu) - x, dy = (int)(qxy >> 16) - y;
            if (dx * dx + dy * dy > r2) continue;
            if (better(P.s_score[q], P.s_idx[q], sc, id)) return true;
        }
    }
    return false;
}

// phase B for one undecided point: is an accepted point within r?
__device__ __forceinline__ bool point_suppressed(const NmsPtrs &P, const NmsLayout &L, uint32_t p, long long r2)
{
    const uint32_t xy = P.s_xy[p];
    const int x = (int)(xy & 0xFFFFu), y = (int)(xy >> 16);
    const int cx = x / L.cs, cy = y / L.cs;
    const int cx0 = cx > 0 ? cx - 1 : 0, cx1 = cx + 1 < L.gw ? cx + 1 : L.gw - 1;
    const int cy0 = cy > 0 ? cy - 1 : 0, cy1 = cy + 1 < L.gh ? cy + 1 : L.gh - 1;
    for (int yy = cy0; yy <= cy1; yy++) {
        const uint32_t q0 = P.cell_start[yy * L.gw + cx0], q1 = P.cell_start[yy * L.gw + cx1 + 1];
        for (uint32_t q = q0; q < q1; q++) {
            const uint8_t stq = P.state[q];
            if (stq != ST_NEW && stq != ST_ACCEPTED) continue;
            const uint32_t qxy = P.s_xy[q];
            const long long dx = (int)(qxy & 0xFFFFu) - x, dy = (int)(qxy >> 16) - y;
            if (dx * dx + dy * dy <= r2) return true;
        }
    }
    return false;
}

// One wavefront per grid cell.  Lanes hold the 3x3-cell neighbourhood (three contiguous runs in
// cell order) in registers, 64 x NB_REG neighbours per pass; the wave then walks the cell's own
// undecided points one at a time -- every lane tests its neighbours against the broadcast centre
// and a ballot decides -- so work is (undecided centres) x (neighbours / 64) wave-instructions with
// all lanes busy, instead of (neighbours) x (centres / 64).
// PHASE 0 = "is a better undecided point within r" (phase A), PHASE 1 = "is an accepted point
// within r" (phase B, which also retires this cell's NEW marks to ACCEPTED).
constexpr int NB_REG = 4;     // neighbour registers per lane and pass (256 neighbours)
constexpr int CEN_MAX = 64;   // centre points staged per batch

template <int PHASE>
__global__ __launch_bounds__(256) void k_nms_phase_cell(const int32_t *__restrict__ n_raw_all, int n_cap, int W, int H,
                                                        int radius, unsigned char *ws_all, size_t ws_stride)
{
    __shared__ uint32_t c_xy[4][CEN_MAX];
    __shared__ int32_t c_score[4][CEN_MAX];
    __shared__ uint32_t c_idx[4][CEN_MAX];
    __shared__ uint32_t c_flag[4][CEN_MAX]; // bit0: undecided at entry, bit1: hit
    const int f = blockIdx.y, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (clamp_n(n_raw_all, f, n_cap) == 0) return;
    const NmsLayout L = nms_layout(W, H, radius, n_cap);
    const int c = blockIdx.x * 4 + wv;
    if (c >= L.ncell) return; // wave-uniform
    NmsPtrs P = nms_ptrs(ws_all + (size_t)f * ws_stride, L);
    const uint32_t p0 = P.cell_start[c], p1 = P.cell_start[c + 1];
    if (p0 == p1) return;
    const long long r2 = (long long)radius * (long long)radius;
    const int cy = c / L.gw, cx = c - cy * L.gw;
    const int cx0 = cx > 0 ? cx - 1 : 0, cx1 = cx + 1 < L.gw ? cx + 1 : L.gw - 1;
    // the three neighbour runs (rows cy-1, cy, cy+1 of cells), as one flat index space
    uint32_t rq[3], rlen[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const int yy = cy - 1 + k;
        if (yy < 0 || yy >= L.gh) { rq[k] = 0; rlen[k] = 0; continue; }
        rq[k] = P.cell_start[yy * L.gw + cx0];
        rlen[k] = P.cell_start[yy * L.gw + cx1 + 1] - rq[k];
    }
    const uint32_t ntot = rlen[0] + rlen[1] + rlen[2];

    for (uint32_t pb = p0; pb < p1; pb += CEN_MAX) { // centre batches
        const uint32_t p = pb + lane;
        const bool have = p < p1;
        uint8_t st = have ? P.state[p] : (uint8_t)ST_SUPPRESSED;
        if (PHASE == 1 && have && st == ST_NEW) { P.state[p] = ST_ACCEPTED; st = ST_ACCEPTED; }
        const bool und = have && st == ST_UNDECIDED;
        if (!__any(und)) continue;
        __builtin_amdgcn_wave_barrier();
        c_flag[wv][lane] = und ? 1u : 0u;
        if (und) {
            c_xy[wv][lane] = P.s_xy[p];
            if (PHASE == 0) { c_score[wv][lane] = P.s_score[p]; c_idx[wv][lane] = P.s_idx[p]; }
        }
        __builtin_amdgcn_wave_barrier();
        const int ncen = (int)((p1 - pb < (uint32_t)CEN_MAX) ? p1 - pb : (uint32_t)CEN_MAX);

        for (uint32_t nb0 = 0; nb0 < ntot; nb0 += 64 * NB_REG) { // neighbour passes
            uint32_t nq[NB_REG], nxy[NB_REG], nid[NB_REG];
            int32_t nsc[NB_REG];
            bool nok[NB_REG];
#pragma unroll
            for (int k = 0; k < NB_REG; k++) {
                const uint32_t fi = nb0 + k * 64 + lane;
                bool ok = fi < ntot;
                uint32_t q = 0;
                if (ok) q = fi < rlen[0] ? rq[0] + fi : (fi < rlen[0] + rlen[1] ? rq[1] + (fi - rlen[0]) : rq[2] + (fi - rlen[0] - rlen[1]));
                uint8_t sq = ok ? P.state[q] : (uint8_t)ST_SUPPRESSED;
                if (PHASE == 0) ok = ok && (sq == ST_UNDECIDED || sq == ST_NEW); // NEW was undecided when the round began
                else ok = ok && (sq == ST_NEW || sq == ST_ACCEPTED);
                nq[k] = q;
                nok[k] = ok;
                nxy[k] = ok ? P.s_xy[q] : 0u;
                if (PHASE == 0) { nsc[k] = ok ? P.s_score[q] : 0; nid[k] = ok ? P.s_idx[q] : 0u; }
            }
            const int nchunks = (int)(((ntot - nb0 < (uint32_t)(64 * NB_REG) ? ntot - nb0 : (uint32_t)(64 * NB_REG)) + 63) / 64);
            for (int k = 0; k < ncen; k++) {
                const uint32_t fl = c_flag[wv][k];
                if (fl != 1u) continue; // decided, or already hit in an earlier pass (uniform)
                const uint32_t cxy = c_xy[wv][k];
                const int x = (int)(cxy & 0xFFFFu), y = (int)(cxy >> 16);
                const int sc = PHASE == 0 ? c_score[wv][k] : 0;
                const uint32_t id = PHASE == 0 ? c_idx[wv][k] : 0u;
                bool hit = false;
#pragma unroll
                for (int j = 0; j < NB_REG; j++) {
                    if (j >= nchunks) break; // uniform
                    const long long dx = (int)(nxy[j] & 0xFFFFu) - x, dy = (int)(nxy[j] >> 16) - y;
                    bool h = nok[j] && (dx * dx + dy * dy <= r2);
                    if (PHASE == 0) h = h && nq[j] != pb + (uint32_t)k && better(nsc[j], nid[j], sc, id);
                    hit = hit || h;
                }
                if (__any(hit)) { if (lane == 0) c_flag[wv][k] = 3u; }
            }
            __builtin_amdgcn_wave_barrier();
        }
        // decisions of this batch
        if (und) {
            const bool hit = (c_flag[wv][lane] & 2u) != 0u;
            if (PHASE == 0) { if (!hit) P.state[p] = ST_NEW; }
            else { if (hit) P.state[p] = ST_SUPPRESSED; }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

__device__ void bitonic_sort_u64(unsigned long long *keys, uint32_t n2p)
{
    const uint32_t tid = threadIdx.x, nth = blockDim.x;
    for (uint32_t k = 2; k <= n2p; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            __syncthreads();
            for (uint32_t t = tid; t < n2p / 2; t += nth) {
                const uint32_t lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const uint32_t hi = lo | j;
                const bool up = (lo & k) == 0;
                const unsigned long long a = keys[lo], b = keys[hi];
                if ((a > b) == up) { keys[lo] = b; keys[hi] = a; }
            }
        }
    }
    __syncthreads();
}

__global__ __launch_bounds__(NT) void k_nms_tail(const int32_t *__restrict__ raw_score_all,
                                                 const int32_t *__restrict__ n_raw_all, int n_cap, int W, int H,
                                                 int radius, int wide_done, unsigned char *ws_all, size_t ws_stride,
                                                 uint32_t *__restrict__ order_all, int32_t *__restrict__ n_kept_all,
                                                 int kp_cap, int *status)
{
    extern __shared__ unsigned long long lds_keys[];
    __shared__ uint32_t wsum[NT / 64];
    __shared__ uint32_t sh_cnt;
    __shared__ int sh_max;
    const int f = blockIdx.x, tid = threadIdx.x;
    const int n = clamp_n(n_raw_all, f, n_cap);
    const int32_t *raw_score = raw_score_all + (size_t)f * n_cap;
    uint32_t *order = order_all + (size_t)f * kp_cap;
    const NmsLayout L = nms_layout(W, H, radius, n_cap);
    NmsPtrs P = nms_ptrs(ws_all + (size_t)f * ws_stride, L);
    if (n == 0) {
        if (tid == 0) n_kept_all[f] = 0;
        return;
    }
    const long long r2 = (long long)radius * (long long)radius;

    if (radius >= 0 && wide_done) {
        // ---- remaining rounds on an active list ----
        if (tid == 0) sh_cnt = 0;
        __syncthreads();
        for (int p = tid; p < n; p += NT)
            if (P.state[p] == ST_UNDECIDED) P.listA[atomicAdd(&sh_cnt, 1u)] = (uint32_t)p;
        __syncthreads();
        uint32_t *cur = P.listA, *nxt = P.listB;
        int n_act = (int)sh_cnt;
        __syncthreads();
        while (n_act > 0) {
            for (int a = tid; a < n_act; a += NT) {
                const uint32_t p = cur[a];
                if (!point_beaten(P, L, p, r2)) P.state[p] = ST_NEW;
            }
            __syncthreads();
            if (tid == 0) sh_cnt = 0;
            __syncthreads();
            for (int a = tid; a < n_act; a += NT) {
                const uint32_t p = cur[a];
                if (P.state[p] == ST_NEW) continue;
                if (point_suppressed(P, L, p, r2)) P.state[p] = ST_SUPPRESSED;
                else nxt[atomicAdd(&sh_cnt, 1u)] = p; // order inside the list is irrelevant
            }
            __syncthreads();
            for (int a = tid; a < n_act; a += NT) {
                const uint32_t p = cur[a];
                if (P.state[p] == ST_NEW) P.state[p] = ST_ACCEPTED;
            }
            n_act = (int)sh_cnt;
            uint32_t *t = cur; cur = nxt; nxt = t;
            __syncthreads();
        }
    }

    // ---- gather the accepted points: key = (score descending, input index ascending) ----
    const bool all_accepted = radius < 0; // distance > r is always true for r < 0
    if (tid == 0) sh_cnt = 0;
    __syncthreads();
    for (int p = tid; p < n; p += NT) {
        const bool acc = all_accepted || P.state[p] == ST_ACCEPTED;
        if (acc) {
            const int sc = all_accepted ? raw_score[p] : P.s_score[p];
            const uint32_t id = all_accepted ? (uint32_t)p : P.s_idx[p];
            const uint32_t inv = ~((uint32_t)sc ^ 0x80000000u); // larger score -> smaller key
            P.sortkeys[atomicAdd(&sh_cnt, 1u)] = ((unsigned long long)inv << 32) | id;
        }
    }
    __syncthreads();
    const uint32_t nacc = sh_cnt;
    const uint32_t n2p = [](uint32_t v) { uint32_t p = 1; while (p < v) p <<= 1; return p; }(nacc > 1 ? nacc : 1);
    if (n2p <= SORT_LDS_MAX) {
        for (uint32_t i = tid; i < n2p; i += NT) lds_keys[i] = i < nacc ? P.sortkeys[i] : ~0ull;
        bitonic_sort_u64(lds_keys, n2p);
        for (uint32_t i = tid; i < nacc && i < (uint32_t)kp_cap; i += NT) order[i] = (uint32_t)lds_keys[i];
    } else {
        // large survivor sets: one stable compaction pass per distinct score, highest first
        for (int i = tid; i < n; i += NT) P.accflag[i] = 0;
        __syncthreads();
        for (uint32_t i = tid; i < nacc; i += NT) P.accflag[(uint32_t)P.sortkeys[i]] = 1;
        __syncthreads();
        const int per = (n + NT - 1) / NT;
        const int i0 = tid * per, i1 = (i0 + per < n) ? i0 + per : n;
        uint32_t base = 0;
        long long bound = (long long)INT32_MAX + 1;
        while (true) {
            if (tid == 0) sh_max = INT32_MIN;
            __syncthreads();
            int lm = INT32_MIN;
            bool have = false;
            for (int i = i0; i < i1; i++)
                if (P.accflag[i] && (long long)raw_score[i] < bound) { int s = raw_score[i]; if (!have || s > lm) lm = s; have = true; }
            if (have) atomicMax(&sh_max, lm);
            __syncthreads();
            const int m = sh_max;
            uint32_t cnt = 0;
            for (int i = i0; i < i1; i++) cnt += (P.accflag[i] && raw_score[i] == m && (long long)m < bound) ? 1u : 0u;
            uint32_t ex;
            const uint32_t tot = block_excl_scan(cnt, &ex, wsum);
            if (tot == 0) break;
            uint32_t o = base + ex;
            for (int i = i0; i < i1; i++)
                if (P.accflag[i] && raw_score[i] == m) { if (o < (uint32_t)kp_cap) order[o] = (uint32_t)i; o++; }
            base += tot;
            bound = m;
            __syncthreads();
        }
    }
    if (tid == 0) {
        n_kept_all[f] = (int32_t)(nacc < (uint32_t)kp_cap ? nacc : (uint32_t)kp_cap);
        if (nacc > (uint32_t)kp_cap) atomicOr(status, (int)PGX_ST_KP_CAP);
    }
}

} // namespace

size_t pgx_nms_ws_bytes(int W, int H, int radius, int n_cap) { return nms_layout(W, H, radius, n_cap).total; }

void pgx_launch_nms(hipStream_t s, const uint32_t *raw_xy, const int32_t *raw_score, const int32_t *n_raw, int F,
                    int n_cap, int W, int H, int radius, void *wsv, size_t ws_stride, uint32_t *order,
                    int32_t *n_kept, int kp_cap, int *status)
{
    if (F <= 0) return;
    unsigned char *ws = reinterpret_cast<unsigned char *>(wsv);
    const NmsLayout L = nms_layout(W, H, radius, n_cap);
    const bool rounds = radius >= 0;
    const dim3 pgrid((n_cap + 255) / 256, F);
    const dim3 cgrid((L.ncell + 3) / 4, F);
    if (rounds) {
        hipLaunchKernelGGL(k_nms_zero, dim3((L.ncell + 256) / 256, F), dim3(256), 0, s, n_cap, W, H, radius, ws, ws_stride);
        hipLaunchKernelGGL(k_nms_count, pgrid, dim3(256), 0, s, raw_xy, n_raw, n_cap, W, H, radius, ws, ws_stride);
        hipLaunchKernelGGL(k_nms_cellscan, dim3(F), dim3(NT), 0, s, n_cap, W, H, radius, ws, ws_stride);
        hipLaunchKernelGGL(k_nms_scatter, pgrid, dim3(256), 0, s, raw_xy, raw_score, n_raw, n_cap, W, H, radius, ws,
                           ws_stride);
        for (int r = 0; r < WIDE_ROUNDS; r++) {
            hipLaunchKernelGGL(k_nms_phase_cell<0>, cgrid, dim3(256), 0, s, n_raw, n_cap, W, H, radius, ws, ws_stride);
            hipLaunchKernelGGL(k_nms_phase_cell<1>, cgrid, dim3(256), 0, s, n_raw, n_cap, W, H, radius, ws, ws_stride);
        }
    }
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_nms_tail), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)(SORT_LDS_MAX * 8));
        attr_set = true;
    }
    hipLaunchKernelGGL(k_nms_tail, dim3(F), dim3(NT), SORT_LDS_MAX * 8, s, raw_score, n_raw, n_cap, W, H, radius,
                       rounds ? 1 : 0, ws, ws_stride, order, n_kept, kp_cap, status);
}
