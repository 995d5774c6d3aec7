// k_nms.hip -- RedundantKeypointEliminator.EliminateRedundantKeypoints on gfx950.
//
// Reference: ImageProcessing/RedundantKeypointEliminator.cs:16-39 -- stable sort by FastScore
// descending, then repeatedly keep the head and drop everything with
// sqrt(dx^2+dy^2) > r FALSE, i.e. dx^2+dy^2 <= r^2 (integers, so the test is exact).
//
// Parallel formulation (checked against the literal oracle): priority =
// (score desc, input index asc).  Round: every undecided point with no higher-priority
// undecided point within r is accepted (phase A); every undecided point within r of an accepted
// one is suppressed (phase B).  Output = accepted points in priority order.  On a 1080p frame with
// 1e5 raw hits and r = 16 the undecided set roughly halves per round (about ten rounds); larger radii on dense
// hits form long dependency chains and need dozens.
//
// Two launch structures (all frames of a batch in every launch, blockIdx.y = frame).  The fused detect path
// with 10 <= r <= 192 uses the CHAMPION ROUNDS described further down (k_nms_bin_planes, k_nms_champ,
// k_nms_phase_c); everything else (stage API, other radii) uses the general one:
//   k_nms_zero / k_nms_count / k_nms_cellscan / k_nms_scatter   counting sort of the points into a
//        uniform grid of cells >= r; each point becomes one 16-byte record {xy, score, index, state}
//        in cell order, so a 3x3-cell neighbourhood is three contiguous runs of records;
//   WIDE_ROUNDS x (k_nms_phase_a, k_nms_push), whole chip.
//        phase A: one wavefront per cell that still has undecided points; lanes hold the neighbourhood
//                 in registers, the cell's undecided points are taken one at a time (own cell first,
//                 then the rest) and a ballot decides "beaten"; survivors are accepted and queued;
//        phase B: one wavefront per NEWLY accepted point (a few thousand per frame in total): lanes =
//                 its neighbours, every undecided one within r is suppressed; emits the sort key;
//   k_nms_tail   one 1024-thread workgroup per frame: finishes whatever is still undecided (champion path: the
//        same champion rounds on a compact list of the open cells, a block barrier instead of a kernel
//        boundary -- long dependency chains need dozens of nearly empty rounds), then orders the accepted
//        points (champion path: per-score-level rank bitmaps; otherwise a bitonic sort in LDS).
// Integer work on L2-resident data (a few MB per frame); latency/issue-bound, not HBM-bound.
#include "pgx_internal.h"
#include <cstdlib>

namespace {

constexpr int NT = 1024;
constexpr int WIDE_ROUNDS = 8; // champion path: 3..10 measure the same on 128 1080p frames (r = 16); 8..10 are best on 4K frames at r = 30
constexpr uint32_t SORT_LDS_MAX = 16384; // u64 keys -> 128 KiB

struct NmsLayout {
    int gw, gh, cs, ncell;
    int R;     // neighbourhood reach in cells: every point within r of a cell's point lies in the (2R+1)^2 block
    int champ; // 1: "champion" rounds (cells small enough that any two points of a cell are within r)
    int cgw;   // width of the champion grid, padded by R empty cells on every side
    int mask;  // 1: champion rounds on per-cell 64-bit pixel masks (8-pixel cells), no per-hit records at all
    size_t off_cellstart, off_cellfill, off_cellund, off_counters, off_rec, off_listA, off_listB, off_accflag, off_sortkeys,
        off_champ, off_alive, off_planes, off_disk, total;
};

// champion rounds need: the FAST planes (cells aligned to the 64-pixel segments, scores 12..16, raster ranks),
// a cell size cs in {8,16,32,64} with 2*(cs-1)^2 <= r^2, reach <= 3 cells, and ranks that fit 24 bits
__host__ __device__ inline int nms_champ_cs(int radius, int n_cap, bool planes)
{
    if (!planes || radius < 10 || radius > 192 || n_cap > (1 << 24)) return 0;
    return radius >= 90 ? 64 : (radius >= 44 ? 32 : (radius >= 22 ? 16 : 8));
}

// mask rounds: 8-pixel cells (r = 10..21) on frames whose coordinates fit 14 bits each (position-ordered 32-bit keys)
__host__ __device__ inline bool nms_mask_ok(int W, int H, int radius, int n_cap, bool planes)
{
    return nms_champ_cs(radius, n_cap, planes) == 8 && W <= 16384 && H <= 16384;
}

__host__ __device__ inline NmsLayout nms_layout(int W, int H, int radius, int n_cap, bool planes)
{
    NmsLayout L;
    const int ccs = nms_champ_cs(radius, n_cap, planes);
    // otherwise 16/32/64 keep cells aligned with the 64-pixel FAST row segments (plane-based binning); larger radii use r
    int cs = ccs ? ccs : (radius <= 16 ? 16 : (radius <= 32 ? 32 : (radius <= 64 ? 64 : radius)));
    L.cs = cs;
    L.champ = ccs ? 1 : 0;
    L.R = ccs ? (radius + cs - 1) / cs : 1;
    L.gw = (W + cs - 1) / cs; if (L.gw < 1) L.gw = 1;
    L.gh = (H + cs - 1) / cs; if (L.gh < 1) L.gh = 1;
    L.ncell = L.gw * L.gh;
    L.cgw = L.gw + 2 * L.R;
    L.mask = nms_mask_ok(W, H, radius, n_cap, planes) ? 1 : 0;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o += (bytes + 255) & ~(size_t)255; return r; };
    if (L.mask) { // per padded cell: alive word, three score planes, champion key; per frame: open-cell lists, the accepted points' keys
        const size_t npc = (size_t)L.cgw * (L.gh + 2 * L.R);
        L.off_cellstart = L.off_cellfill = L.off_cellund = L.off_rec = 0;
        L.off_counters = take(64);
        L.off_accflag = take(n_cap > (1 << 20) ? (size_t)n_cap : 0); // only the output order of > 1M raw hits uses it
        L.off_alive = take(npc * 8);
        L.off_planes = take(npc * 24);
        L.off_champ = take(npc * 4);
        L.off_listA = take((size_t)L.ncell * 4);
        L.off_listB = take((size_t)L.ncell * 4);
        L.off_sortkeys = take((size_t)L.ncell * 8); // at most one accepted point per cell
        L.off_disk = take((size_t)(2 * L.R + 1) * (2 * L.R + 1) * 64 * 8);
        L.total = o;
        return L;
    }
    L.off_alive = L.off_planes = L.off_disk = 0;
    L.off_cellstart = take((size_t)(L.ncell + 1) * 4);
    L.off_cellfill = take((size_t)(L.ncell + 1) * 4);
    L.off_cellund = take((size_t)(L.ncell + 1) * 4);
    L.off_counters = take(64);
    L.off_rec = take((size_t)n_cap * 16);
    L.off_listA = take((size_t)n_cap * 4);
    L.off_listB = take((size_t)n_cap * 4);
    L.off_accflag = take((size_t)n_cap);
    L.off_sortkeys = take((size_t)n_cap * 8);
    L.off_champ = take(L.champ ? (size_t)L.cgw * (L.gh + 2 * L.R) * 8 : 0);
    L.total = o;
    return L;
}

// rec = {x: y<<16|x, y: score, z: input index, w: state}
struct NmsPtrs {
    uint32_t *cell_start, *cell_fill, *cell_und, *listA, *listB;
    uint32_t *counters; // [0],[1] newly accepted this round (by round parity), [2] accepted so far (sort keys)
    uint4 *rec;
    uint8_t *accflag;
    unsigned long long *sortkeys;
    uint2 *champ; // champion rounds: per cell {priority key of its best undecided point (0 = none), its xy}, padded grid
};

// champion rounds fill the raster-order raw lists (xy, score by raster rank) for the points they accept: the
// descriptor stage looks the kept points up there, and the full lists are never needed (k_fast_compact is skipped)
struct RawOut { uint32_t *xy; int32_t *score; };

__device__ __forceinline__ NmsPtrs nms_ptrs(unsigned char *ws, const NmsLayout &L)
{
    NmsPtrs p;
    p.cell_start = reinterpret_cast<uint32_t *>(ws + L.off_cellstart);
    p.cell_fill = reinterpret_cast<uint32_t *>(ws + L.off_cellfill);
    p.cell_und = reinterpret_cast<uint32_t *>(ws + L.off_cellund);
    p.counters = reinterpret_cast<uint32_t *>(ws + L.off_counters);
    p.rec = reinterpret_cast<uint4 *>(ws + L.off_rec);
    p.listA = reinterpret_cast<uint32_t *>(ws + L.off_listA);
    p.listB = reinterpret_cast<uint32_t *>(ws + L.off_listB);
    p.accflag = reinterpret_cast<uint8_t *>(ws + L.off_accflag);
    p.sortkeys = reinterpret_cast<unsigned long long *>(ws + L.off_sortkeys);
    p.champ = reinterpret_cast<uint2 *>(ws + L.off_champ);
    return p;
}

__device__ __forceinline__ uint32_t *rec_state(uint4 *rec, uint32_t p) { return reinterpret_cast<uint32_t *>(rec + p) + 3; }

// 0 undecided, 1 accepted in the running round, 2 suppressed, 3 accepted earlier
// champion rounds stamp an accepted point with ST_ACC_ROUND + round: "accepted earlier in the running phase" (it
// was undecided when the round began) can then be told from "accepted in an earlier round" without a fix-up pass
enum : uint32_t { ST_UNDECIDED = 0, ST_NEW = 1, ST_SUPPRESSED = 2, ST_ACCEPTED = 3, ST_ACC_ROUND = 4 };

__device__ __forceinline__ bool better(int sq, uint32_t iq, int si, uint32_t ii)
{
    return sq > si || (sq == si && iq < ii);
}

// champion rounds: 32-bit priority key of a FAST hit, larger = better, never 0 (score 12..16, rank < 2^24)
__device__ __forceinline__ uint32_t champ_key(uint32_t score, uint32_t idx) { return ((score - 11u) << 24) | (0xFFFFFFu - idx); }

__device__ __forceinline__ bool within(uint32_t axy, uint32_t bxy, long long r2)
{
    const long long dx = (int)(axy & 0xFFFFu) - (int)(bxy & 0xFFFFu), dy = (int)(axy >> 16) - (int)(bxy >> 16);
    return dx * dx + dy * dy <= r2;
}

// same test for points known to be a few cells apart (champion rounds: |d| <= 7 * 64, r <= 192)
__device__ __forceinline__ bool within_near(uint32_t axy, uint32_t bxy, int r2)
{
    const int dx = (int)(axy & 0xFFFFu) - (int)(bxy & 0xFFFFu), dy = (int)(axy >> 16) - (int)(bxy >> 16);
    return dx * dx + dy * dy <= r2;
}

__device__ __forceinline__ int clamp_n(const int32_t *n_raw_all, int f, int n_cap)
{
    int n = n_raw_all[f];
    return n < 0 ? 0 : (n > n_cap ? n_cap : n);
}

// exclusive scan of one value per thread over the block (<= 1024 threads); returns the block total
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t *excl, uint32_t *wsum /*[16]*/)
{
    const int nwv = (int)blockDim.x >> 6;
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
    for (int w = 0; w < nwv; w++) {
        uint32_t s = wsum[w];
        if (w < wv) woff += s;
        total += s;
    }
    *excl = woff + incl - v;
    return total;
}

__global__ __launch_bounds__(256) void k_nms_zero(NmsLayout L, unsigned char *ws_all, size_t ws_stride)
{
    NmsPtrs P = nms_ptrs(ws_all + (size_t)blockIdx.y * ws_stride, L);
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c <= L.ncell) P.cell_fill[c] = 0;
    if (c < 16) P.counters[c] = 0;
    if (L.champ) { // border cells stay {0,0} = "no champion"; the interior is rewritten every round
        const int nch = L.cgw * (L.gh + 2 * L.R);
        for (int i = c; i < nch; i += gridDim.x * 256) P.champ[i] = make_uint2(0u, 0u);
    }
}

__global__ __launch_bounds__(256) void k_nms_count(const uint32_t *__restrict__ raw_xy_all,
                                                   const int32_t *__restrict__ n_raw_all, int n_cap, NmsLayout L,
                                                   unsigned char *ws_all, size_t ws_stride)
{
    const int f = blockIdx.y;
    const int n = clamp_n(n_raw_all, f, n_cap);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    NmsPtrs P = nms_ptrs(ws_all + (size_t)f * ws_stride, L);
    const uint32_t xy = raw_xy_all[(size_t)f * n_cap + i];
    const int cx = (int)(xy & 0xFFFFu) / L.cs, cy = (int)(xy >> 16) / L.cs;
    atomicAdd(&P.cell_fill[cy * L.gw + cx], 1u);
}

__global__ __launch_bounds__(NT) void k_nms_cellscan(NmsLayout L, unsigned char *ws_all, size_t ws_stride)
{
    __shared__ uint32_t wsum[NT / 64];
    const int f = blockIdx.x, tid = threadIdx.x;
    NmsPtrs P = nms_ptrs(ws_all + (size_t)f * ws_stride, L);
    uint32_t carry = 0;
    for (int base = 0; base <= L.ncell; base += NT) {
        const int c = base + tid;
        const uint32_t v = (c < L.ncell) ? P.cell_fill[c] : 0u;
        uint32_t ex;
        const uint32_t tot = block_excl_scan(v, &ex, wsum);
        if (c <= L.ncell) P.cell_start[c] = carry + ex;
        if (c < L.ncell) { P.cell_fill[c] = carry + ex; P.cell_und[c] = v; } // scatter cursor; undecided = all
        carry += tot;
    }
}

__global__ __launch_bounds__(256) void k_nms_scatter(const uint32_t *__restrict__ raw_xy_all,
                                                     const int32_t *__restrict__ raw_score_all,
                                                     const int32_t *__restrict__ n_raw_all, int n_cap, NmsLayout L,
                                                     unsigned char *ws_all, size_t ws_stride)
{
    const int f = blockIdx.y;
    const int n = clamp_n(n_raw_all, f, n_cap);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    NmsPtrs P = nms_ptrs(ws_all + (size_t)f * ws_stride, L);
    const uint32_t xy = raw_xy_all[(size_t)f * n_cap + i];
    const int cx = (int)(xy & 0xFFFFu) / L.cs, cy = (int)(xy >> 16) / L.cs;
    const uint32_t pos = atomicAdd(&P.cell_fill[cy * L.gw + cx], 1u);
    P.rec[pos] = make_uint4(xy, (uint32_t)raw_score_all[(size_t)f * n_cap + i], (uint32_t)i, ST_UNDECIDED);
}

// ---- atomic-free binning straight from the FAST ballot planes (fused detect path, cs in {16,32,64}) ----
// seg[F][H][ntx][4] = three score-bit planes + count per 64-pixel row segment; segoff = raster-order offsets.
// One thread per cell: MODE 0 counts the hits of its cs x cs pixels into cell_fill, MODE 1 (after the
// cell scan) writes their records at cell_start[c]... in raster order.  The input index of a hit is its
// raster rank = segoff + popcount(lower bits), exactly what k_fast_compact wrote into the raw list.
// 8-pixel cells (r = 10..21, the common case): the cell's 8x8 bits of each plane are packed into ONE 64-bit word
// (bit = 8 * row + column, raster order inside the cell) right after the loads, so the row data lives in 14
// registers instead of 56 and eight wavefronts per SIMD hide the load latency.  Own kernel: inside the general
// one its register count would still be the general path's.
template <int MODE>
__global__ __launch_bounds__(256) void k_nms_bin_planes8(const unsigned long long *__restrict__ seg_all,
                                                         const uint32_t *__restrict__ segoff_all, int W, int H, int ntx,
                                                         int n_cap, NmsLayout L, unsigned char *ws_all, size_t ws_stride)
{
    const int f = blockIdx.y;
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= L.ncell) return;
    NmsPtrs P = nms_ptrs(ws_all + (size_t)f * ws_stride, L);
    const int cy = c / L.gw, cx = c - cy * L.gw;
    const int x0 = cx * 8, tx = x0 >> 6, bo = x0 & 63;
    const int y0 = cy * 8, y1 = (y0 + 8 < H) ? y0 + 8 : H;
    const size_t nseg = (size_t)H * ntx;
    const unsigned long long *seg = seg_all + (size_t)f * nseg * 4;
    const uint32_t *segoff = segoff_all + (size_t)f * nseg;
    unsigned long long A0 = 0, A1 = 0, A2 = 0;
    uint32_t base[8];
    {
        ulonglong2 rp8[8];
        unsigned long long rb8[8];
        uint32_t so8[8];
#pragma unroll
        for (int i = 0; i < 8; i++) { // all loads first
            const bool ok = y0 + i < y1;
            const size_t si = (size_t)(ok ? y0 + i : y0) * ntx + tx;
            rp8[i] = *reinterpret_cast<const ulonglong2 *>(seg + si * 4);
            rb8[i] = seg[si * 4 + 2];
            so8[i] = segoff[si];
            if (!ok) { rp8[i] = make_ulonglong2(0ull, 0ull); rb8[i] = 0ull; }
        }
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const unsigned long long any = rp8[i].x | rp8[i].y | rb8[i];
            base[i] = so8[i] + (uint32_t)__popcll(any & ((1ull << bo) - 1ull)); // raster rank of the row slice's first hit
            // hits past the raw capacity are dropped everywhere (PGX_E_CAPACITY is raised): keep the first n_cap - base
            uint32_t keep = (uint32_t)(any >> bo) & 0xFFu;
            const uint32_t room = base[i] < (uint32_t)n_cap ? (uint32_t)n_cap - base[i] : 0u;
            while ((uint32_t)__popc(keep) > room) keep &= ~(1u << (31 - __builtin_clz(keep)));
            A0 |= ((rp8[i].x >> bo) & (unsigned long long)keep) << (8 * i);
            A1 |= ((rp8[i].y >> bo) & (unsigned long long)keep) << (8 * i);
            A2 |= ((rb8[i] >> bo) & (unsigned long long)keep) << (8 * i);
        }
    }
    const unsigned long long any64 = A0 | A1 | A2;
    if (MODE == 0) { P.cell_fill[c] = (uint32_t)__popcll(any64); return; }
    const uint32_t first = P.cell_start[c];
    // codes: 5 = 101, 4 = 100, 3 = 011, 2 = 010, 1 = 001 (planes A2 A1 A0); MODE 2 = priority order, see below
    uint32_t q5 = first, q4 = first, q3 = first, q2 = first, q1 = first;
    if (MODE == 2) {
        const uint32_t c5 = (uint32_t)__popcll(A2 & A0), c4 = (uint32_t)__popcll(A2 & ~A0);
        const uint32_t c3 = (uint32_t)__popcll(~A2 & A1 & A0), c2 = (uint32_t)__popcll(~A2 & A1 & ~A0);
        q4 += c5; q3 += c5 + c4; q2 += c5 + c4 + c3; q1 += c5 + c4 + c3 + c2;
    }
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint32_t m = (uint32_t)(any64 >> (8 * i)) & 0xFFu;
        const uint32_t r0 = (uint32_t)(A0 >> (8 * i)) & 0xFFu, r1 = (uint32_t)(A1 >> (8 * i)) & 0xFFu, r2 = (uint32_t)(A2 >> (8 * i)) & 0xFFu;
        uint32_t idx = base[i];
        while (m) {
            const int l = __builtin_ctz(m);
            m &= m - 1;
            const int code = (int)((r0 >> l) & 1u) | ((int)((r1 >> l) & 1u) << 1) | ((int)((r2 >> l) & 1u) << 2);
            uint32_t at;
            if (MODE == 2) {
                at = code == 5 ? q5 : (code == 4 ? q4 : (code == 3 ? q3 : (code == 2 ? q2 : q1)));
                q5 += code == 5; q4 += code == 4; q3 += code == 3; q2 += code == 2; q1 += code == 1;
            } else {
                at = q1++;
            }
            P.rec[at] = make_uint4(((uint32_t)(y0 + i) << 16) | (uint32_t)(x0 + l), (uint32_t)(code + 11), idx, ST_UNDECIDED);
            idx++;
        }
    }
    if (MODE == 2) {
        uint2 ch = make_uint2(0u, 0u);
        if (q1 > first) { const uint4 r0 = P.rec[first]; ch = make_uint2(champ_key(r0.y, r0.z), r0.x); }
        P.champ[(cy + L.R) * L.cgw + cx + L.R] = ch;
    }
}

template <int MODE>
__global__ __launch_bounds__(256) void k_nms_bin_planes(const unsigned long long *__restrict__ seg_all,
                                                        const uint32_t *__restrict__ segoff_all, int W, int H, int ntx,
                                                        int n_cap, NmsLayout L, unsigned char *ws_all, size_t ws_stride)
{
    const int f = blockIdx.y;
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= L.ncell) return;
    NmsPtrs P = nms_ptrs(ws_all + (size_t)f * ws_stride, L);
    const int cy = c / L.gw, cx = c - cy * L.gw;
    const int x0 = cx * L.cs, tx = x0 >> 6, bo = x0 & 63;
    const unsigned long long cmask = (L.cs >= 64 ? ~0ull : ((1ull << L.cs) - 1ull)) << bo;
    const int y0 = cy * L.cs, y1 = (y0 + L.cs < H) ? y0 + L.cs : H;
    const size_t nseg = (size_t)H * ntx;
    const unsigned long long *seg = seg_all + (size_t)f * nseg * 4;
    const uint32_t *segoff = segoff_all + (size_t)f * nseg;
    // Rows go through in chunks of eight with all loads of a chunk issued before the first use (a load ->
    // use loop is one L2 round trip per row, and a thread walks up to 64 rows).
    constexpr int CH = 8;
    ulonglong2 rp[CH];
    unsigned long long rb[CH];
    uint32_t rso[CH];
    auto load_chunk = [&](int yc) {
#pragma unroll
        for (int i = 0; i < CH; i++) {
            const bool ok = yc + i < y1;
            const size_t si = (size_t)(ok ? yc + i : y0) * ntx + tx;
            rp[i] = *reinterpret_cast<const ulonglong2 *>(seg + si * 4);
            rb[i] = seg[si * 4 + 2];
            rso[i] = segoff[si];
            if (!ok) { rp[i] = make_ulonglong2(0ull, 0ull); rb[i] = 0ull; }
        }
    };
    // hits past the raw capacity are dropped everywhere (PGX_E_CAPACITY is raised): of a row slice whose
    // first hit has raster rank idx0 only the first n_cap - idx0 hits exist
    auto cut = [&](unsigned long long m, uint32_t idx0) {
        const uint32_t room = idx0 < (uint32_t)n_cap ? (uint32_t)n_cap - idx0 : 0u;
        while ((uint32_t)__popcll(m) > room) m &= ~(1ull << (63 - __builtin_clzll(m)));
        return m;
    };
    const unsigned long long below = (1ull << bo) - 1ull;

    if (MODE == 0) {
        uint32_t count = 0;
        for (int yc = y0; yc < y1; yc += CH) {
            load_chunk(yc);
#pragma unroll
            for (int i = 0; i < CH; i++) {
                const unsigned long long any = rp[i].x | rp[i].y | rb[i];
                const uint32_t cnt = (uint32_t)__popcll(any & cmask);
                const uint32_t idx0 = rso[i] + (uint32_t)__popcll(any & below);
                const uint32_t room = idx0 < (uint32_t)n_cap ? (uint32_t)n_cap - idx0 : 0u;
                count += cnt < room ? cnt : room;
            }
        }
        P.cell_fill[c] = count;
        return;
    }

    const uint32_t first = P.cell_start[c];
    // MODE 1: raster order.  MODE 2 (champion rounds): PRIORITY order (score descending, raster rank ascending),
    // so that the best undecided point of a cell is always the first undecided record of its run: pass 1 counts
    // the hits per score level (-> start of each level's sub-run), pass 2 places every hit.
    uint32_t q5 = first, q4 = first, q3 = first, q2 = first, q1 = first;
    auto count_levels = [&]() {
#pragma unroll
        for (int i = 0; i < CH; i++) {
            const unsigned long long any = rp[i].x | rp[i].y | rb[i];
            if (!(any & cmask)) continue;
            const unsigned long long m = cut(any & cmask, rso[i] + (uint32_t)__popcll(any & below));
            // codes: 5 = 101, 4 = 100, 3 = 011, 2 = 010, 1 = 001; q_k accumulates the hits ABOVE level k
            const uint32_t c5 = (uint32_t)__popcll(m & rb[i] & rp[i].x), c4 = (uint32_t)__popcll(m & rb[i] & ~rp[i].x);
            const uint32_t c3 = (uint32_t)__popcll(m & ~rb[i] & rp[i].y & rp[i].x), c2 = (uint32_t)__popcll(m & ~rb[i] & rp[i].y & ~rp[i].x);
            q4 += c5; q3 += c5 + c4; q2 += c5 + c4 + c3; q1 += c5 + c4 + c3 + c2;
        }
    };
    auto place = [&](int yc) {
#pragma unroll
        for (int i = 0; i < CH; i++) {
            const unsigned long long any = rp[i].x | rp[i].y | rb[i];
            if (!(any & cmask)) continue;
            uint32_t idx = rso[i] + (uint32_t)__popcll(any & below); // raster rank of the first hit here
            unsigned long long m = cut(any & cmask, idx);
            while (m) {
                const int l = __builtin_ctzll(m);
                m &= m - 1;
                const int code = (int)((rp[i].x >> l) & 1ull) | ((int)((rp[i].y >> l) & 1ull) << 1) | ((int)((rb[i] >> l) & 1ull) << 2);
                uint32_t at;
                if (MODE == 2) {
                    at = code == 5 ? q5 : (code == 4 ? q4 : (code == 3 ? q3 : (code == 2 ? q2 : q1)));
                    q5 += code == 5; q4 += code == 4; q3 += code == 3; q2 += code == 2; q1 += code == 1;
                } else {
                    at = q1++;
                }
                P.rec[at] = make_uint4(((uint32_t)(yc + i) << 16) | (uint32_t)(tx * 64 + l), (uint32_t)(code + 11), idx, ST_UNDECIDED);
                idx++;
            }
        }
    };
    if (y1 - y0 <= CH) { // 8-pixel cells: one chunk serves both passes
        load_chunk(y0);
        if (MODE == 2) count_levels();
        place(y0);
    } else {
        if (MODE == 2)
            for (int yc = y0; yc < y1; yc += CH) { load_chunk(yc); count_levels(); }
        for (int yc = y0; yc < y1; yc += CH) { load_chunk(yc); place(yc); }
    }
    if (MODE == 2) {
        uint2 ch = make_uint2(0u, 0u);
        if (q1 > first) { const uint4 r0 = P.rec[first]; ch = make_uint2(champ_key(r0.y, r0.z), r0.x); }
        P.champ[(cy + L.R) * L.cgw + cx + L.R] = ch;
    }
}

// The (2R+1)^2-cell neighbourhood of cell (cx, cy) as up to 7 runs of records (one per cell row), one flat
// index space.  All indexing below is compile-time after unrolling, so the arrays live in registers.
constexpr int MAXRUN = 7;
struct Runs { uint32_t q[MAXRUN], l[MAXRUN], total; };

// `cs` = the cell offsets: P.cell_start (bias 0) or an LDS copy of the slice [bias, ...) a workgroup needs
__device__ __forceinline__ Runs cell_runs_from(const uint32_t *cs, int bias, const NmsLayout &L, int cx, int cy)
{
    Runs R;
    const int cx0 = cx - L.R > 0 ? cx - L.R : 0, cx1 = cx + L.R < L.gw ? cx + L.R : L.gw - 1;
    R.total = 0;
#pragma unroll
    for (int j = 0; j < MAXRUN; j++) {
        const int yy = cy - L.R + j;
        uint32_t q = 0, len = 0;
        if (j <= 2 * L.R && yy >= 0 && yy < L.gh) {
            q = cs[yy * L.gw + cx0 - bias];
            len = cs[yy * L.gw + cx1 + 1 - bias] - q;
        }
        R.q[j] = q; R.l[j] = len;
        R.total += len;
    }
    return R;
}

__device__ __forceinline__ Runs cell_runs(const NmsPtrs &P, const NmsLayout &L, int cx, int cy)
{
    return cell_runs_from(P.cell_start, 0, L, cx, cy);
}

__device__ __forceinline__ uint32_t run_pos(const Runs &R, uint32_t fi)
{
    uint32_t pos = 0;
    bool done = false;
#pragma unroll
    for (int j = 0; j < MAXRUN; j++) {
        const bool in = !done && fi < R.l[j];
        pos = in ? R.q[j] + fi : pos;
        done = done || in;
        fi -= R.l[j]; // only meaningful while !done
    }
    return pos;
}

constexpr int NB_REG = 2; // neighbour records per lane and pass (128 neighbours)

// A lane's neighbour record in the champion and push rounds.  EVERY lane loads (lanes past the end of the block re-read its
// first record and get a state that takes no part in anything), and the callers evaluate what they loaded with bitwise
// arithmetic instead of short-circuit conditions: no load of mutable state under a per-lane exec mask and no divergent
// branch between such a load and its use -- the two ingredients of the predicated-load hazard of the mask rounds (DESIGN.md
// section 4), which these rounds share the third one with (other workgroups' state stores in flight).
// tests/test_nms_codegen.py checks the generated code for exactly that.
__device__ __forceinline__ uint4 load_rec_all_lanes(const uint4 *rec, const Runs &R, uint32_t fi, uint32_t &q)
{
    const bool in = fi < R.total;
    q = run_pos(R, in ? fi : 0u);
    uint4 v = rec[q];
    v.w = in ? v.w : (uint32_t)ST_SUPPRESSED;
    return v;
}

__device__ __forceinline__ unsigned long long sort_key(int score, uint32_t idx)
{
    const uint32_t inv = ~((uint32_t)score ^ 0x80000000u); // larger score -> smaller key
    return ((unsigned long long)inv << 32) | idx;
}

// Phase A, one wavefront per cell: accept the cell's undecided points that no better undecided
// point within r beats, and append them to this round's "new" list.
__global__ __launch_bounds__(256) void k_nms_phase_a(NmsLayout L, int radius, int round, unsigned char *ws_all,
                                                     size_t ws_stride)
{
    __shared__ uint4 c_rec[4][64];
    const int f = blockIdx.y, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int c = blockIdx.x * 4 + wv;
    if (c >= L.ncell) return; // wave-uniform
    NmsPtrs P = nms_ptrs(ws_all + (size_t)f * ws_stride, L);
    if (P.cell_und[c] == 0) return; // nothing left to decide here (also covers empty cells)
    const uint32_t p0 = P.cell_start[c], p1 = P.cell_start[c + 1];
    const long long r2 = (long long)radius * (long long)radius;
    const int cy = c / L.gw, cx = c - cy * L.gw;
    uint32_t und_seen = 0;
    bool runs_ready = false;
    Runs R;

    for (uint32_t pb = p0; pb < p1; pb += 64) { // the cell's own points, 64 at a time
        const uint32_t p = pb + lane;
        const bool have = p < p1;
        uint4 me = P.rec[have ? p : p0]; // every lane loads (see load_rec_all_lanes)
        me.w = have ? me.w : (uint32_t)ST_SUPPRESSED;
        const bool und = me.w == ST_UNDECIDED;
        unsigned long long alive = __ballot(und);
        if (!alive) continue;
        und_seen += (uint32_t)__popcll(alive);
        if (!runs_ready) { R = cell_runs(P, L, cx, cy); runs_ready = true; }
        // ---- own cell first: cheap and decides most points ----
        __builtin_amdgcn_wave_barrier();
        c_rec[wv][lane] = me;
        __builtin_amdgcn_wave_barrier();
        {
            unsigned long long todo = alive;
            while (todo) {
                const int k = __builtin_ctzll(todo);
                todo &= todo - 1;
                const uint4 ce = c_rec[wv][k];
                const bool h = und && lane != k && within(me.x, ce.x, r2) && better((int)me.y, me.z, (int)ce.y, ce.z);
                if (__any(h)) alive &= ~(1ull << k);
            }
        }
        // ---- then the whole neighbourhood, 256 records per pass, for the centres still standing ----
        for (uint32_t nb0 = 0; nb0 < R.total && alive; nb0 += 64 * NB_REG) {
            uint4 nb[NB_REG];
            uint32_t nq[NB_REG];
#pragma unroll
            for (int k = 0; k < NB_REG; k++) nb[k] = load_rec_all_lanes(P.rec, R, nb0 + k * 64 + lane, nq[k]);
            const int nch = (int)(((R.total - nb0 < (uint32_t)(64 * NB_REG) ? R.total - nb0 : (uint32_t)(64 * NB_REG)) + 63) / 64);
            unsigned long long todo = alive;
            while (todo) {
                const int k = __builtin_ctzll(todo);
                todo &= todo - 1;
                const uint4 ce = c_rec[wv][k];
                uint32_t h = 0;
#pragma unroll
                for (int j = 0; j < NB_REG; j++) {
                    if (j >= nch) break; // uniform
                    // a point accepted earlier in THIS phase (NEW) was undecided when the round began
                    const uint32_t bet = (uint32_t)((int)nb[j].y > (int)ce.y) | ((uint32_t)((int)nb[j].y == (int)ce.y) & (uint32_t)(nb[j].z < ce.z));
                    h |= ((uint32_t)(nb[j].w == ST_UNDECIDED) | (uint32_t)(nb[j].w == ST_NEW)) & (uint32_t)(nq[j] != pb + (uint32_t)k) &
                         (uint32_t)within(nb[j].x, ce.x, r2) & bet;
                }
                if (__any(h != 0)) alive &= ~(1ull << k);
            }
        }
        if (alive) { // nobody better within r: accept, and queue for this round's suppression pass
            const bool mine = (alive >> lane) & 1ull;
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&P.counters[round & 1], (uint32_t)__popcll(alive));
            base = (uint32_t)__shfl((int)base, 0);
            if (mine) {
                *rec_state(P.rec, p) = ST_NEW;
                P.listA[base + (uint32_t)__popcll(alive & ((1ull << lane) - 1ull))] = p;
            }
        }
    }
    if (und_seen == 0 && lane == 0) P.cell_und[c] = 0; // everything here was decided by last round's suppression pass
}

// ---- champion rounds (L.champ): cells are small enough that any two points of a cell are within r, so in
// every round only the best undecided point of a cell (its champion) can be locally best; every other
// undecided point of the cell is beaten by its champion without a single distance test.
//
// k_nms_champ     one lane per cell: advance the cell's cursor past decided records (records are in priority
//                 order) and publish {key, xy} of the champion in the padded champion grid;
// k_nms_phase_c   one lane per cell: compare the champion with the (2R+1)^2-1 neighbouring champions
//                 (coalesced 8-byte loads): a better champion within r beats it; no better champion at all
//                 in the block accepts it (any better undecided point q within r would make the champion of
//                 q's cell better still); only "better champion, but farther than r" needs the exact test
//                 over the block's records, done by the whole wavefront for one such cell at a time.
// The accepted set of a round is exactly the locally-best set of the plain formulation above.
__device__ __forceinline__ void champ_cell(const NmsPtrs &P, const NmsLayout &L, int c)
{
    const uint32_t und = P.cell_und[c]; // the three per-cell words in one round trip, not three
    uint32_t h = P.cell_fill[c];
    const uint32_t e = P.cell_start[c + 1];
    if (und == 0) return; // its grid entry is already {0,0}
    uint4 r = make_uint4(0, 0, 0, ST_SUPPRESSED);
    while (h < e) { // eight records per trip: the walk is a chain of dependent L2 round trips otherwise
        uint4 rr[8];
#pragma unroll
        for (int k = 0; k < 8; k++) rr[k] = h + k < e ? P.rec[h + k] : make_uint4(0, 0, 0, ST_SUPPRESSED);
        int hit = -1;
#pragma unroll
        for (int k = 7; k >= 0; k--) if (rr[k].w == ST_UNDECIDED) { hit = k; r = rr[k]; }
        if (hit >= 0) { h += (uint32_t)hit; break; }
        h += 8;
    }
    if (h > e) h = e;
    const int cy = c / L.gw, cx = c - cy * L.gw;
    P.cell_fill[c] = h;
    if (h < e) {
        P.champ[(cy + L.R) * L.cgw + cx + L.R] = make_uint2(champ_key(r.y, r.z), r.x);
    } else {
        P.champ[(cy + L.R) * L.cgw + cx + L.R] = make_uint2(0u, 0u);
        P.cell_und[c] = 0;
    }
}


__global__ __launch_bounds__(256) void k_nms_champ(NmsLayout L, unsigned char *ws_all, size_t ws_stride)
{
    const int f = blockIdx.y;
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= L.ncell) return;
    const NmsPtrs P = nms_ptrs(ws_all + (size_t)f * ws_stride, L);
    champ_cell(P, L, c);
}

// called by whole wavefronts: lane = one cell (c, valid when incell)
template <int RR>
__device__ __forceinline__ void phase_c_wave(const NmsPtrs &P, const NmsLayout &L, int radius, int round, int c, bool incell,
                                             const uint32_t *cs, int cs_bias /* cell offsets: P.cell_start, 0 or an LDS slice */,
                                             const RawOut &raw)
{
    const int lane = threadIdx.x & 63;
    const int cc = incell ? c : 0;
    const int cy = cc / L.gw, cx = cc - cy * L.gw;
    const uint2 *crow = P.champ + (size_t)(cy + RR) * L.cgw + cx + RR;
    const uint2 me = *crow;
    const bool live = incell && me.x != 0u;
    if (!__any(live)) return;
    const int r2 = radius * radius;

    bool beaten = false, far_better = false;
#pragma unroll
    for (int dy = -RR; dy <= RR; dy++) {
        uint2 o[2 * RR + 1];
#pragma unroll
        for (int dx = -RR; dx <= RR; dx++) o[dx + RR] = crow[dy * L.cgw + dx];
#pragma unroll
        for (int dx = -RR; dx <= RR; dx++) {
            if (dx == 0 && dy == 0) continue;
            const bool b = o[dx + RR].x > me.x;
            const bool w = within_near(o[dx + RR].y, me.y, r2);
            beaten = beaten || (b && w);
            far_better = far_better || (b && !w);
        }
    }
    bool accept = live && !beaten && !far_better;

    // One centre at a time over the whole wavefront (lanes = the records of its (2R+1)^2 block):
    //   - the undecided cases get the exact test;
    //   - every accepted centre suppresses its neighbours right away.  Doing that inside the phase is safe:
    //     a point within r of an accepted one can never be accepted itself, so retiring it early only stops
    //     it from holding up others; readers that still see it undecided are merely conservative.
    const unsigned long long deep = __ballot(live && !beaten && far_better);
    unsigned long long todo = deep | __ballot(accept);
    unsigned long long acc = 0;
    while (todo) {
        const int k = __builtin_ctzll(todo);
        todo &= todo - 1;
        const uint32_t ckey = (uint32_t)__shfl((int)me.x, k), cxy = (uint32_t)__shfl((int)me.y, k);
        const int kx = __shfl(cx, k), ky = __shfl(cy, k);
        const Runs R = cell_runs_from(cs, cs_bias, L, kx, ky);
        if ((deep >> k) & 1ull) {
            uint32_t h = 0;
            for (uint32_t nb0 = 0; nb0 < R.total; nb0 += 64 * NB_REG) {
                uint4 nb[NB_REG];
                uint32_t nq[NB_REG];
#pragma unroll
                for (int j = 0; j < NB_REG; j++) nb[j] = load_rec_all_lanes(P.rec, R, nb0 + j * 64 + lane, nq[j]);
#pragma unroll
                for (int j = 0; j < NB_REG; j++) {
                    // a point accepted earlier in THIS phase (stamped with this round) was undecided when the
                    // round began; the centre itself has key == ckey and drops out of the strict comparison
                    h |= ((uint32_t)(nb[j].w == ST_UNDECIDED) | (uint32_t)(nb[j].w == ST_ACC_ROUND + (uint32_t)round)) &
                         (uint32_t)(champ_key(nb[j].y, nb[j].z) > ckey) & (uint32_t)within_near(nb[j].x, cxy, r2);
                }
                if (__any(h != 0)) break;
            }
            if (__any(h != 0)) continue; // beaten: stays undecided
        }
        acc |= 1ull << k;
        for (uint32_t nb0 = 0; nb0 < R.total; nb0 += 64 * NB_REG) {
            uint4 nb[NB_REG];
            uint32_t nq[NB_REG];
#pragma unroll
            for (int j = 0; j < NB_REG; j++) nb[j] = load_rec_all_lanes(P.rec, R, nb0 + j * 64 + lane, nq[j]);
#pragma unroll
            for (int j = 0; j < NB_REG; j++) { // the centre itself: key == ckey, skipped
                const uint32_t kill = (uint32_t)(nb[j].w == ST_UNDECIDED) & (uint32_t)(champ_key(nb[j].y, nb[j].z) != ckey) &
                                      (uint32_t)within_near(nb[j].x, cxy, r2);
                if (kill) *rec_state(P.rec, nq[j]) = ST_SUPPRESSED;
            }
        }
    }

    if (acc) { // stamp the accepted champions and emit their sort keys
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&P.counters[2], (uint32_t)__popcll(acc));
        base = (uint32_t)__shfl((int)base, 0);
        if ((acc >> lane) & 1ull) {
            *rec_state(P.rec, P.cell_fill[c]) = ST_ACC_ROUND + (uint32_t)round; // the champion's record
            const int score = (int)(me.x >> 24) + 11;
            const uint32_t rank = 0xFFFFFFu - (me.x & 0xFFFFFFu);
            P.sortkeys[base + (uint32_t)__popcll(acc & ((1ull << lane) - 1ull))] = sort_key(score, rank);
            raw.xy[rank] = me.y;
            raw.score[rank] = score;
        }
    }
}

constexpr int CS_LDS_MAX = 4096; // cell offsets a phase-C workgroup may stage (16 KiB)

// STAGE (first two rounds only): the exact test and the suppression walk start with the run table of a centre's
// (2R+1)^2 block, ten cell offsets.  The 256 cells of a workgroup only ever need the offsets of their own grid
// rows +-R, a few KB: staged in LDS once, so that per centre the dependent global round trips are "records" only.
// Pays only while a workgroup has dozens of centres (rounds 0 and 1); later rounds have few and skip it.
template <int RR, bool STAGE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) // latency-bound: 8 waves/SIMD (<= 64 VGPRs) measured 6 % faster than 7
void k_nms_phase_c(NmsLayout L, int radius, int round, unsigned char *ws_all,
                                                     size_t ws_stride, uint32_t *raw_xy_all, int32_t *raw_score_all, int n_cap)
{
    __shared__ uint32_t cs_lds[STAGE ? CS_LDS_MAX : 1];
    const int f = blockIdx.y;
    const int c0 = blockIdx.x * 256;
    const int c = c0 + threadIdx.x;
    const NmsPtrs P = nms_ptrs(ws_all + (size_t)f * ws_stride, L);
    const RawOut raw{raw_xy_all + (size_t)f * n_cap, raw_score_all + (size_t)f * n_cap};
    if (STAGE) {
        const int clast = c0 + 255 < L.ncell ? c0 + 255 : L.ncell - 1;
        int ylo = c0 / L.gw - RR, yhi = clast / L.gw + RR;
        ylo = ylo < 0 ? 0 : ylo;
        yhi = yhi >= L.gh ? L.gh - 1 : yhi;
        const int lo = ylo * L.gw, n = (yhi + 1) * L.gw + 1 - lo; // flat slice [lo, lo + n) of cell_start
        if (n <= CS_LDS_MAX) {                                     // block-uniform
            for (int i = threadIdx.x; i < n; i += 256) cs_lds[i] = P.cell_start[lo + i];
            __syncthreads();
            phase_c_wave<RR>(P, L, radius, round, c, c < L.ncell, cs_lds, lo, raw); // separate call: the address space stays known
            return;
        }
    }
    phase_c_wave<RR>(P, L, radius, round, c, c < L.ncell, P.cell_start, 0, raw);
}

// Phase B, one wavefront per NEWLY accepted point (a few thousand per frame over all rounds): retire
// it to ACCEPTED, emit its sort key, and suppress every undecided point within r (lanes = neighbours).
__global__ __launch_bounds__(256) void k_nms_push(NmsLayout L, int radius, int round, unsigned char *ws_all,
                                                  size_t ws_stride)
{
    const int f = blockIdx.y, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    NmsPtrs P = nms_ptrs(ws_all + (size_t)f * ws_stride, L);
    const uint32_t nnew = P.counters[round & 1];
    if (blockIdx.x == 0 && threadIdx.x == 0) P.counters[(round + 1) & 1] = 0; // next round's list starts empty
    const long long r2 = (long long)radius * (long long)radius;
    for (uint32_t a = blockIdx.x * 4 + wv; a < nnew; a += gridDim.x * 4) {
        const uint32_t p = P.listA[a];
        const uint4 me = P.rec[p];
        const int cx = (int)(me.x & 0xFFFFu) / L.cs, cy = (int)(me.x >> 16) / L.cs;
        const Runs R = cell_runs(P, L, cx, cy);
        for (uint32_t nb0 = 0; nb0 < R.total; nb0 += 64 * NB_REG) {
            uint4 nb[NB_REG];
            uint32_t nq[NB_REG];
#pragma unroll
            for (int k = 0; k < NB_REG; k++) nb[k] = load_rec_all_lanes(P.rec, R, nb0 + k * 64 + lane, nq[k]);
#pragma unroll
            for (int k = 0; k < NB_REG; k++) {
                const uint32_t kill = (uint32_t)(nb[k].w == ST_UNDECIDED) & (uint32_t)within(nb[k].x, me.x, r2);
                if (kill) *rec_state(P.rec, nq[k]) = ST_SUPPRESSED;
            }
        }
        if (lane == 0) {
            *rec_state(P.rec, p) = ST_ACCEPTED;
            P.sortkeys[atomicAdd(&P.counters[2], 1u)] = sort_key((int)me.y, me.z);
        }
    }
}

// ---- mask rounds (L.mask): champion rounds for 8-pixel cells (r = 10..21, the default r = 16) without per-hit records.
// An 8x8 cell IS one 64-bit word (bit = 8 * row + column): per padded cell the workspace holds
//     alive   the hits still undecided (plus the accepted one, if any: it stays visible to its neighbours)
//     planes  the three FAST score-bit planes of the cell (read-only after setup)
//     ent     the cell's champion as a 32-bit key (score level, then raster position), bit 31 = "accepted"
// and per radius one table disk[offset][position] = the pixels of the neighbouring cell at `offset` within r of in-cell
// position `position`.  Everything a round needs is then register arithmetic on coalesced 8-byte loads:
//     champion of a cell      = lowest set bit of (alive & highest non-empty score level)
//     "better than me" there  = (levels above mine) | (my level & raster-earlier pixels)
//     blockers of a champion  = alive[n] & disk & better   over the neighbouring cells n whose champion is better
//     suppression             = atomicAnd(alive[n], ~disk)  (no return value: fire and forget)
// The champion grid is only a FILTER (an entry is never below the cell's true champion, so a cell whose entry is not
// better than me holds nothing better); exactness comes from the masks.  Entries may therefore be stale, and one
// launch per round does both the refresh of a cell's own entry and the test.  A bit cleared while a round is running
// is a suppressed point, which can never block anybody, so reading alive words mid-round is safe in both directions.
// No sort of the hits into cells, no records: setup is one pass over the FAST planes.
constexpr uint32_t MK_FLAG = 0x80000000u, MK_KEY = 0x7FFFFFFFu;

struct MaskPtrs {
    unsigned long long *alive, *pl0, *pl1, *pl2;
    uint32_t *ent, *counters, *listA, *listB;
    unsigned long long *sortkeys;
};

__device__ __forceinline__ MaskPtrs mask_ptrs(unsigned char *ws, const NmsLayout &L)
{
    MaskPtrs M;
    const size_t npc = (size_t)L.cgw * (L.gh + 2 * L.R);
    M.alive = reinterpret_cast<unsigned long long *>(ws + L.off_alive);
    M.pl0 = reinterpret_cast<unsigned long long *>(ws + L.off_planes);
    M.pl1 = M.pl0 + npc;
    M.pl2 = M.pl1 + npc;
    M.ent = reinterpret_cast<uint32_t *>(ws + L.off_champ);
    M.counters = reinterpret_cast<uint32_t *>(ws + L.off_counters);
    M.listA = reinterpret_cast<uint32_t *>(ws + L.off_listA);
    M.listB = reinterpret_cast<uint32_t *>(ws + L.off_listB);
    M.sortkeys = reinterpret_cast<unsigned long long *>(ws + L.off_sortkeys);
    return M;
}

// key: larger = better; score code 1..5 (score - 11), then raster position ascending (== raster rank ascending)
__device__ __forceinline__ uint32_t mask_key(int code, int x, int y) { return ((uint32_t)code << 28) | (0x0FFFFFFFu - (((uint32_t)y << 14) | (uint32_t)x)); }

__device__ __forceinline__ uint32_t mask_champ(unsigned long long a, unsigned long long A0, unsigned long long A1, unsigned long long A2,
                                               int cx, int cy)
{
    if (!a) return 0u;
    // codes (planes A2 A1 A0): 5 = 101, 4 = 100, 3 = 011, 2 = 010, 1 = 001
    unsigned long long m = a & A2 & A0;
    int code = 5;
    if (!m) { m = a & A2; code = 4; }
    if (!m) { m = a & A1 & A0; code = 3; }
    if (!m) { m = a & A1; code = 2; }
    if (!m) { m = a; code = 1; }
    const int b = __builtin_ctzll(m);
    return mask_key(code, cx * 8 + (b & 7), cy * 8 + (b >> 3));
}

// alive words are changed by L2 atomics; the tail kernel reads them again after a block barrier, so the load goes to L2
__device__ __forceinline__ unsigned long long load_alive(const unsigned long long *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int RR>
__global__ __launch_bounds__(256) void k_nmsm_setup(const unsigned long long *__restrict__ seg_all,
                                                    const uint32_t *__restrict__ segoff_all, const int32_t *__restrict__ n_raw_all,
                                                    int W, int H, int ntx, int n_cap, int radius, NmsLayout L,
                                                    unsigned char *ws_all, size_t ws_stride)
{
    constexpr int ND = 2 * RR + 1;
    const int f = blockIdx.y;
    const int q = blockIdx.x * 256 + threadIdx.x;
    const int cgh = L.gh + 2 * RR, npc = L.cgw * cgh;
    if (f == 0) { // the disk table of this radius (frame 0's copy serves every frame)
        unsigned long long *disk = reinterpret_cast<unsigned long long *>(ws_all + L.off_disk);
        const int r2 = radius * radius;
        for (int t = q; t < ND * ND * 64; t += gridDim.x * 256) {
            const int o = t >> 6, pos = t & 63, ux = pos & 7, uy = pos >> 3;
            const int dx = o % ND - RR, dy = o / ND - RR;
            unsigned long long m = 0;
            for (int i = 0; i < 8; i++) {
                const int ddy = dy * 8 + i - uy, rem = r2 - ddy * ddy;
                if (rem < 0) continue;
                int hw = (int)sqrtf((float)rem);
                while ((hw + 1) * (hw + 1) <= rem) hw++;
                while (hw * hw > rem) hw--;
                int lo = ux - dx * 8 - hw, hi = ux - dx * 8 + hw; // columns j of the neighbouring cell with |dx*8 + j - ux| <= hw
                lo = lo < 0 ? 0 : lo;
                hi = hi > 7 ? 7 : hi;
                if (lo <= hi) m |= (unsigned long long)(((1u << (hi + 1)) - 1u) & ~((1u << lo) - 1u)) << (8 * i);
            }
            disk[t] = m;
        }
    }
    if (q >= npc) return;
    const MaskPtrs M = mask_ptrs(ws_all + (size_t)f * ws_stride, L);
    if (q < 16) M.counters[q] = 0;
    const int py = q / L.cgw, px = q - py * L.cgw;
    const int cx = px - RR, cy = py - RR;
    if (cx < 0 || cx >= L.gw || cy < 0 || cy >= L.gh) { // border: never alive, never a champion
        M.alive[q] = 0ull;
        M.ent[q] = 0u;
        return;
    }
    const int x0 = cx * 8, tx = x0 >> 6, bo = x0 & 63;
    const int y0 = cy * 8, y1 = (y0 + 8 < H) ? y0 + 8 : H;
    const size_t nseg = (size_t)H * ntx;
    const unsigned long long *seg = seg_all + (size_t)f * nseg * 4;
    const bool over = n_raw_all[f] > n_cap; // hits past the raw capacity are dropped everywhere (PGX_E_CAPACITY is raised)
    unsigned long long A0 = 0, A1 = 0, A2 = 0;
    ulonglong2 rp8[8];
    unsigned long long rb8[8];
#pragma unroll
    for (int i = 0; i < 8; i++) { // all loads first
        const bool ok = y0 + i < y1;
        const size_t si = (size_t)(ok ? y0 + i : y0) * ntx + tx;
        rp8[i] = *reinterpret_cast<const ulonglong2 *>(seg + si * 4);
        rb8[i] = seg[si * 4 + 2];
        if (!ok) { rp8[i] = make_ulonglong2(0ull, 0ull); rb8[i] = 0ull; }
    }
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const unsigned long long any = rp8[i].x | rp8[i].y | rb8[i];
        uint32_t keep = (uint32_t)(any >> bo) & 0xFFu;
        if (over && keep) { // keep the first n_cap - base hits of the row slice (raster ranks below n_cap)
            const uint32_t base = segoff_all[(size_t)f * nseg + (size_t)(y0 + i) * ntx + tx] + (uint32_t)__popcll(any & ((1ull << bo) - 1ull));
            const uint32_t room = base < (uint32_t)n_cap ? (uint32_t)n_cap - base : 0u;
            while ((uint32_t)__popc(keep) > room) keep &= ~(1u << (31 - __builtin_clz(keep)));
        }
        A0 |= ((rp8[i].x >> bo) & (unsigned long long)keep) << (8 * i);
        A1 |= ((rp8[i].y >> bo) & (unsigned long long)keep) << (8 * i);
        A2 |= ((rb8[i] >> bo) & (unsigned long long)keep) << (8 * i);
    }
    const unsigned long long a = A0 | A1 | A2;
    M.alive[q] = a;
    M.pl0[q] = A0; M.pl1[q] = A1; M.pl2[q] = A2;
    M.ent[q] = mask_champ(a, A0, A1, A2, cx, cy);
}

struct MaskOut { uint32_t *raw_xy; int32_t *raw_score; const unsigned long long *seg; const uint32_t *segoff; int ntx; };

// One round for 64 cells (lane = cell c, valid when incell); called by whole wavefronts.
__device__ __forceinline__ uint32_t ent_load(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void ent_store(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// ---- pieces of one round, shared by the whole-chip kernel (neighbours staged in LDS) and the tail (neighbours from global)

// own cell at the start of a round: close it if its champion was accepted, take a stale entry down if it ran empty,
// move on to the next champion if the current one was suppressed.  Returns true when the cell has a champion to test.
__device__ __forceinline__ bool mask_own(const MaskPtrs &M, int pc, int cx, int cy, bool incell, uint32_t &me, unsigned long long a,
                                         unsigned long long q0, unsigned long long q1, unsigned long long q2)
{
    if (!incell) return false;
    if (me & MK_FLAG) { // accepted in an earlier round: everything else in the cell is suppressed; close it
        ent_store(M.ent + pc, 0u);
        // through the atomic path like every other change of an alive word
        __hip_atomic_store(M.alive + pc, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return false;
    }
    if (a == 0ull) {
        if (me) ent_store(M.ent + pc, 0u);
        return false;
    }
    const uint32_t pos = 0x0FFFFFFFu - (me & 0x0FFFFFFFu);
    const int b = (int)((((pos >> 14) & 7u) << 3) | (pos & 7u));
    if (me == 0u || !((a >> b) & 1ull)) { // the champion was suppressed: next one
        me = mask_champ(a, q0, q1, q2, cx, cy);
        ent_store(M.ent + pc, me);
    }
    return true;
}

// per-lane constants of a champion under test
struct MaskMe {
    uint32_t key;
    int mx, my, ux, uy, code;
    unsigned long long rows_below, row_mine, s5, s4, s3, s2, s1;
};

__device__ __forceinline__ MaskMe mask_me(uint32_t me)
{
    MaskMe m;
    m.key = me;
    const uint32_t mypos = 0x0FFFFFFFu - (me & 0x0FFFFFFFu);
    m.mx = (int)(mypos & 0x3FFFu); m.my = (int)(mypos >> 14);
    m.ux = m.mx & 7; m.uy = m.my & 7; m.code = (int)(me >> 28);
    m.rows_below = (1ull << (8 * m.uy)) - 1ull;
    m.row_mine = 0xFFull << (8 * m.uy);
    // "better than me" = score level above mine, or my level and raster-earlier.  With ge_t = pixels of level >= t
    // (ge5 = p2 & p0, ge4 = p2, ge3 = p2 | (p1 & p0), ge2 = p2 | p1, ge1 = every hit) that is ge(code + 1) | (ge(code) & E);
    // the level is picked with per-lane all-ones masks, so a step is straight-line code for the whole wavefront.
    m.s5 = m.code == 5 ? ~0ull : 0ull; m.s4 = m.code == 4 ? ~0ull : 0ull; m.s3 = m.code == 3 ? ~0ull : 0ull;
    m.s2 = m.code == 2 ? ~0ull : 0ull; m.s1 = m.code == 1 ? ~0ull : 0ull;
    return m;
}

// the filter: one bit per neighbouring cell whose champion is better than mine and whose nearest pixel is within r
template <int RR>
__device__ __forceinline__ unsigned long long mask_need(const MaskMe &m, const uint32_t *e /*[NO]*/, int r2)
{
    constexpr int ND = 2 * RR + 1, NO = ND * ND;
    unsigned long long needm = 0ull;
#pragma unroll
    for (int o = 0; o < NO; o++) {
        const int dx = o % ND - RR, dy = o / ND - RR;
        if (dx == 0 && dy == 0) continue;
        const int ndx = dx > 0 ? dx * 8 - m.ux : (dx < 0 ? m.ux - (dx * 8 + 7) : 0);
        const int ndy = dy > 0 ? dy * 8 - m.uy : (dy < 0 ? m.uy - (dy * 8 + 7) : 0);
        if ((e[o] & MK_KEY) > m.key && ndx * ndx + ndy * ndy <= r2) needm |= 1ull << o;
    }
    return needm;
}

// blockers in one neighbouring cell (dx, dy compile-time or uniform): alive & within r & better than me
__device__ __forceinline__ unsigned long long mask_blockers(const MaskMe &m, int dx, int dy, unsigned long long an, unsigned long long dm,
                                                            unsigned long long p0, unsigned long long p1, unsigned long long p2)
{
    // raster-earlier pixels of that cell (same score level: the earlier one wins)
    const unsigned long long E = dy < 0 ? ~0ull : (dy > 0 ? 0ull : (m.rows_below | (dx < 0 ? m.row_mine : 0ull)));
    const unsigned long long ge5 = p2 & p0, ge4 = p2, ge3 = p2 | (p1 & p0), ge2 = p2 | p1;
    const unsigned long long above = (ge5 & m.s4) | (ge4 & m.s3) | (ge3 & m.s2) | (ge2 & m.s1);
    const unsigned long long mine = (ge5 & m.s5) | (ge4 & m.s4) | (ge3 & m.s3) | (ge2 & m.s2) | m.s1;
    return an & dm & (above | (mine & E));
}

// accept the champions of the lanes with `accept`: stamp, suppress everything within r, emit the output records.
// Called by whole wavefronts.
// `snap` (optional): alive words staged earlier in this launch, snap[k + j * snap_w] = the cell at offset (k - RR, j - RR)
// from the accepting cell.  Bits only ever clear, so a staged word is a superset of the current one and a neighbour whose
// staged word has nothing inside the disk needs no atomic at all (two of three neighbouring cells on the bench frames).
template <int RR>
__device__ __forceinline__ void mask_accept(const MaskPtrs &M, const NmsLayout &L, const unsigned long long *dk, int pc, const MaskMe &m,
                                            bool accept, const MaskOut &out, const unsigned long long *snap = nullptr, int snap_w = 0)
{
    constexpr int ND = 2 * RR + 1;
    const int lane = threadIdx.x & 63;
    const unsigned long long acc = __ballot(accept);
    if (!acc) return;
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(&M.counters[2], (uint32_t)__popcll(acc));
    base = (uint32_t)__shfl((int)base, 0);
    if (accept) {
        ent_store(M.ent + pc, m.key | MK_FLAG);
        atomicAnd(M.alive + pc, 1ull << (m.uy * 8 + m.ux)); // any two points of a cell are within r
#pragma unroll 1
        for (int j = 0; j < ND; j++) {
            const uint32_t prow = (uint32_t)(pc + (j - RR) * L.cgw - RR);
            unsigned long long d[ND];
#pragma unroll
            for (int k = 0; k < ND; k++) d[k] = dk[(j * ND + k) * 64];
            if (snap) {
#pragma unroll
                for (int k = 0; k < ND; k++) d[k] &= snap[j * snap_w + k];
            }
#pragma unroll
            for (int k = 0; k < ND; k++)
                if (d[k] && !(j == RR && k == RR)) atomicAnd(M.alive + prow + k, ~d[k]);
        }
        // the point's raster rank (its index in the raw lists): offset of its 64-pixel row segment + hits before it there
        const size_t si = (size_t)m.my * out.ntx + (m.mx >> 6);
        const ulonglong2 s01 = *reinterpret_cast<const ulonglong2 *>(out.seg + si * 4);
        const unsigned long long s2 = out.seg[si * 4 + 2];
        const uint32_t rank = out.segoff[si] + (uint32_t)__popcll((s01.x | s01.y | s2) & ((1ull << (m.mx & 63)) - 1ull));
        const int score = m.code + 11;
        M.sortkeys[base + (uint32_t)__popcll(acc & ((1ull << lane) - 1ull))] = sort_key(score, rank);
        out.raw_xy[rank] = ((uint32_t)m.my << 16) | (uint32_t)m.mx;
        out.raw_score[rank] = score;
    }
}

// One round for 64 cells (lane = cell c, valid when incell), neighbours read from global memory; called by whole
// wavefronts (the tail kernel's compact lists).
template <int RR>
__device__ __forceinline__ void mask_round_wave(const MaskPtrs &M, const NmsLayout &L, const unsigned long long *__restrict__ disk,
                                                int radius, int c, bool incell, const MaskOut &out)
{
    constexpr int ND = 2 * RR + 1, NO = ND * ND;
    const int cc = incell ? c : 0;
    const int cy = cc / L.gw, cx = cc - cy * L.gw;
    const int pc = (cy + RR) * L.cgw + cx + RR;
    uint32_t me = incell ? ent_load(M.ent + pc) : 0u;
    const unsigned long long a = incell ? load_alive(M.alive + pc) : 0ull;
    const bool live = mask_own(M, pc, cx, cy, incell, me, a, M.pl0[pc], M.pl1[pc], M.pl2[pc]);
    if (!__any(live)) return;
    const MaskMe m = mask_me(me);
    const int r2 = radius * radius;
    const unsigned long long *dk = disk + (m.uy * 8 + m.ux);

    unsigned long long needm;
    {
        uint32_t e[NO]; // all neighbouring champions, loads issued together
#pragma unroll
        for (int o = 0; o < NO; o++) {
            const int dx = o % ND - RR, dy = o / ND - RR;
            e[o] = (dx == 0 && dy == 0) ? 0u : M.ent[pc + dy * L.cgw + dx]; // plain: a stale entry is an older, higher one
        }
        needm = live ? mask_need<RR>(m, e, r2) : 0ull;
    }
    bool blocked = false;
#pragma unroll 1
    for (int j = 0; j < ND; j++) { // one row of neighbouring cells per step: the loads of a step are issued together
        const int dy = j - RR, g0 = j * ND;
        const unsigned long long gm = blocked ? 0ull : (needm >> g0) & ((1ull << ND) - 1ull);
        if (!__any(gm != 0ull)) continue; // nobody in the wavefront needs this row
        // Every lane loads the whole row and the need bits are applied afterwards: loads under per-lane predicates
        // (divergent skips of single cells) gave rare wrong "not blocked" results on gfx950 that were never explained
        // (DESIGN.md section 4); wave-uniform control flow around full-exec loads has not shown them.
        const uint32_t prow = (uint32_t)(pc + dy * L.cgw - RR);
        const unsigned long long *dkr = dk + (size_t)g0 * 64;
        unsigned long long an[ND], dm[ND], p0[ND], p1[ND], p2[ND];
#pragma unroll
        for (int k = 0; k < ND; k++) {
            an[k] = load_alive(M.alive + prow + k);
            dm[k] = dkr[k * 64];
            p0[k] = M.pl0[prow + k]; p1[k] = M.pl1[prow + k]; p2[k] = M.pl2[prow + k];
        }
        unsigned long long hit = 0ull;
#pragma unroll
        for (int k = 0; k < ND; k++)
            hit |= mask_blockers(m, k - RR, dy, an[k], dm[k], p0[k], p1[k], p2[k]) & (((gm >> k) & 1ull) ? ~0ull : 0ull);
        blocked = blocked || hit != 0ull;
    }
    mask_accept<RR>(M, L, dk, pc, m, live && !blocked, out);
}

// Whole-chip round: one 256-lane workgroup per tile of 32 x 8 cells.  The tile and its halo of R cells (alive word,
// planes, champion entry: 36 bytes per cell) are staged in LDS once, so the 24 or 48 neighbour reads of a lane are LDS
// reads instead of L1 requests (the per-lane global version was bound by the L1 request rate: 184 us for round 0 of 64
// frames).  The staged words are a snapshot from the start of the launch: stale alive bits are set bits, which only make
// a lane wait (safe), and own-cell state is read before anything in this launch can have changed what it means.
constexpr int MT_W = 32, MT_H = 8;

template <int RR>
__global__ __launch_bounds__(256) void k_nmsm_round(NmsLayout L, int radius, unsigned char *ws_all, size_t ws_stride,
                                                    uint32_t *raw_xy_all, int32_t *raw_score_all, int n_cap,
                                                    const unsigned long long *__restrict__ seg_all,
                                                    const uint32_t *__restrict__ segoff_all, int H, int ntx, int tiles_x)
{
    constexpr int ND = 2 * RR + 1, NO = ND * ND, SW = MT_W + 2 * RR, SH = MT_H + 2 * RR, NS = SW * SH;
    __shared__ unsigned long long s_a[NS], s_p0[NS], s_p1[NS], s_p2[NS];
    __shared__ uint32_t s_e[NS];
    __shared__ uint2 s_list[256]; // {staged index, champion key} of the tile's cells under test
    __shared__ uint32_t s_nlive;
    const int f = blockIdx.y;
    if (threadIdx.x == 0) s_nlive = 0;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const MaskPtrs M = mask_ptrs(ws_all + (size_t)f * ws_stride, L);
    const size_t nseg = (size_t)H * ntx;
    const MaskOut out{raw_xy_all + (size_t)f * n_cap, raw_score_all + (size_t)f * n_cap, seg_all + (size_t)f * nseg * 4,
                      segoff_all + (size_t)f * nseg, ntx};
    const unsigned long long *disk = reinterpret_cast<const unsigned long long *>(ws_all + L.off_disk);
    const int cgh = L.gh + 2 * RR;
    for (int i = threadIdx.x; i < NS; i += 256) { // staged cell (sx, sy) = padded cell (tx * 32 + sx, ty * 8 + sy)
        const int sy = i / SW, sx = i - sy * SW;
        const int px = tx * MT_W + sx, py = ty * MT_H + sy;
        const bool ok = px < L.cgw && py < cgh;
        const int q = ok ? py * L.cgw + px : 0;
        const unsigned long long a = ok ? load_alive(M.alive + q) : 0ull;
        const uint32_t e = ok ? M.ent[q] : 0u;
        const bool pl = a != 0ull; // closed cells need no planes (most cells after the first rounds)
        s_a[i] = a;
        s_e[i] = e;
        s_p0[i] = pl ? M.pl0[q] : 0ull;
        s_p1[i] = pl ? M.pl1[q] : 0ull;
        s_p2[i] = pl ? M.pl2[q] : 0ull;
    }
    __syncthreads();
    // own cells first; the cells that have a champion to test are then packed into a list, so that a tile with a few open
    // cells costs one wavefront's worth of vector work instead of four (after two rounds most cells are closed)
    {
        const int ly = threadIdx.x / MT_W, lx = threadIdx.x - ly * MT_W;
        const int cx = tx * MT_W + lx, cy = ty * MT_H + ly;
        const bool incell = cx < L.gw && cy < L.gh;
        const int si = (ly + RR) * SW + lx + RR;
        const int pc = (cy + RR) * L.cgw + cx + RR; // lanes outside the frame never use it
        uint32_t me = s_e[si];
        const bool live = mask_own(M, pc, cx, cy, incell, me, s_a[si], s_p0[si], s_p1[si], s_p2[si]);
        const unsigned long long lm = __ballot(live);
        uint32_t base = 0;
        if ((threadIdx.x & 63) == 0 && lm) base = atomicAdd(&s_nlive, (uint32_t)__popcll(lm));
        base = (uint32_t)__shfl((int)base, 0);
        if (live) {
            const uint32_t at = base + (uint32_t)__popcll(lm & ((1ull << (threadIdx.x & 63)) - 1ull));
            s_list[at] = make_uint2((uint32_t)si, me);
        }
    }
    __syncthreads();
    const int n_live = (int)s_nlive;
    const int r2 = radius * radius;
    for (int i0 = (threadIdx.x >> 6) * 64; i0 < n_live; i0 += 256) { // wave-uniform; no barrier below
        const int i = i0 + (threadIdx.x & 63);
        const bool live = i < n_live;
        const uint2 le = live ? s_list[i] : make_uint2((uint32_t)(RR * SW + RR), 0u);
        const int si = (int)le.x;
        const int sy = si / SW, sx = si - sy * SW;
        const int pc = (ty * MT_H + sy) * L.cgw + tx * MT_W + sx;
        const MaskMe m = mask_me(le.y);
        const unsigned long long *dk = disk + (m.uy * 8 + m.ux);
        unsigned long long needm;
        {
            uint32_t e[NO];
#pragma unroll
            for (int o = 0; o < NO; o++) {
                const int dx = o % ND - RR, dy = o / ND - RR;
                e[o] = s_e[si + dy * SW + dx];
            }
            needm = live ? mask_need<RR>(m, e, r2) : 0ull;
        }
        bool blocked = false;
#pragma unroll 1
        for (int j = 0; j < ND; j++) {
            const int dy = j - RR, g0 = j * ND;
            const unsigned long long gm = blocked ? 0ull : (needm >> g0) & ((1ull << ND) - 1ull);
            if (!__any(gm != 0ull)) continue; // nobody in the wavefront needs this row (see mask_round_wave on per-lane skips)
            const int srow = si + dy * SW - RR;
            const unsigned long long *dkr = dk + (size_t)g0 * 64;
            unsigned long long dm[ND];
#pragma unroll
            for (int k = 0; k < ND; k++) dm[k] = dkr[k * 64];
            unsigned long long hit = 0ull;
#pragma unroll
            for (int k = 0; k < ND; k++)
                hit |= mask_blockers(m, k - RR, dy, s_a[srow + k], dm[k], s_p0[srow + k], s_p1[srow + k], s_p2[srow + k]) &
                       (((gm >> k) & 1ull) ? ~0ull : 0ull);
            blocked = blocked || hit != 0ull;
        }
        mask_accept<RR>(M, L, dk, pc, m, live && !blocked, out, s_a + si - RR * SW - RR, SW);
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

// serial-per-thread neighbour walks for the few leftovers of the tail
__device__ __forceinline__ bool tail_beaten(const NmsPtrs &P, const NmsLayout &L, uint32_t p, long long r2)
{
    const uint4 me = P.rec[p];
    const int cx = (int)(me.x & 0xFFFFu) / L.cs, cy = (int)(me.x >> 16) / L.cs;
    const Runs R = cell_runs(P, L, cx, cy);
    for (uint32_t fi = 0; fi < R.total; fi++) {
        const uint32_t q = run_pos(R, fi);
        if (q == p) continue;
        const uint4 o = P.rec[q];
        if (o.w != ST_UNDECIDED && o.w != ST_NEW) continue;
        if (within(o.x, me.x, r2) && better((int)o.y, o.z, (int)me.y, me.z)) return true;
    }
    return false;
}

__device__ __forceinline__ bool tail_suppressed(const NmsPtrs &P, const NmsLayout &L, uint32_t p, long long r2)
{
    const uint4 me = P.rec[p];
    const int cx = (int)(me.x & 0xFFFFu) / L.cs, cy = (int)(me.x >> 16) / L.cs;
    const Runs R = cell_runs(P, L, cx, cy);
    for (uint32_t fi = 0; fi < R.total; fi++) {
        const uint4 o = P.rec[run_pos(R, fi)];
        if ((o.w == ST_NEW || o.w >= ST_ACCEPTED) && within(o.x, me.x, r2)) return true;
    }
    return false;
}

// Output order of a frame's accepted points (their keys were appended to P.sortkeys as they were accepted); called by the
// whole workgroup of a tail kernel (any multiple of 64 threads up to 1024); lds_cap = u64 keys the dynamic LDS holds.
__device__ __forceinline__ void tail_order(const NmsPtrs &P, const NmsLayout &L, int radius, int n, const int32_t *raw_score,
                                           uint32_t *order, int32_t *n_kept_all, int f, int kp_cap, int kp_soft, int *status,
                                           unsigned long long *lds_keys, uint32_t lds_cap, uint32_t *wsum /*[16]*/, uint32_t *sh_cnt_p,
                                           int *sh_max_p)
{
    const int tid = threadIdx.x;
    const int NT = (int)blockDim.x; // shadows the file's 1024: the mask-round tail runs with fewer threads
    uint32_t &sh_cnt = *sh_cnt_p;
    int &sh_max = *sh_max_p;
    // ---- the accepted points' keys (score descending, input index ascending) were appended as they were
    //      accepted; only the degenerate r < 0 case ("distance > r" is always true) gathers everything here
    __syncthreads();
    if (radius < 0) {
        if (tid == 0) sh_cnt = 0;
        __syncthreads();
        for (int p = tid; p < n; p += NT) P.sortkeys[atomicAdd(&sh_cnt, 1u)] = sort_key(raw_score[p], (uint32_t)p);
        __syncthreads();
    } else {
        if (tid == 0) sh_cnt = P.counters[2];
        __syncthreads();
    }
    const uint32_t nacc = sh_cnt;
    const uint32_t n2p = [](uint32_t v) { uint32_t p = 1; while (p < v) p <<= 1; return p; }(nacc > 1 ? nacc : 1);
    const uint32_t nw = ((uint32_t)n + 63u) / 64u * 2u;        // 32-bit bitmap words for the raster ranks of one score level
    if (L.champ && radius >= 0 && nw <= lds_cap * 2) {
        // FAST scores are 12..16 and the input index is the raster rank, so the output order is "level by
        // level, rank ascending": one bitmap of ranks per level in LDS, a popcount scan, and every set bit
        // knows its place.  No comparison sort.
        uint32_t *bm = reinterpret_cast<uint32_t *>(lds_keys);
        const uint32_t lp = (lds_cap * 2) / nw < 5u ? (lds_cap * 2) / nw : 5u; // levels per pass
        const int lane = tid & 63, wv = tid >> 6;
        uint32_t base = 0;
        for (int lev_hi = 16; lev_hi >= 12; lev_hi -= (int)lp) {
            const uint32_t tw = lp * nw;
            __syncthreads();
            for (uint32_t i = tid; i < tw; i += NT) bm[i] = 0u;
            __syncthreads();
            for (uint32_t i = tid; i < nacc; i += NT) {
                const unsigned long long k = P.sortkeys[i];
                const int sc = (int)(~(uint32_t)(k >> 32) ^ 0x80000000u);
                const uint32_t idx = (uint32_t)k;
                const int slot = lev_hi - sc;
                if (slot >= 0 && slot < (int)lp && sc >= 12) atomicOr(&bm[(uint32_t)slot * nw + (idx >> 5)], 1u << (idx & 31u));
            }
            __syncthreads();
            // each wavefront owns a contiguous span of words (rows of 64, lanes on consecutive words)
            const uint32_t rows = (tw + 63u) / 64u, rpw = (rows + NT / 64 - 1) / (NT / 64);
            const uint32_t r0 = (uint32_t)wv * rpw, r1 = r0 + rpw < rows ? r0 + rpw : rows;
            uint32_t cnt = 0;
            for (uint32_t r = r0; r < r1; r++) { const uint32_t w = r * 64u + lane; cnt += w < tw ? (uint32_t)__popc(bm[w]) : 0u; }
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) cnt += (uint32_t)__shfl_xor((int)cnt, d);
            if (lane == 0) wsum[wv] = cnt;
            __syncthreads();
            uint32_t run = base, tot = 0;
            for (int w = 0; w < NT / 64; w++) { const uint32_t v = wsum[w]; if (w < wv) run += v; tot += v; }
            for (uint32_t r = r0; r < r1; r++) {
                const uint32_t w = r * 64u + lane;
                uint32_t bits = w < tw ? bm[w] : 0u;
                const uint32_t c = (uint32_t)__popc(bits);
                uint32_t incl = c;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)incl, d); if (lane >= d) incl += t; }
                uint32_t o = run + incl - c;
                const uint32_t wl = w % nw; // word inside its level
                while (bits) {
                    const int b = __builtin_ctz(bits);
                    bits &= bits - 1;
                    if (o < (uint32_t)kp_cap) order[o] = wl * 32u + (uint32_t)b;
                    o++;
                }
                run += (uint32_t)__shfl((int)incl, 63);
            }
            base += tot;
        }
    } else if (n2p <= lds_cap) {
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
        if (nacc > (uint32_t)kp_cap && !kp_soft) atomicOr(status, (int)PGX_ST_KP_CAP); // a soft limit cuts the list silently
    }
}

__global__ __launch_bounds__(NT) void k_nms_tail(const int32_t *raw_score_all,
                                                 const int32_t *__restrict__ n_raw_all, int n_cap, NmsLayout L,
                                                 int radius, unsigned char *ws_all, size_t ws_stride,
                                                 uint32_t *__restrict__ order_all, int32_t *__restrict__ n_kept_all,
                                                 int kp_cap, int *status, int round0, uint32_t *raw_xy_out,
                                                 int32_t *raw_score_out, int kp_soft)
{
    extern __shared__ unsigned long long lds_keys[];
    __shared__ uint32_t wsum[NT / 64];
    __shared__ uint32_t sh_cnt;
    __shared__ int sh_max;
    const int f = blockIdx.x, tid = threadIdx.x;
    const int n = clamp_n(n_raw_all, f, n_cap);
    const int32_t *raw_score = raw_score_all + (size_t)f * n_cap;
    uint32_t *order = order_all + (size_t)f * kp_cap;
    NmsPtrs P = nms_ptrs(ws_all + (size_t)f * ws_stride, L);
    if (n == 0) {
        if (tid == 0) n_kept_all[f] = 0;
        return;
    }
    const long long r2 = (long long)radius * (long long)radius;

    if (radius >= 0 && L.champ) {
        // ---- champion rounds continued by this one workgroup on a compact list of the cells that are still
        //      open: same two steps as the wide rounds (champ_cell, phase_c_wave), a block barrier in place of
        //      the kernel boundary (all waves of a workgroup share one L1, so global writes made before the
        //      barrier are seen after it).  Long dependency chains (large r, dense hits) need dozens of rounds
        //      with little work each; here a round costs a few microseconds instead of two launches. ----
        if (tid == 0) sh_cnt = 0;
        __syncthreads();
        for (int c = tid; c < L.ncell; c += NT)
            if (P.cell_und[c] != 0) P.listA[atomicAdd(&sh_cnt, 1u)] = (uint32_t)c;
        __syncthreads();
        uint32_t *cur = P.listA, *nxt = P.listB;
        int n_live = (int)sh_cnt;
        const int n_open0 = n_live;
        int round = round0;
        __syncthreads();
        while (n_live > 0) {
            for (int i = tid; i < n_live; i += NT) champ_cell(P, L, (int)cur[i]);
            if (tid == 0) sh_cnt = 0;
            __syncthreads();
            for (int i = tid; i < n_live; i += NT) {
                const uint32_t c = cur[i];
                if (P.cell_und[c] != 0) nxt[atomicAdd(&sh_cnt, 1u)] = c; // order inside the list is irrelevant
            }
            __syncthreads();
            n_live = (int)sh_cnt;
            { uint32_t *t = cur; cur = nxt; nxt = t; }
            for (int base = 0; base < n_live; base += NT) { // whole wavefronts: lane = one open cell
                const int i = base + tid;
                const bool in = i < n_live;
                const int c = in ? (int)cur[i] : 0;
                const RawOut raw{raw_xy_out + (size_t)f * n_cap, raw_score_out + (size_t)f * n_cap};
                if (L.R <= 2) phase_c_wave<2>(P, L, radius, round, c, in, P.cell_start, 0, raw);
                else phase_c_wave<3>(P, L, radius, round, c, in, P.cell_start, 0, raw);
            }
            __syncthreads();
            round++;
            if (round - round0 > n_open0 + 16) { // every round closes at least the best open cell; anything else is a bug
                if (tid == 0) atomicOr(status, (int)PGX_ST_INTERNAL);
                break;
            }
        }
    } else if (radius >= 0) {
        // ---- remaining rounds on an active list (normally empty after the wide rounds) ----
        if (tid == 0) sh_cnt = 0;
        __syncthreads();
        for (int c = tid; c < L.ncell; c += NT) {
            if (P.cell_und[c] == 0) continue;
            for (uint32_t p = P.cell_start[c]; p < P.cell_start[c + 1]; p++)
                if (P.rec[p].w == ST_UNDECIDED) P.listA[atomicAdd(&sh_cnt, 1u)] = p;
        }
        __syncthreads();
        uint32_t *cur = P.listA, *nxt = P.listB;
        int n_act = (int)sh_cnt;
        __syncthreads();
        while (n_act > 0) {
            for (int a = tid; a < n_act; a += NT) {
                const uint32_t p = cur[a];
                if (!tail_beaten(P, L, p, r2)) *rec_state(P.rec, p) = ST_NEW;
            }
            __syncthreads();
            if (tid == 0) sh_cnt = 0;
            __syncthreads();
            for (int a = tid; a < n_act; a += NT) {
                const uint32_t p = cur[a];
                if (P.rec[p].w == ST_NEW) continue;
                if (tail_suppressed(P, L, p, r2)) *rec_state(P.rec, p) = ST_SUPPRESSED;
                else nxt[atomicAdd(&sh_cnt, 1u)] = p; // order inside the list is irrelevant
            }
            __syncthreads();
            for (int a = tid; a < n_act; a += NT) {
                const uint32_t p = cur[a];
                const uint4 me = P.rec[p];
                if (me.w == ST_NEW) {
                    *rec_state(P.rec, p) = ST_ACCEPTED;
                    P.sortkeys[atomicAdd(&P.counters[2], 1u)] = sort_key((int)me.y, me.z);
                }
            }
            const int n_prev = n_act;
            n_act = (int)sh_cnt;
            uint32_t *t = cur; cur = nxt; nxt = t;
            __syncthreads();
            if (n_act >= n_prev) { // every round decides at least the best open point; anything else is a bug
                if (tid == 0) atomicOr(status, (int)PGX_ST_INTERNAL);
                break;
            }
        }
    }

    tail_order(P, L, radius, n, raw_score, order, n_kept_all, f, kp_cap, kp_soft, status, lds_keys, SORT_LDS_MAX, wsum, &sh_cnt, &sh_max);
}

// mask rounds continued by one workgroup per frame on a compact list of the cells whose champion entry is not zero yet
// (open cells, plus closed ones that still have to take their stale entry down), a block barrier in place of the kernel
// boundary; then the output order.
// 512 threads and 64 KiB of LDS: two waves per SIMD at <= 128 registers, so that a workgroup finds room on a CU that still
// runs ONE distance-kernel workgroup of another job (240 registers per wave, 35 KiB of LDS; 256 threads were measured too:
// `nms` 0.54 ms per step against 0.49).  With 1024 threads x 121
// registers and 128 KiB it needed a completely EMPTY CU, which never happens while the other job's distance kernel has
// workgroups waiting: the whole detect chain of a second job in flight stalled here until that launch had drained (2.5 ms
// instead of 0.1).  The rounds work on a few hundred listed cells, the order pass on 15 k bitmap words: alone it is as fast.
constexpr int MT_NT = 512;
constexpr uint32_t MT_SORT_LDS = 8192; // u64 keys -> 64 KiB
template <int RR>
__global__ __launch_bounds__(MT_NT) void k_nmsm_tail(const int32_t *raw_score_all, const int32_t *__restrict__ n_raw_all, int n_cap,
                                                  NmsLayout L, int radius, unsigned char *ws_all, size_t ws_stride,
                                                  uint32_t *__restrict__ order_all, int32_t *__restrict__ n_kept_all, int kp_cap,
                                                  int *status, uint32_t *raw_xy_out, int32_t *raw_score_out, int kp_soft,
                                                  const unsigned long long *seg_all, const uint32_t *segoff_all, int H, int ntx)
{
    extern __shared__ unsigned long long lds_keys[];
    __shared__ uint32_t wsum[MT_NT / 64];
    __shared__ uint32_t sh_cnt;
    __shared__ int sh_max;
    const int f = blockIdx.x, tid = threadIdx.x;
    const int n = clamp_n(n_raw_all, f, n_cap);
    if (n == 0) {
        if (tid == 0) n_kept_all[f] = 0;
        return;
    }
    NmsPtrs P = nms_ptrs(ws_all + (size_t)f * ws_stride, L);
    {
        const MaskPtrs M = mask_ptrs(ws_all + (size_t)f * ws_stride, L);
        const unsigned long long *disk = reinterpret_cast<const unsigned long long *>(ws_all + L.off_disk);
        const size_t nseg = (size_t)H * ntx;
        const MaskOut out{raw_xy_out + (size_t)f * n_cap, raw_score_out + (size_t)f * n_cap, seg_all + (size_t)f * nseg * 4,
                          segoff_all + (size_t)f * nseg, ntx};
        if (tid == 0) sh_cnt = 0;
        __syncthreads();
        for (int c = tid; c < L.ncell; c += MT_NT) {
            const int cy = c / L.gw, cx = c - cy * L.gw;
            if (M.ent[(cy + RR) * L.cgw + cx + RR] != 0u) M.listA[atomicAdd(&sh_cnt, 1u)] = (uint32_t)c;
        }
        __syncthreads();
        uint32_t *cur = M.listA, *nxt = M.listB;
        int n_live = (int)sh_cnt;
        const int n_open0 = n_live;
        int rounds = 0;
        __syncthreads();
        while (n_live > 0) {
            for (int base = 0; base < n_live; base += MT_NT) { // whole wavefronts: lane = one listed cell
                const int i = base + tid;
                const bool in = i < n_live;
                const int c = in ? (int)cur[i] : 0;
                mask_round_wave<RR>(M, L, disk, radius, c, in, out);
            }
            if (tid == 0) sh_cnt = 0;
            __syncthreads();
            for (int i = tid; i < n_live; i += MT_NT) {
                const uint32_t c = cur[i];
                const int cy = (int)c / L.gw, cx = (int)c - cy * L.gw;
                if (M.ent[(cy + RR) * L.cgw + cx + RR] != 0u) nxt[atomicAdd(&sh_cnt, 1u)] = c; // order inside the list is irrelevant
            }
            __syncthreads();
            n_live = (int)sh_cnt;
            { uint32_t *t = cur; cur = nxt; nxt = t; }
            __syncthreads();
            // with fresh entries every round accepts the best open point; an entry may be one round stale
            if (++rounds > 2 * n_open0 + 16) {
                if (tid == 0) atomicOr(status, (int)PGX_ST_INTERNAL);
                break;
            }
        }
    }
    __syncthreads();
    tail_order(P, L, radius, n, raw_score_all + (size_t)f * n_cap, order_all + (size_t)f * kp_cap, n_kept_all, f, kp_cap, kp_soft,
               status, lds_keys, MT_SORT_LDS, wsum, &sh_cnt, &sh_max);
}

} // namespace

size_t pgx_nms_ws_bytes(int W, int H, int radius, int n_cap, bool planes) { return nms_layout(W, H, radius, n_cap, planes).total; }

bool pgx_nms_fills_raw_lists(int W, int H, int radius, int n_cap) { return radius >= 0 && nms_layout(W, H, radius, n_cap, true).champ != 0; }

namespace {

struct NmsLaunch {
    hipStream_t s;
    uint32_t *raw_xy; int32_t *raw_score; const int32_t *n_raw;
    int F, n_cap, W, H, radius;
    unsigned char *ws; size_t ws_stride;
    uint32_t *order; int32_t *n_kept; int kp_cap; int kp_soft = 0; int *status;
    const unsigned long long *seg; const uint32_t *segoff;
    NmsLayout L;
    bool planes;
};

void nms_setup(const NmsLaunch &a)
{
    const NmsLayout &L = a.L;
    hipStream_t s = a.s;
    const dim3 pgrid((a.n_cap + 255) / 256, a.F);
    const dim3 bgrid((L.ncell + 255) / 256, a.F);
    const int ntx = (a.W + 63) / 64;
    if (L.mask) {
        const dim3 sgrid((L.cgw * (L.gh + 2 * L.R) + 255) / 256, a.F);
        if (L.R <= 2) hipLaunchKernelGGL(k_nmsm_setup<2>, sgrid, dim3(256), 0, s, a.seg, a.segoff, a.n_raw, a.W, a.H, ntx, a.n_cap, a.radius, L, a.ws, a.ws_stride);
        else hipLaunchKernelGGL(k_nmsm_setup<3>, sgrid, dim3(256), 0, s, a.seg, a.segoff, a.n_raw, a.W, a.H, ntx, a.n_cap, a.radius, L, a.ws, a.ws_stride);
        return;
    }
    hipLaunchKernelGGL(k_nms_zero, dim3((L.ncell + 256) / 256, a.F), dim3(256), 0, s, L, a.ws, a.ws_stride);
    if (a.planes && L.cs == 8) hipLaunchKernelGGL(k_nms_bin_planes8<0>, bgrid, dim3(256), 0, s, a.seg, a.segoff, a.W, a.H, ntx, a.n_cap, L, a.ws, a.ws_stride);
    else if (a.planes) hipLaunchKernelGGL(k_nms_bin_planes<0>, bgrid, dim3(256), 0, s, a.seg, a.segoff, a.W, a.H, ntx, a.n_cap, L, a.ws, a.ws_stride);
    else hipLaunchKernelGGL(k_nms_count, pgrid, dim3(256), 0, s, a.raw_xy, a.n_raw, a.n_cap, L, a.ws, a.ws_stride);
    hipLaunchKernelGGL(k_nms_cellscan, dim3(a.F), dim3(NT), 0, s, L, a.ws, a.ws_stride);
    if (L.champ && L.cs == 8) hipLaunchKernelGGL(k_nms_bin_planes8<2>, bgrid, dim3(256), 0, s, a.seg, a.segoff, a.W, a.H, ntx, a.n_cap, L, a.ws, a.ws_stride);
    else if (L.champ) hipLaunchKernelGGL(k_nms_bin_planes<2>, bgrid, dim3(256), 0, s, a.seg, a.segoff, a.W, a.H, ntx, a.n_cap, L, a.ws, a.ws_stride);
    else if (a.planes) hipLaunchKernelGGL(k_nms_bin_planes<1>, bgrid, dim3(256), 0, s, a.seg, a.segoff, a.W, a.H, ntx, a.n_cap, L, a.ws, a.ws_stride);
    else hipLaunchKernelGGL(k_nms_scatter, pgrid, dim3(256), 0, s, a.raw_xy, a.raw_score, a.n_raw, a.n_cap, L, a.ws, a.ws_stride);
}

// whole-chip rounds [r0, r0 + n)
void nms_rounds(const NmsLaunch &a, int r0, int n)
{
    const NmsLayout &L = a.L;
    hipStream_t s = a.s;
    const dim3 bgrid((L.ncell + 255) / 256, a.F);
    const dim3 cgrid((L.ncell + 3) / 4, a.F);
    const int ntx = (a.W + 63) / 64;
    for (int r = r0; r < r0 + n; r++) {
        if (L.mask) {
            const int tiles_x = (L.gw + MT_W - 1) / MT_W, tiles_y = (L.gh + MT_H - 1) / MT_H;
            const dim3 tgrid(tiles_x * tiles_y, a.F);
            if (L.R <= 2) hipLaunchKernelGGL(k_nmsm_round<2>, tgrid, dim3(256), 0, s, L, a.radius, a.ws, a.ws_stride, a.raw_xy, a.raw_score, a.n_cap, a.seg, a.segoff, a.H, ntx, tiles_x);
            else hipLaunchKernelGGL(k_nmsm_round<3>, tgrid, dim3(256), 0, s, L, a.radius, a.ws, a.ws_stride, a.raw_xy, a.raw_score, a.n_cap, a.seg, a.segoff, a.H, ntx, tiles_x);
        } else if (L.champ) {
            if (r > 0) hipLaunchKernelGGL(k_nms_champ, bgrid, dim3(256), 0, s, L, a.ws, a.ws_stride);
            if (L.R <= 2) {
                if (r <= 1) hipLaunchKernelGGL((k_nms_phase_c<2, true>), bgrid, dim3(256), 0, s, L, a.radius, r, a.ws, a.ws_stride, a.raw_xy, a.raw_score, a.n_cap);
                else hipLaunchKernelGGL((k_nms_phase_c<2, false>), bgrid, dim3(256), 0, s, L, a.radius, r, a.ws, a.ws_stride, a.raw_xy, a.raw_score, a.n_cap);
            } else {
                if (r <= 1) hipLaunchKernelGGL((k_nms_phase_c<3, true>), bgrid, dim3(256), 0, s, L, a.radius, r, a.ws, a.ws_stride, a.raw_xy, a.raw_score, a.n_cap);
                else hipLaunchKernelGGL((k_nms_phase_c<3, false>), bgrid, dim3(256), 0, s, L, a.radius, r, a.ws, a.ws_stride, a.raw_xy, a.raw_score, a.n_cap);
            }
        } else {
            hipLaunchKernelGGL(k_nms_phase_a, cgrid, dim3(256), 0, s, L, a.radius, r, a.ws, a.ws_stride);
            hipLaunchKernelGGL(k_nms_push, dim3(r == 0 ? 256 : 64, a.F), dim3(256), 0, s, L, a.radius, r, a.ws, a.ws_stride);
        }
    }
}

void nms_finish(const NmsLaunch &a, int round0)
{
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_nms_tail), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)(SORT_LDS_MAX * 8));
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_nmsm_tail<2>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)(MT_SORT_LDS * 8));
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_nmsm_tail<3>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)(MT_SORT_LDS * 8));
        attr_set = true;
    }
    if (a.L.mask) {
        auto kern = a.L.R <= 2 ? &k_nmsm_tail<2> : &k_nmsm_tail<3>;
        hipLaunchKernelGGL(kern, dim3(a.F), dim3(MT_NT), MT_SORT_LDS * 8, a.s, a.raw_score, a.n_raw, a.n_cap, a.L, a.radius, a.ws,
                           a.ws_stride, a.order, a.n_kept, a.kp_cap, a.status, a.raw_xy, a.raw_score, a.kp_soft, a.seg, a.segoff, a.H,
                           (a.W + 63) / 64);
        return;
    }
    hipLaunchKernelGGL(k_nms_tail, dim3(a.F), dim3(NT), SORT_LDS_MAX * 8, a.s, a.raw_score, a.n_raw, a.n_cap, a.L, a.radius, a.ws,
                       a.ws_stride, a.order, a.n_kept, a.kp_cap, a.status, round0, a.raw_xy, a.raw_score, a.kp_soft);
}

NmsLaunch nms_args(hipStream_t s, uint32_t *raw_xy, int32_t *raw_score, const int32_t *n_raw, int F, int n_cap,
                   int W, int H, int radius, void *wsv, size_t ws_stride, uint32_t *order, int32_t *n_kept, int kp_cap,
                   int *status, const unsigned long long *seg, const uint32_t *segoff)
{
    NmsLaunch a;
    a.s = s; a.raw_xy = raw_xy; a.raw_score = raw_score; a.n_raw = n_raw; a.F = F; a.n_cap = n_cap; a.W = W; a.H = H;
    a.radius = radius; a.ws = reinterpret_cast<unsigned char *>(wsv); a.ws_stride = ws_stride; a.order = order;
    a.n_kept = n_kept; a.kp_cap = kp_cap; a.status = status; a.seg = seg; a.segoff = segoff;
    const bool have_planes = seg && segoff;
    a.L = nms_layout(W, H, radius, n_cap, have_planes);
    a.planes = have_planes && (a.L.cs == 8 || a.L.cs == 16 || a.L.cs == 32 || a.L.cs == 64);
    return a;
}

// mask rounds: 5 measured best on 64 frames of 1080p at r = 16 (2: 0.75 ms, 3: 0.57, 4: 0.50, 5: 0.49, 6: 0.50, 8: 0.52)
int wide_rounds_default(bool mask = false)
{
    return mask ? 5 : WIDE_ROUNDS;
}

} // namespace

// asynchronous form (fused detect path): a fixed number of whole-chip rounds, the tail kernel finishes
void pgx_launch_nms(hipStream_t s, uint32_t *raw_xy, int32_t *raw_score, const int32_t *n_raw, int F,
                    int n_cap, int W, int H, int radius, void *wsv, size_t ws_stride, uint32_t *order,
                    int32_t *n_kept, int kp_cap, int *status, const unsigned long long *seg, const uint32_t *segoff,
                    bool kp_soft)
{
    if (F <= 0) return;
    NmsLaunch a = nms_args(s, raw_xy, raw_score, n_raw, F, n_cap, W, H, radius, wsv, ws_stride, order, n_kept, kp_cap,
                                 status, seg, segoff);
    a.kp_soft = kp_soft ? 1 : 0;
    int used = 0;
    if (radius >= 0) {
        nms_setup(a);
        // the general path's tail is a plain serial finish: give it fewer leftovers than the champion tail needs
        used = a.L.mask ? wide_rounds_default(true) : (a.L.champ ? wide_rounds_default() : wide_rounds_default() + 6);
        nms_rounds(a, 0, used);
    }
    nms_finish(a, used);
}

// synchronous form (stage API, one list, general path): whole-chip rounds until a round accepts nothing -- then
// everything is decided, since every round accepts at least the best undecided point.  The host reads one
// counter per batch of rounds; long dependency chains (large r on dense lists need 50-100 rounds) never reach
// the serial tail, which only sorts.
hipError_t pgx_launch_nms_sync(hipStream_t s, const uint32_t *raw_xy, const int32_t *raw_score, const int32_t *n_raw,
                               int n_cap, int W, int H, int radius, void *wsv, size_t ws_stride, uint32_t *order,
                               int32_t *n_kept, int kp_cap, int *status)
{
    const NmsLaunch a = nms_args(s, const_cast<uint32_t *>(raw_xy), const_cast<int32_t *>(raw_score), n_raw, 1, n_cap, W, H, radius,
                                 wsv, ws_stride, order, n_kept, kp_cap, status, nullptr, nullptr); // general path: lists are read only
    int r0 = 0;
    if (radius >= 0) {
        nms_setup(a);
        const int batch = 16;
        const uint32_t *counters = reinterpret_cast<const uint32_t *>(a.ws + a.L.off_counters);
        for (; r0 < (1 << 20); r0 += batch) {
            nms_rounds(a, r0, batch);
            uint32_t last = 0; // accepted in the batch's last round (k_nms_push leaves that round's counter in place)
            hipError_t e = hipMemcpyAsync(&last, counters + ((r0 + batch - 1) & 1), 4, hipMemcpyDeviceToHost, s);
            if (e == hipSuccess) e = hipStreamSynchronize(s);
            if (e != hipSuccess) return e;
            if (last == 0) { r0 += batch; break; }
        }
    }
    nms_finish(a, r0);
    return hipGetLastError();
}

