// k_tracks.hip -- the global track graph over the (gathered) match lists, built where the lists already sit: in HBM.
//
// north_star: "a single RCCL all-gather ... to collect per-pair match lists into the global track graph"; SURVEY 8f-3.  The
// reference has no multi-frame structure (TestService.cs:80-96 handles exactly one image pair); what it does hold
// is the distance gate (python_src/scripts/match_keypoints.py:23,127; `new KeypointMatching(100)` in the commented code of
// dotnet_src/Photogrammetry/Program.cs:165,224).  Semantics (include/pgx.h, oracle/tracks_np.py) are order-independent so
// that this parallel form is bit-identical to a sequential one:
//   nodes  (frame, keypoint) -> id = frame * stride + keypoint      (frame = the caller's global frame number)
//   edges  match entries with dist <= max_dist (never the (0, 0, int.MaxValue) tail)
//   tracks connected components; a component with two keypoints of one frame is dropped as a whole
//   order  tracks by first node, nodes ascending
//
// Kernels (all integer, latency / atomic bound; the lists are read once: 12 B per match entry):
//   k_trk_init     parent[x] = x, zeroes, empty per-frame hash tables, track_of = -1
//   k_trk_union    one thread per match entry: lock-free union-find with RANDOM linking -- of two roots the one with the larger
//                  hashed id hooks under the other (atomicCAS on a root only).  Linking by node id (ids grow with the frame
//                  number, and so do the tracks) builds long chains under 2016 racing edges per track: 0.91 ms for the bench
//                  job's 5.7 M edges against 0.57 with hashed priorities; a first pass that points every node at its smallest
//                  neighbour (0.21 ms) and a first pass over the consecutive-frame pairs only bought nothing: the time is the
//                  56 k-node junk component (hub descriptors near the image border: 3.5 M of the edges), whose 460 k failed
//                  CAS are spread over its nodes (at most 12 k on one address) -- measured, see DESIGN.md
//   k_trk_flatten  one thread per node: root (into its own array); per workgroup (= one frame's nodes) the distinct roots are
//                  counted in LDS first, then one global atomicAdd (component size), one atomicMin (the component's first
//                  node: its canonical name, since the root is whichever node the linking left on top) and the frame-conflict
//                  test per distinct root: the root goes into its FRAME's hash table (open addressing, atomicCAS); finding it
//                  there already = two keypoints of one frame in one component
//   k_trk_scan_*   exclusive scan over node ids of (kept components at their first node, their sizes): track index and node
//                  offset in first-node order
//   k_trk_place    nodes into their track's segment (atomic cursor: unordered)
//   k_trk_rank     a node's final place = number of segment entries below it (a kept track has at most one node per frame, so
//                  segments are short: <= n_frames)
// Same-address atomics serialise at about 10 ns each on this part: one global counter of the edges used cost 1.3 ms per launch
// (130 k wavefront atomics), per-node atomicAdds on the junk component's root 0.56 ms -- hence the LDS stages and TRK_ECNT.
#include "pgx_internal.h"

namespace {

constexpr int TRK_NT = 256;          // threads per workgroup of the per-entry / per-node kernels
constexpr int SCAN_NT = 256;         // scan kernels: 256 threads x 4 items.  Small on purpose: a 1024-thread workgroup needs 4 free wave
                                     // slots on every SIMD of one CU at once, and beside another job's distance kernel it waited
                                     // 3 ms for them (the one-workgroup k_trk_scan_sums, kernel trace of round 5)
constexpr int SCAN_ITEMS = 4 * SCAN_NT;
constexpr uint32_t TRK_EMPTY = 0xFFFFFFFFu;
constexpr int TRK_ECNT = 256;        // slots of the edge counter
// Workgroups of k_trk_union (grid-stride over the match entries).  The bench job's graph alone, ms per call: 32 workgroups 2.79,
// 64: 1.45, 128: 0.82, 192: 0.62, 256: 0.55, 512: 0.59, 1024: 0.65, 2048: 0.81, 4096: 0.90, one per 256 entries (32 k): 0.70 --
// fewer threads in flight see fresher trees and lose fewer CAS, too few leave the dependent L2 reads uncovered; one per CU it is
constexpr int TRK_UNION_GRID = 256;
static_assert(TRK_ECNT <= SCAN_NT, "k_trk_scan_sums adds the edge-counter slots up in one block scan");
constexpr int FL_SLOTS = 512;        // LDS aggregation table of k_trk_flatten (>= 2 * TRK_NT)

struct TrkArgs {
    const pgx_pair *matches;   // [M][stride]
    const int32_t *counts;     // [F] by slot
    const int32_t *pairlist;   // [M][2] slots
    const int32_t *frame_ids;  // [F] slot -> global frame number, -1 = not part of this graph; nullptr = identity
    int M, F, stride, n_frames, max_dist, min_len;
    long long N;               // n_frames * stride
    int T;                     // hash table entries per frame (power of two >= 2 * stride)
    int32_t *parent, *root, *rep, *size, *flag, *cursor, *tidx, *noff, *tmp;
    uint32_t *table;           // [n_frames][T]
    unsigned long long *bsum;  // per scan block: (kept roots << 32) | their nodes
    int32_t *ecnt;             // [TRK_ECNT] partial counts of the edges used (one hot counter would serialise 130 k atomics)
    int32_t *track_of, *offsets, *nodes, *summary;
};

__device__ __forceinline__ int ld(const int32_t *p)
{
    // a plain load the compiler may not cache in a register across the loops below (no cache-bypass bits at this scope)
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}
__device__ __forceinline__ void st(int32_t *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); }

__device__ __forceinline__ uint32_t trk_prio(int x) { return (uint32_t)x * 2654435761u; }   // a bijection on 32 bits: no ties

// Root of x with intermediate pointer jumping.  Invariant: prio(parent[v]) < prio(v) unless v is a root (parent[v] == v), and
// parent[v] only ever moves to an ancestor of v.  A value read here may be STALE (another XCD's L2, this CU's L1): every
// earlier value of parent[v] is v itself or an ancestor, so a walk over stale values still descends in priority (it ends), and
// a stale read can only name a node that is no longer a root -- the caller's atomicCAS then fails and returns the truth.
__device__ __forceinline__ int trk_find(int32_t *parent, int x)
{
    int curr = ld(parent + x);
    if (curr != x) {
        int prev = x, next;
        while (curr != (next = ld(parent + curr))) {
            st(parent + prev, next);
            prev = curr;
            curr = next;
        }
    }
    return curr;
}

__device__ __forceinline__ int fid_of(const TrkArgs &a, int slot) { return a.frame_ids ? a.frame_ids[slot] : slot; }

__global__ __launch_bounds__(TRK_NT) void k_trk_init(TrkArgs a)
{
    const long long nthreads = (long long)gridDim.x * TRK_NT;
    const long long t0 = (long long)blockIdx.x * TRK_NT + threadIdx.x;
    for (long long x = t0; x < a.N; x += nthreads) {
        a.parent[x] = (int)x;
        a.root[x] = -1;
        a.rep[x] = 0x7FFFFFFF;
        a.size[x] = 0;
        a.flag[x] = 0;
        a.cursor[x] = 0;
        a.track_of[x] = -1;
    }
    const long long nt = (long long)a.n_frames * a.T;
    for (long long i = t0; i < nt; i += nthreads) a.table[i] = TRK_EMPTY;
    if (t0 < 8) a.summary[t0] = 0;
    if (t0 < TRK_ECNT) a.ecnt[t0] = 0;
}

// The gated edge of this thread's match entry: work item -> (image pair, chunk of its list), thread -> entry.  False for the
// whole workgroup when the pair names a slot outside the graph.
__device__ __forceinline__ bool trk_edge(const TrkArgs &a, long long item, int &u, int &v)
{
    const int chunks = (a.stride + TRK_NT - 1) / TRK_NT;
    const int m = (int)(item / chunks), ch = (int)(item % chunks);
    const int sa = a.pairlist[2 * m], sb = a.pairlist[2 * m + 1];
    if ((unsigned)sa >= (unsigned)a.F || (unsigned)sb >= (unsigned)a.F) return false;
    const int fa = fid_of(a, sa), fb = fid_of(a, sb);
    if ((unsigned)fa >= (unsigned)a.n_frames || (unsigned)fb >= (unsigned)a.n_frames) return false;
    int ca = a.counts[sa], cb = a.counts[sb];
    ca = ca > a.stride ? a.stride : ca;
    cb = cb > a.stride ? a.stride : cb;
    const int e = ch * TRK_NT + threadIdx.x;
    if (e >= ca) return false;
    const pgx_pair p = a.matches[(size_t)m * a.stride + e];
    u = fa * a.stride + p.k1;
    v = fb * a.stride + p.k2;
    // KeypointMatching.cs:40-42: the tail entries carry int.MaxValue and never link, whatever max_dist is
    return p.dist <= a.max_dist && p.dist != PGX_DIST_NONE && (unsigned)p.k1 < (unsigned)ca && (unsigned)p.k2 < (unsigned)cb;
}

// Lock-free union of every edge's two trees.  Work items = M * ceil(stride / TRK_NT) chunks of TRK_NT match entries, taken
// grid-stride by a bounded grid (pgx_launch_tracks): the kernel is latency-bound on dependent L2 reads and atomics, and a
// flood of 32 k tiny workgroups takes every wave slot that comes free on the chip -- another job's distance kernel (240
// registers per wave) then never gets its second wave per SIMD back (bench step 7.2 -> 8.2 ms for a 0.8 ms graph).
__global__ __launch_bounds__(TRK_NT) void k_trk_union(TrkArgs a, long long n_items)
{
    int mine = 0;   // edges used, counted by lane 0 of every wave
    for (long long item = blockIdx.x; item < n_items; item += gridDim.x) {
        int u = 0, v = 0;
        const bool edge = trk_edge(a, item, u, v);
        const unsigned long long bal = __ballot(edge);
        mine += __popcll(bal);
        if (!edge) continue;
        int ru = trk_find(a.parent, u), rv = trk_find(a.parent, v);
        while (ru != rv) {
            if (trk_prio(ru) < trk_prio(rv)) { const int t = ru; ru = rv; rv = t; }   // ru hooks under rv
            const int old = atomicCAS(a.parent + ru, ru, rv);
            if (old == ru) break;
            ru = trk_find(a.parent, old);   // ru was no root any more: go on from the root above its true parent
        }
    }
    // edges used: through LDS into one of TRK_ECNT slots (same-address atomics serialise); k_trk_scan_sums adds them up
    __shared__ int n_edges;
    if (threadIdx.x == 0) n_edges = 0;
    __syncthreads();
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(&n_edges, mine);
    __syncthreads();
    if (threadIdx.x == 0 && n_edges) atomicAdd(a.ecnt + (blockIdx.x & (TRK_ECNT - 1)), n_edges);
}

// grid: F * ceil(stride / TRK_NT); workgroup -> (slot, chunk of its keypoints)
// Component sizes and the frame-conflict test are per (root, frame) facts, and a workgroup's nodes all lie in ONE frame: the
// roots are first counted in an LDS table, and each distinct root of the workgroup then costs one global atomicAdd and one
// insert into the frame's table.  (Straight per-node global atomics serialise on a big component's root: the 56 k nodes of
// the bench job's largest (inconsistent) component took 0.56 of the launch's 0.64 ms.)  Two nodes of the workgroup with one
// root are two keypoints of this frame in one component: flagged here, without the global table.
__global__ __launch_bounds__(TRK_NT) void k_trk_flatten(TrkArgs a)
{
    __shared__ uint32_t hkey[FL_SLOTS];
    __shared__ int hcnt[FL_SLOTS], hmin[FL_SLOTS];
    const int chunks = (a.stride + TRK_NT - 1) / TRK_NT;
    const int s = blockIdx.x / chunks, k = (blockIdx.x % chunks) * TRK_NT + threadIdx.x;
    const int f = fid_of(a, s);
    if ((unsigned)f >= (unsigned)a.n_frames) return;   // uniform
    int c = a.counts[s];
    c = c > a.stride ? a.stride : c;
    if ((blockIdx.x % chunks) * TRK_NT >= c) return;   // uniform
    for (int i = threadIdx.x; i < FL_SLOTS; i += TRK_NT) { hkey[i] = TRK_EMPTY; hcnt[i] = 0; hmin[i] = 0x7FFFFFFF; }
    __syncthreads();
    if (k < c) {
        const int x = f * a.stride + k;
        const int r = trk_find(a.parent, x);
        // into an array of its own: other threads' pointer jumping still stores (older) ancestors into parent[x] while this runs
        a.root[x] = r;
        uint32_t h = ((uint32_t)r * 2654435761u) >> 9;
        for (;;) {
            h &= (uint32_t)(FL_SLOTS - 1);
            const uint32_t old = atomicCAS(&hkey[h], TRK_EMPTY, (uint32_t)r);
            if (old == TRK_EMPTY || old == (uint32_t)r) { atomicAdd(&hcnt[h], 1); atomicMin(&hmin[h], x); break; }
            h++;
        }
    }
    __syncthreads();
    uint32_t *tab = a.table + (size_t)f * a.T;
    for (int i = threadIdx.x; i < FL_SLOTS; i += TRK_NT) {
        const uint32_t r = hkey[i];
        if (r == TRK_EMPTY) continue;
        const int n = hcnt[i];
        atomicAdd(a.size + r, n);
        atomicMin(a.rep + r, hmin[i]);
        if (n > 1) a.flag[r] = 1;
        // the root into this frame's table; already there = another workgroup saw a keypoint of this frame in the component
        uint32_t h = (r * 2654435761u) >> 7;
        for (;;) {
            h &= (uint32_t)(a.T - 1);
            const uint32_t old = atomicCAS(tab + h, TRK_EMPTY, r);
            if (old == TRK_EMPTY) break;
            if (old == r) { a.flag[r] = 1; break; }
            h++;
        }
    }
}

// A component is named by its FIRST node (smallest id): that is where it stands in the scan.  -> root of the component x names, or -1
__device__ __forceinline__ int trk_named_here(const TrkArgs &a, long long x)
{
    if (x >= a.N) return -1;
    const int r = a.root[x];          // -1 for slots beyond a frame's count
    return (r >= 0 && a.rep[r] == (int)x) ? r : -1;
}

__device__ __forceinline__ unsigned long long kept_item(const TrkArgs &a, int r)
{
    if (r < 0) return 0ull;
    const int sz = a.size[r];
    if (sz < a.min_len || a.flag[r]) return 0ull;
    return (1ull << 32) | (unsigned)sz;
}

// block-wide exclusive scan of one value per thread (SCAN_NT threads); returns the exclusive prefix, *total = block sum
__device__ __forceinline__ unsigned long long block_excl_scan(unsigned long long v, unsigned long long *lds /*[16]*/, unsigned long long *total)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    unsigned long long inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned long long o = __shfl_up(inc, d, 64);
        if (lane >= d) inc += o;
    }
    if (lane == 63) lds[w] = inc;
    __syncthreads();
    unsigned long long base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < SCAN_NT / 64; i++) {
        const unsigned long long s = lds[i];
        if (i < w) base += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

__global__ __launch_bounds__(SCAN_NT) void k_trk_scan_reduce(TrkArgs a)
{
    __shared__ unsigned long long lds[SCAN_NT / 64];
    const long long x0 = (long long)blockIdx.x * SCAN_ITEMS + (long long)threadIdx.x * 4;
    unsigned long long s = 0;
    int dropped = 0, dropped_nodes = 0, longest = 0, largest_dropped = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int r = trk_named_here(a, x0 + i);
        const unsigned long long it = kept_item(a, r);
        s += it;
        if (it) longest = max(longest, (int)(unsigned)it);
        if (r >= 0 && a.flag[r]) {
            dropped++;
            dropped_nodes += a.size[r];
            largest_dropped = max(largest_dropped, a.size[r]);
        }
    }
    __shared__ int agg[4];   // dropped components, their nodes, longest kept, largest dropped: one global atomic each per workgroup
    if (threadIdx.x < 4) agg[threadIdx.x] = 0;
    unsigned long long tot;
    (void)block_excl_scan(s, lds, &tot);   // (its barriers order the zeroing above before the adds below)
    if (threadIdx.x == 0) a.bsum[blockIdx.x] = tot;
    if (dropped) { atomicAdd(&agg[0], dropped); atomicAdd(&agg[1], dropped_nodes); atomicMax(&agg[3], largest_dropped); }
    if (longest) atomicMax(&agg[2], longest);
    __syncthreads();
    if (threadIdx.x == 0) {
        if (agg[0]) { atomicAdd(a.summary + 2, agg[0]); atomicAdd(a.summary + 3, agg[1]); atomicMax(a.summary + 6, agg[3]); }
        if (agg[2]) atomicMax(a.summary + 5, agg[2]);
    }
}

// one workgroup: exclusive scan of the block sums in place; totals into the summary and the closing offset
__global__ __launch_bounds__(SCAN_NT) void k_trk_scan_sums(TrkArgs a, int nb)
{
    __shared__ unsigned long long lds[SCAN_NT / 64];
    unsigned long long carry = 0;
    for (int b0 = 0; b0 < nb; b0 += SCAN_NT) {
        const int b = b0 + threadIdx.x;
        const unsigned long long v = b < nb ? a.bsum[b] : 0ull;
        unsigned long long tot;
        const unsigned long long ex = block_excl_scan(v, lds, &tot);
        if (b < nb) a.bsum[b] = carry + ex;
        carry += tot;
    }
    unsigned long long etot;
    (void)block_excl_scan(threadIdx.x < TRK_ECNT ? (unsigned long long)a.ecnt[threadIdx.x] : 0ull, lds, &etot);
    if (threadIdx.x == 0) {
        const int nt = (int)(carry >> 32), nn = (int)(unsigned)carry;
        a.summary[0] = nt;
        a.summary[1] = nn;
        a.summary[4] = (int)etot;
        a.offsets[nt] = nn;
    }
}

__global__ __launch_bounds__(SCAN_NT) void k_trk_scan_apply(TrkArgs a)
{
    __shared__ unsigned long long lds[SCAN_NT / 64];
    const long long x0 = (long long)blockIdx.x * SCAN_ITEMS + (long long)threadIdx.x * 4;
    unsigned long long it[4], s = 0;
    int rr[4];
#pragma unroll
    for (int i = 0; i < 4; i++) { rr[i] = trk_named_here(a, x0 + i); it[i] = kept_item(a, rr[i]); s += it[i]; }
    unsigned long long tot;
    unsigned long long ex = a.bsum[blockIdx.x] + block_excl_scan(s, lds, &tot);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        if (it[i]) {
            const int t = (int)(ex >> 32), o = (int)(unsigned)ex;
            a.tidx[rr[i]] = t;      // by root: what the nodes know
            a.noff[rr[i]] = o;
            a.offsets[t] = o;
        }
        ex += it[i];
    }
}

__global__ __launch_bounds__(TRK_NT) void k_trk_place(TrkArgs a)
{
    const int chunks = (a.stride + TRK_NT - 1) / TRK_NT;
    const int s = blockIdx.x / chunks, k = (blockIdx.x % chunks) * TRK_NT + threadIdx.x;
    const int f = fid_of(a, s);
    if ((unsigned)f >= (unsigned)a.n_frames) return;
    int c = a.counts[s];
    c = c > a.stride ? a.stride : c;
    if (k >= c) return;
    const int x = f * a.stride + k;
    const int r = a.root[x];
    if (a.flag[r]) { a.track_of[x] = -2; return; }
    const int sz = a.size[r];
    if (sz < a.min_len) return;   // track_of stays -1
    const int pos = atomicAdd(a.cursor + r, 1);
    a.tmp[a.noff[r] + pos] = x;
    a.track_of[x] = a.tidx[r];
}

__global__ __launch_bounds__(TRK_NT) void k_trk_rank(TrkArgs a)
{
    const int chunks = (a.stride + TRK_NT - 1) / TRK_NT;
    const int s = blockIdx.x / chunks, k = (blockIdx.x % chunks) * TRK_NT + threadIdx.x;
    const int f = fid_of(a, s);
    if ((unsigned)f >= (unsigned)a.n_frames) return;
    int c = a.counts[s];
    c = c > a.stride ? a.stride : c;
    if (k >= c) return;
    const int x = f * a.stride + k;
    const int r = a.root[x];
    const int sz = a.size[r];
    if (a.flag[r] || sz < a.min_len) return;
    const int o = a.noff[r];
    const int32_t *seg = a.tmp + o;
    int rank = 0;
    for (int i = 0; i < sz; i++) rank += seg[i] < x;
    a.nodes[2 * (size_t)(o + rank)] = f;
    a.nodes[2 * (size_t)(o + rank) + 1] = k;
}

} // namespace

size_t pgx_tracks_ws_bytes(int n_frames, int stride)
{
    const size_t N = (size_t)n_frames * stride;
    int T = 64;
    while (T < 2 * stride) T <<= 1;
    const size_t nb = (N + SCAN_ITEMS - 1) / SCAN_ITEMS;
    return 9 * N * 4 + (size_t)n_frames * T * 4 + nb * 8 + TRK_ECNT * 4 + 1024;
}

void pgx_launch_tracks(hipStream_t s, const pgx_pair *d_matches, const int32_t *d_counts, const int32_t *d_pairlist, int M, int F,
                       int stride, const int32_t *d_frame_ids, int n_frames, int max_dist, int min_len, void *ws,
                       int32_t *d_track_of, int32_t *d_offsets, int32_t *d_nodes, int32_t *d_summary)
{
    TrkArgs a;
    a.matches = d_matches; a.counts = d_counts; a.pairlist = d_pairlist; a.frame_ids = d_frame_ids;
    a.M = M; a.F = F; a.stride = stride; a.n_frames = n_frames; a.max_dist = max_dist; a.min_len = min_len < 1 ? 1 : min_len;
    a.N = (long long)n_frames * stride;
    a.T = 64;
    while (a.T < 2 * stride) a.T <<= 1;
    const size_t N = (size_t)a.N;
    int32_t *w = static_cast<int32_t *>(ws);
    a.parent = w; a.size = w + N; a.flag = w + 2 * N; a.cursor = w + 3 * N; a.tidx = w + 4 * N; a.noff = w + 5 * N; a.tmp = w + 6 * N;
    a.root = w + 7 * N;
    a.rep = w + 8 * N;
    a.table = reinterpret_cast<uint32_t *>(w + 9 * N);
    const size_t tab_end = (9 * N + (size_t)n_frames * a.T) * 4;
    a.bsum = reinterpret_cast<unsigned long long *>(static_cast<char *>(ws) + ((tab_end + 7) & ~(size_t)7));
    a.ecnt = reinterpret_cast<int32_t *>(a.bsum + (a.N + SCAN_ITEMS - 1) / SCAN_ITEMS);
    a.track_of = d_track_of; a.offsets = d_offsets; a.nodes = d_nodes; a.summary = d_summary;

    const int chunks = (stride + TRK_NT - 1) / TRK_NT;
    const long long init_items = a.N > (long long)n_frames * a.T ? a.N : (long long)n_frames * a.T;
    long long gi = (init_items + TRK_NT - 1) / TRK_NT;
    if (gi > 4096) gi = 4096;
    if (gi < 1) gi = 1;
    hipLaunchKernelGGL(k_trk_init, dim3((unsigned)gi), dim3(TRK_NT), 0, s, a);
    if (M > 0) {
        const long long n_items = (long long)M * chunks;
        const long long gu = n_items < TRK_UNION_GRID ? n_items : TRK_UNION_GRID;
        hipLaunchKernelGGL(k_trk_union, dim3((unsigned)gu), dim3(TRK_NT), 0, s, a, n_items);
    }
    if (F > 0) hipLaunchKernelGGL(k_trk_flatten, dim3((unsigned)((size_t)F * chunks)), dim3(TRK_NT), 0, s, a);
    const int nb = (int)((a.N + SCAN_ITEMS - 1) / SCAN_ITEMS);
    hipLaunchKernelGGL(k_trk_scan_reduce, dim3(nb), dim3(SCAN_NT), 0, s, a);
    hipLaunchKernelGGL(k_trk_scan_sums, dim3(1), dim3(SCAN_NT), 0, s, a, nb);
    hipLaunchKernelGGL(k_trk_scan_apply, dim3(nb), dim3(SCAN_NT), 0, s, a);
    if (F > 0) {
        hipLaunchKernelGGL(k_trk_place, dim3((unsigned)((size_t)F * chunks)), dim3(TRK_NT), 0, s, a);
        hipLaunchKernelGGL(k_trk_rank, dim3((unsigned)((size_t)F * chunks)), dim3(TRK_NT), 0, s, a);
    }
}
