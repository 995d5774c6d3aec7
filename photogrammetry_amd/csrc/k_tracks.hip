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
//   k_trk_union    one thread per match entry: lock-free union, the larger root hooks under the smaller (atomicCAS), so a
//                  component's root ends up its smallest node id whatever the schedule
//   k_trk_flatten  one thread per node: root (into its own array), component size (atomicAdd), and the frame-conflict test: the node's root goes
//                  into its FRAME's hash table (open addressing, atomicCAS); finding it there already = two keypoints of one
//                  frame in one component
//   k_trk_scan_*   exclusive scan over node ids of (kept roots, their sizes): track index and node offset in root order
//   k_trk_place    nodes into their track's segment (atomic cursor: unordered)
//   k_trk_rank     a node's final place = number of segment entries below it (a kept track has at most one node per frame, so
//                  segments are short: <= n_frames)
#include "pgx_internal.h"

namespace {

constexpr int TRK_NT = 256;          // threads per workgroup of the per-entry / per-node kernels
constexpr int SCAN_NT = 1024;        // scan kernels: 1024 threads x 4 items
constexpr int SCAN_ITEMS = 4 * SCAN_NT;
constexpr uint32_t TRK_EMPTY = 0xFFFFFFFFu;

struct TrkArgs {
    const pgx_pair *matches;   // [M][stride]
    const int32_t *counts;     // [F] by slot
    const int32_t *pairlist;   // [M][2] slots
    const int32_t *frame_ids;  // [F] slot -> global frame number, -1 = not part of this graph; nullptr = identity
    int M, F, stride, n_frames, max_dist, min_len;
    long long N;               // n_frames * stride
    int T;                     // hash table entries per frame (power of two >= 2 * stride)
    int32_t *parent, *root, *size, *flag, *cursor, *tidx, *noff, *tmp;
    uint32_t *table;           // [n_frames][T]
    unsigned long long *bsum;  // per scan block: (kept roots << 32) | their nodes
    int32_t *track_of, *offsets, *nodes, *summary;
};

__device__ __forceinline__ int ld(const int32_t *p)
{
    // a plain load the compiler may not cache in a register across the loops below (no cache-bypass bits at this scope)
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}
__device__ __forceinline__ void st(int32_t *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); }

// Root of x with intermediate pointer jumping.  Invariant: parent[v] <= v, and parent[v] only ever moves to an ancestor of v.
// A value read here may be STALE (another XCD's L2, this CU's L1): every earlier value of parent[v] is v itself or an ancestor,
// so a stale read can only name a node that is no longer a root -- the caller's atomicCAS then fails and returns the truth.
__device__ __forceinline__ int trk_find(int32_t *parent, int x)
{
    int curr = ld(parent + x);
    if (curr != x) {
        int prev = x, next;
        while (curr > (next = ld(parent + curr))) {
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
        a.size[x] = 0;
        a.flag[x] = 0;
        a.cursor[x] = 0;
        a.track_of[x] = -1;
    }
    const long long nt = (long long)a.n_frames * a.T;
    for (long long i = t0; i < nt; i += nthreads) a.table[i] = TRK_EMPTY;
    if (t0 < 8) a.summary[t0] = 0;
}

// grid: M * ceil(stride / TRK_NT) workgroups; workgroup -> (image pair, chunk of its list)
__global__ __launch_bounds__(TRK_NT) void k_trk_union(TrkArgs a)
{
    const int chunks = (a.stride + TRK_NT - 1) / TRK_NT;
    const int m = blockIdx.x / chunks, ch = blockIdx.x % chunks;
    const int sa = a.pairlist[2 * m], sb = a.pairlist[2 * m + 1];
    if ((unsigned)sa >= (unsigned)a.F || (unsigned)sb >= (unsigned)a.F) return;
    const int fa = fid_of(a, sa), fb = fid_of(a, sb);
    if ((unsigned)fa >= (unsigned)a.n_frames || (unsigned)fb >= (unsigned)a.n_frames) return;
    int ca = a.counts[sa], cb = a.counts[sb];
    ca = ca > a.stride ? a.stride : ca;
    cb = cb > a.stride ? a.stride : cb;
    const int e = ch * TRK_NT + threadIdx.x;
    bool edge = false;
    int u = 0, v = 0;
    if (e < ca) {
        const pgx_pair p = a.matches[(size_t)m * a.stride + e];
        // KeypointMatching.cs:40-42: the tail entries carry int.MaxValue and never link, whatever max_dist is
        edge = p.dist <= a.max_dist && p.dist != PGX_DIST_NONE && (unsigned)p.k1 < (unsigned)ca && (unsigned)p.k2 < (unsigned)cb;
        u = fa * a.stride + p.k1;
        v = fb * a.stride + p.k2;
    }
    const unsigned long long bal = __ballot(edge);
    if ((threadIdx.x & 63) == 0 && bal) atomicAdd(a.summary + 4, __popcll(bal));
    if (!edge) return;
    int ru = trk_find(a.parent, u), rv = trk_find(a.parent, v);
    while (ru != rv) {
        if (ru < rv) { const int t = ru; ru = rv; rv = t; }   // ru > rv: hook ru under rv
        const int old = atomicCAS(a.parent + ru, ru, rv);
        if (old == ru) break;
        ru = old;   // ru was no root any more: go on from its parent (smaller, so this ends)
    }
}

// grid: F * ceil(stride / TRK_NT); workgroup -> (slot, chunk of its keypoints)
__global__ __launch_bounds__(TRK_NT) void k_trk_flatten(TrkArgs a)
{
    const int chunks = (a.stride + TRK_NT - 1) / TRK_NT;
    const int s = blockIdx.x / chunks, k = (blockIdx.x % chunks) * TRK_NT + threadIdx.x;
    const int f = fid_of(a, s);
    if ((unsigned)f >= (unsigned)a.n_frames) return;
    int c = a.counts[s];
    c = c > a.stride ? a.stride : c;
    if (k >= c) return;
    const int x = f * a.stride + k;
    const int r = trk_find(a.parent, x);
    // into an array of its own: other threads' pointer jumping still stores (older) ancestors into parent[x] while this runs
    a.root[x] = r;
    atomicAdd(a.size + r, 1);
    // the root into this frame's table; already there = a second keypoint of this frame in the component
    uint32_t *tab = a.table + (size_t)f * a.T;
    uint32_t h = ((uint32_t)r * 2654435761u) >> 7;
    for (;;) {
        h &= (uint32_t)(a.T - 1);
        const uint32_t old = atomicCAS(tab + h, TRK_EMPTY, (uint32_t)r);
        if (old == TRK_EMPTY) break;
        if (old == (uint32_t)r) { a.flag[r] = 1; break; }
        h++;
    }
}

__device__ __forceinline__ unsigned long long kept_item(const TrkArgs &a, long long x)
{
    if (x >= a.N) return 0ull;
    const int sz = a.size[x];   // > 0 only at roots
    if (sz < a.min_len || sz <= 0 || a.flag[x]) return 0ull;
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
        const long long x = x0 + i;
        const unsigned long long it = kept_item(a, x);
        s += it;
        if (it) longest = max(longest, (int)(unsigned)it);
        if (x < a.N && a.flag[x]) {   // flags are only ever set at roots
            dropped++;
            dropped_nodes += a.size[x];
            largest_dropped = max(largest_dropped, a.size[x]);
        }
    }
    unsigned long long tot;
    (void)block_excl_scan(s, lds, &tot);
    if (threadIdx.x == 0) a.bsum[blockIdx.x] = tot;
    if (dropped) { atomicAdd(a.summary + 2, dropped); atomicAdd(a.summary + 3, dropped_nodes); atomicMax(a.summary + 6, largest_dropped); }
    if (longest) atomicMax(a.summary + 5, longest);
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
    if (threadIdx.x == 0) {
        const int nt = (int)(carry >> 32), nn = (int)(unsigned)carry;
        a.summary[0] = nt;
        a.summary[1] = nn;
        a.offsets[nt] = nn;
    }
}

__global__ __launch_bounds__(SCAN_NT) void k_trk_scan_apply(TrkArgs a)
{
    __shared__ unsigned long long lds[SCAN_NT / 64];
    const long long x0 = (long long)blockIdx.x * SCAN_ITEMS + (long long)threadIdx.x * 4;
    unsigned long long it[4], s = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) { it[i] = kept_item(a, x0 + i); s += it[i]; }
    unsigned long long tot;
    unsigned long long ex = a.bsum[blockIdx.x] + block_excl_scan(s, lds, &tot);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        if (it[i]) {
            const int t = (int)(ex >> 32), o = (int)(unsigned)ex;
            a.tidx[x0 + i] = t;
            a.noff[x0 + i] = o;
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
    return 8 * N * 4 + (size_t)n_frames * T * 4 + nb * 8 + 1024;
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
    a.table = reinterpret_cast<uint32_t *>(w + 8 * N);
    const size_t tab_end = (8 * N + (size_t)n_frames * a.T) * 4;
    a.bsum = reinterpret_cast<unsigned long long *>(static_cast<char *>(ws) + ((tab_end + 7) & ~(size_t)7));
    a.track_of = d_track_of; a.offsets = d_offsets; a.nodes = d_nodes; a.summary = d_summary;

    const int chunks = (stride + TRK_NT - 1) / TRK_NT;
    const long long init_items = a.N > (long long)n_frames * a.T ? a.N : (long long)n_frames * a.T;
    long long gi = (init_items + TRK_NT - 1) / TRK_NT;
    if (gi > 4096) gi = 4096;
    if (gi < 1) gi = 1;
    hipLaunchKernelGGL(k_trk_init, dim3((unsigned)gi), dim3(TRK_NT), 0, s, a);
    if (M > 0) hipLaunchKernelGGL(k_trk_union, dim3((unsigned)((size_t)M * chunks)), dim3(TRK_NT), 0, s, a);
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
