// k_match.hip -- KeypointMatching.MatchKeypoints on gfx950.
//
// Reference: ImageProcessing/KeypointMatching.cs:14-69.  dist[k1][k2] = popcount(d1 ^ d2)
// (:20-31, :71-82); then N1 times: take the first strict minimum over the still-available
// (k1, k2) in (k1 asc, k2 asc) order, emit it, retire its row and column (:38-66).  When the
// columns run out the loop keeps emitting (kp1[0], kp2[0], int.MaxValue) (:40-42, :57-62).
//
// Parallel formulation (equal to the literal loop under the strict total order
// (dist, k1, k2), checked against the oracle): per round every available row takes
// min_j (d, j), every available column takes min_i (d, i); an edge that is both is accepted
// and its row and column retire.  Accepted edges sorted by (dist, k1) are the reference's
// emission order.
//
// Data: keys are u32 (dist << 20 | index) so "min" carries the tie-break; per image pair the
// available rows / columns are kept as ascending index lists that are stably compacted after
// every round, so later rounds only evaluate what is left.
//
// Kernels
//   k_match_init      identity lists, keys, counters
//   k_ham_fp4         (256-bit descriptors) block-scaled FP4 MFMA distance + fused row/column argmin, k_match_mfma.inc
//   k_ham_valu        xor+popcount distance/argmin for any other descriptor length
//   k_match_select    one workgroup per image pair: accept mutual edges, compact the lists
//   256-bit descriptors, once a pair's residual is <= PGX_TAIL_MAX (k_match_mfma.inc, k_match_tail.inc):
//   k_tail_rows_fp4   residual distance rows (u8) + every row's nearest column, on the matrix pipe
//   k_match_gs        one workgroup per image pair: row-proposing deferred acceptance off a queue of free rows (no rounds),
//                     sort by (dist, k1), the N1 output entries
//   other descriptor lengths:
//   k_match_finish    one workgroup per image pair: remaining rounds in-kernel (LDS tail below PGX_TAIL_FILL_MAX),
//                     sort (bitonic, LDS), the N1 output entries
#include "pgx_internal.h"
#include <hip/hip_ext.h>

namespace {

constexpr int SEL_NT = 1024;
constexpr int CNT_N1 = 0, CNT_N2 = 1, CNT_NACC = 2, CNT_N1_ORIG = 3, CNT_N2_ORIG = 4, CNT_PARITY = 5, CNT_FILLED = 6, CNT_DONE = 7, CNT_WORDS = 8;

struct PairWs {
    uint32_t *rowkey, *colkey, *rows0, *rows1, *cols0, *cols1; // no runtime-indexed arrays: they would live in scratch
    int32_t *mk2, *md;
    int32_t *cnt;
    uint32_t *skeys;
    uint16_t *dcache;
};

__host__ __device__ inline size_t pow2_ge(size_t v) { size_t p = 1; while (p < v) p <<= 1; return p; }

// per image pair: 8 arrays of S words, counters, sort keys, and the tail's area:
//   256-bit descriptors: the residual's u8 distance matrix (R, C <= PGX_TAIL_MAX, rows on 128-byte lines; k_tail_rows_fp4 ->
//   k_match_gs); other lengths: the u16 matrices D and D^T of the LDS tail (<= PGX_TAIL_FILL_MAX^2 each) over the same bytes;
//   then gs_fallback's two free lists of S words each (any-size path).
constexpr size_t TAIL_MAT_BYTES = (size_t)PGX_TAIL_MAX * PGX_TAIL_MAX + 256;
static_assert((size_t)4 * PGX_TAIL_FILL_MAX * PGX_TAIL_FILL_MAX <= TAIL_MAT_BYTES, "D and D^T of the generic tail share the matrix area");
__host__ __device__ inline size_t dcache_words(int S) { return TAIL_MAT_BYTES / 4 + (size_t)2 * S; }
__host__ __device__ inline size_t pair_ws_words(int S)
{
    return (size_t)8 * S + CNT_WORDS + pow2_ge((size_t)(S > 1 ? S : 1)) + dcache_words(S);
}

__device__ __forceinline__ PairWs pair_ws(uint32_t *ws, int m, int S)
{
    uint32_t *b = ws + (size_t)m * pair_ws_words(S);
    PairWs p;
    p.rowkey = b; p.colkey = b + S;
    p.rows0 = b + 2 * (size_t)S; p.rows1 = b + 3 * (size_t)S;
    p.cols0 = b + 4 * (size_t)S; p.cols1 = b + 5 * (size_t)S;
    p.mk2 = reinterpret_cast<int32_t *>(b + 6 * (size_t)S);
    p.md = reinterpret_cast<int32_t *>(b + 7 * (size_t)S);
    p.cnt = reinterpret_cast<int32_t *>(b + 8 * (size_t)S);
    p.skeys = b + 8 * (size_t)S + CNT_WORDS;
    p.dcache = reinterpret_cast<uint16_t *>(p.skeys + pow2_ge((size_t)(S > 1 ? S : 1)));
    return p;
}

__global__ __launch_bounds__(256) void k_match_init(uint32_t *ws, const int32_t *__restrict__ counts,
                                                    const int32_t *__restrict__ pairlist, int S, int max_n,
                                                    int *status)
{
    const int m = blockIdx.y;
    PairWs p = pair_ws(ws, m, S);
    int n1 = counts[pairlist[2 * m]], n2 = counts[pairlist[2 * m + 1]];
    n1 = n1 < 0 ? 0 : (n1 > max_n ? max_n : n1); // lists are truncated to their first max_count entries
    n2 = n2 < 0 ? 0 : (n2 > max_n ? max_n : n2);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < S; i += gridDim.x * blockDim.x) {
        p.rowkey[i] = PGX_KEY_NONE;
        p.colkey[i] = PGX_KEY_NONE;
        p.rows0[i] = (uint32_t)i;
        p.cols0[i] = (uint32_t)i;
        p.mk2[i] = -1;
        p.md[i] = PGX_DIST_NONE;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        p.cnt[CNT_N1] = n1; p.cnt[CNT_N2] = n2; p.cnt[CNT_NACC] = 0;
        p.cnt[CNT_N1_ORIG] = n1; p.cnt[CNT_N2_ORIG] = n2; p.cnt[CNT_PARITY] = 0; p.cnt[CNT_FILLED] = 0; p.cnt[CNT_DONE] = 0;
        if (n1 > 0 && n2 == 0) atomicOr(status, (int)PGX_ST_EMPTY_SET); // KeypointMatching.cs:61
    }
}

// ---- xor + popcount distance / argmin -------------------------------------------------
// One thread per (compact) row; a block covers RB rows and walks a chunk of columns staged
// through LDS 256 at a time (all lanes read the same LDS address: broadcast, conflict-free).
template <int WORDS>
__device__ __forceinline__ void ham_rows_vs_cols(const uint32_t *__restrict__ rdesc, const uint32_t *__restrict__ rlist,
                                                 int nR, const uint32_t *__restrict__ cdesc,
                                                 const uint32_t *__restrict__ clist, int c_begin, int c_end,
                                                 int r_begin, int words_rt, uint32_t *outkey, bool use_atomic,
                                                 uint32_t *lds /* [256*(W+1)] */)
{
    const int W = WORDS > 0 ? WORDS : words_rt;
    const int nth = blockDim.x, tid = threadIdx.x;
    for (int rb = r_begin; rb < nR; rb += nth) {
        const int r = rb + tid;
        const bool valid = r < nR;
        const uint32_t i = valid ? rlist[r] : 0u;
        uint32_t a[WORDS > 0 ? WORDS : 1];
        if (WORDS > 0) {
#pragma unroll
            for (int w = 0; w < WORDS; w++) a[w] = valid ? rdesc[(size_t)i * WORDS + w] : 0u;
        }
        uint32_t best = PGX_KEY_NONE;
        for (int cs = c_begin; cs < c_end; cs += 256) {
            const int cn = (c_end - cs < 256) ? c_end - cs : 256;
            __syncthreads();
            for (int t = tid; t < cn; t += nth) {
                const uint32_t j = clist[cs + t];
                lds[256 * W + t] = j;
                for (int w = 0; w < W; w++) lds[t * W + w] = cdesc[(size_t)j * W + w];
            }
            __syncthreads();
            if (valid) {
                for (int k = 0; k < cn; k++) {
                    uint32_t d = 0;
                    if (WORDS > 0) {
#pragma unroll
                        for (int w = 0; w < WORDS; w++) d += __popc(a[w] ^ lds[k * WORDS + w]);
                    } else {
                        for (int w = 0; w < W; w++) d += __popc(rdesc[(size_t)i * W + w] ^ lds[k * W + w]);
                    }
                    const uint32_t key = (d << PGX_IDX_BITS) | lds[256 * W + k];
                    best = best < key ? best : key;
                }
            }
        }
        if (valid) {
            if (use_atomic) atomicMin(&outkey[i], best);
            else outkey[i] = best;
        }
        if (use_atomic) break; // multi-block launch: one row block per workgroup
    }
}

constexpr int VALU_CH = 512; // columns per block in the multi-block kernel

template <int WORDS>
__global__ __launch_bounds__(256) void k_ham_valu(uint32_t *ws, const uint32_t *__restrict__ desc,
                                                  const int32_t *__restrict__ pairlist, int S, int words, int skip_below)
{
    extern __shared__ uint32_t lds[];
    const int side = blockIdx.z & 1, m = blockIdx.z >> 1;
    PairWs p = pair_ws(ws, m, S);
    const int n1 = p.cnt[CNT_N1], n2 = p.cnt[CNT_N2], parity = p.cnt[CNT_PARITY];
    if (n1 <= 0 || n2 <= 0) return;
    if (n1 <= skip_below && n2 <= skip_below) return; // small enough: the per-pair tail kernel takes it from here
    const uint32_t *dA = desc + (size_t)pairlist[2 * m] * S * words;
    const uint32_t *dB = desc + (size_t)pairlist[2 * m + 1] * S * words;
    const int nR = side ? n2 : n1, nC = side ? n1 : n2;
    const int r_begin = blockIdx.x * 256, c_begin = blockIdx.y * VALU_CH;
    if (r_begin >= nR || c_begin >= nC) return;
    const int c_end = (c_begin + VALU_CH < nC) ? c_begin + VALU_CH : nC;
    ham_rows_vs_cols<WORDS>(side ? dB : dA, side ? (parity ? p.cols1 : p.cols0) : (parity ? p.rows1 : p.rows0), nR, side ? dA : dB,
                            side ? (parity ? p.rows1 : p.rows0) : (parity ? p.cols1 : p.cols0), c_begin, c_end, r_begin, words,
                            side ? p.colkey : p.rowkey, true, lds);
}

// ---- block-wide helpers ----------------------------------------------------------------
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t *excl, uint32_t *wsum)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = __shfl_up(incl, d);
        if (lane >= d) incl += t;
    }
    __syncthreads();
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    uint32_t woff = 0, total = 0;
    for (int w = 0; w < nw; w++) {
        uint32_t s = wsum[w];
        if (w < wv) woff += s;
        total += s;
    }
    *excl = woff + incl - v;
    return total;
}

// Accept mutual edges of the finished round and stably compact both lists (cur -> nxt).
// Called by every thread of one workgroup; ends with the new counts in p.cnt and a barrier.
__device__ void select_compact_wg(PairWs p, int parity, uint32_t *wsum)
{
    const int tid = threadIdx.x, nth = blockDim.x;
    const int n1 = p.cnt[CNT_N1], n2 = p.cnt[CNT_N2];
    const uint32_t *rows = (parity ? p.rows1 : p.rows0), *cols = (parity ? p.cols1 : p.cols0);
    uint32_t *nrows = (parity ? p.rows0 : p.rows1), *ncols = (parity ? p.cols0 : p.cols1);
    __syncthreads();
    // accept: row i's best column j whose best row is i
    for (int r = tid; r < n1; r += nth) {
        const uint32_t i = rows[r];
        const uint32_t rk = p.rowkey[i];
        const uint32_t j = rk & PGX_IDX_MASK;
        const uint32_t ck = p.colkey[j];
        if (rk != PGX_KEY_NONE && ck != PGX_KEY_NONE && (ck & PGX_IDX_MASK) == i) {
            p.mk2[i] = (int32_t)j;
            p.md[i] = (int32_t)(rk >> PGX_IDX_BITS);
            p.rowkey[i] = 0; // retired marker (a live key is never 0 after the reset below)
        }
    }
    __syncthreads();
    // a column retires iff its best row accepted it (then that row's mk2 points back at it)
    // stable compaction of rows
    {
        const int per = (n1 + nth - 1) / nth;
        const int b = tid * per, e = (b + per < n1) ? b + per : n1;
        uint32_t c = 0;
        for (int r = b; r < e; r++) c += (p.mk2[rows[r]] < 0) ? 1u : 0u;
        uint32_t ex;
        const uint32_t tot = block_excl_scan(c, &ex, wsum);
        uint32_t o = ex;
        for (int r = b; r < e; r++) {
            const uint32_t i = rows[r];
            if (p.mk2[i] < 0) { nrows[o++] = i; p.rowkey[i] = PGX_KEY_NONE; }
        }
        __syncthreads();
        if (tid == 0) { p.cnt[CNT_NACC] += n1 - (int)tot; p.cnt[CNT_N1] = (int)tot; }
    }
    {
        const int per = (n2 + nth - 1) / nth;
        const int b = tid * per, e = (b + per < n2) ? b + per : n2;
        uint32_t c = 0;
        auto col_free = [&](uint32_t j) {
            const uint32_t ck = p.colkey[j];
            if (ck == PGX_KEY_NONE) return true;
            const uint32_t i = ck & PGX_IDX_MASK; // best row of column j this round
            return !(p.mk2[i] == (int32_t)j);
        };
        for (int q = b; q < e; q++) c += col_free(cols[q]) ? 1u : 0u;
        uint32_t ex;
        const uint32_t tot = block_excl_scan(c, &ex, wsum);
        uint32_t o = ex;
        for (int q = b; q < e; q++) {
            const uint32_t j = cols[q];
            if (col_free(j)) ncols[o++] = j;
        }
        __syncthreads();
        // reset the surviving columns' keys only after every thread evaluated col_free
        for (uint32_t q = tid; q < tot; q += nth) p.colkey[ncols[q]] = PGX_KEY_NONE;
        if (tid == 0) { p.cnt[CNT_N2] = (int)tot; p.cnt[CNT_PARITY] = parity ^ 1; }
    }
    __syncthreads();
}

// The same for lists of at most PER * blockDim entries (every bench-sized round): a thread owns PER consecutive list places
// through all phases and keeps what it loaded -- rows[r], the accept decision, cols[q], col_free -- in registers, so each phase
// is ONE chain of dependent loads (list -> key -> partner's key; the general form above walks it three times: accept, count,
// write) and the accept and row-compaction phases share theirs.  37 -> about 20 us per image pair of 4096 x 4096.
template <int PER>
__device__ void select_compact_wg_small(PairWs p, int parity, uint32_t *wsum)
{
    const int tid = threadIdx.x;
    const int n1 = p.cnt[CNT_N1], n2 = p.cnt[CNT_N2];
    const uint32_t *rows = (parity ? p.rows1 : p.rows0), *cols = (parity ? p.cols1 : p.cols0);
    uint32_t *nrows = (parity ? p.rows0 : p.rows1), *ncols = (parity ? p.cols0 : p.cols1);
    __syncthreads();
    {   // accept (row i's best column j whose best row is i) and compact the rows that stay
        uint32_t ii[PER], rk[PER], ck[PER];
        bool in[PER], acc[PER];
#pragma unroll
        for (int u = 0; u < PER; u++) { const int r = tid * PER + u; in[u] = r < n1; ii[u] = in[u] ? rows[r] : 0u; }
#pragma unroll
        for (int u = 0; u < PER; u++) rk[u] = in[u] ? p.rowkey[ii[u]] : PGX_KEY_NONE;
#pragma unroll
        for (int u = 0; u < PER; u++) ck[u] = rk[u] != PGX_KEY_NONE ? p.colkey[rk[u] & PGX_IDX_MASK] : PGX_KEY_NONE;
        uint32_t c = 0;
#pragma unroll
        for (int u = 0; u < PER; u++) {
            acc[u] = rk[u] != PGX_KEY_NONE && ck[u] != PGX_KEY_NONE && (ck[u] & PGX_IDX_MASK) == ii[u];
            if (acc[u]) {
                p.mk2[ii[u]] = (int32_t)(rk[u] & PGX_IDX_MASK);
                p.md[ii[u]] = (int32_t)(rk[u] >> PGX_IDX_BITS);
                p.rowkey[ii[u]] = 0; // retired marker
            }
            c += (in[u] && !acc[u]) ? 1u : 0u;
        }
        uint32_t ex;
        const uint32_t tot = block_excl_scan(c, &ex, wsum); // its barriers also publish mk2 to the column phase below
        uint32_t o = ex;
#pragma unroll
        for (int u = 0; u < PER; u++)
            if (in[u] && !acc[u]) { nrows[o++] = ii[u]; p.rowkey[ii[u]] = PGX_KEY_NONE; }
        if (tid == 0) { p.cnt[CNT_NACC] += n1 - (int)tot; p.cnt[CNT_N1] = (int)tot; }
    }
    __syncthreads();
    {   // a column retires iff its best row accepted it (then that row's mk2 points back at it)
        uint32_t jj[PER], ck[PER];
        int32_t mk[PER];
        bool in[PER], fr[PER];
#pragma unroll
        for (int u = 0; u < PER; u++) { const int q = tid * PER + u; in[u] = q < n2; jj[u] = in[u] ? cols[q] : 0u; }
#pragma unroll
        for (int u = 0; u < PER; u++) ck[u] = in[u] ? p.colkey[jj[u]] : PGX_KEY_NONE;
#pragma unroll
        for (int u = 0; u < PER; u++) mk[u] = ck[u] != PGX_KEY_NONE ? p.mk2[ck[u] & PGX_IDX_MASK] : -1;
        uint32_t c = 0;
#pragma unroll
        for (int u = 0; u < PER; u++) {
            fr[u] = in[u] && (ck[u] == PGX_KEY_NONE || mk[u] != (int32_t)jj[u]);
            c += fr[u] ? 1u : 0u;
        }
        uint32_t ex;
        const uint32_t tot = block_excl_scan(c, &ex, wsum); // every thread has evaluated its columns before anyone resets a key
        uint32_t o = ex;
#pragma unroll
        for (int u = 0; u < PER; u++)
            if (fr[u]) { ncols[o++] = jj[u]; p.colkey[jj[u]] = PGX_KEY_NONE; }
        if (tid == 0) { p.cnt[CNT_N2] = (int)tot; p.cnt[CNT_PARITY] = parity ^ 1; }
    }
    __syncthreads();
}

__global__ __launch_bounds__(SEL_NT) void k_match_select(uint32_t *ws, int S, unsigned long long *evals, int skip_below)
{
    __shared__ uint32_t wsum[SEL_NT / 64];
    PairWs p = pair_ws(ws, blockIdx.x, S);
    if (p.cnt[CNT_N1] <= 0 || p.cnt[CNT_N2] <= 0) return;
    if (p.cnt[CNT_N1] <= skip_below && p.cnt[CNT_N2] <= skip_below) return; // round was skipped
    const int parity = p.cnt[CNT_PARITY];
    if (threadIdx.x == 0) atomicAdd(evals, (unsigned long long)p.cnt[CNT_N1] * (unsigned long long)p.cnt[CNT_N2]);
    const int nmax = p.cnt[CNT_N1] > p.cnt[CNT_N2] ? p.cnt[CNT_N1] : p.cnt[CNT_N2]; // read before the barrier inside: thread 0 rewrites the counts at the end
    if (nmax <= 4 * SEL_NT) select_compact_wg_small<4>(p, parity, wsum);
    else select_compact_wg(p, parity, wsum);
}

// bitonic sort of `n2p` (power of two) u32 keys, ascending, by the whole workgroup.  The sort is bound by LDS traffic
// (one read and one write of every key per stage, 78 stages for 4096 keys: timed inside k_match_gs at 39 us with 512
// threads), so up to three consecutive stages j = 4s, 2s, s of a merge are fused: a thread takes the 8 keys whose indices
// differ in the bits s, 2s, 4s, runs the three compare-exchange layers in registers and writes them back -- 30 passes
// over the keys instead of 78.  The direction bit k lies above all three, so the 8 keys of a thread share it.
__device__ __forceinline__ void bitonic_sort_wg(uint32_t *keys, uint32_t n2p)
{
    const uint32_t tid = threadIdx.x, nth = blockDim.x;
    auto cx = [](uint32_t &a, uint32_t &b, bool up) {
        const uint32_t mn = a < b ? a : b, mx = a < b ? b : a;
        a = up ? mn : mx;
        b = up ? mx : mn;
    };
    for (uint32_t k = 2; k <= n2p; k <<= 1) {
        uint32_t j = k >> 1;
        while (j > 0) {
            __syncthreads();
            if (j >= 4) { // stages j, j/2, j/4 together; s = j/4
                const uint32_t s4 = j >> 2;
                for (uint32_t t = tid; t < n2p / 8; t += nth) {
                    // spread t around three zero bits at s, 2s, 4s
                    const uint32_t base = ((t & ~(s4 - 1)) << 3) | (t & (s4 - 1));
                    const bool up = (base & k) == 0;
                    uint32_t v[8];
#pragma unroll
                    for (int q = 0; q < 8; q++) v[q] = keys[base + (uint32_t)q * s4];
#pragma unroll
                    for (int q = 0; q < 4; q++) cx(v[q], v[q + 4], up);                       // partner distance 4s = j
#pragma unroll
                    for (int q = 0; q < 8; q++) if ((q & 2) == 0) cx(v[q], v[q + 2], up);    // 2s
#pragma unroll
                    for (int q = 0; q < 8; q += 2) cx(v[q], v[q + 1], up);                   // s
#pragma unroll
                    for (int q = 0; q < 8; q++) keys[base + (uint32_t)q * s4] = v[q];
                }
                j >>= 3;
            } else { // the last one or two stages of a merge (j = 2, 1 or j = 1), or k <= 4
                const uint32_t jj = j; // stages jj .. 1 on 2 * jj consecutive keys per group
                for (uint32_t t = tid; t < n2p / (2 * jj); t += nth) {
                    const uint32_t base = t * 2 * jj;
                    const bool up = (base & k) == 0;
                    if (jj == 2) {
                        uint32_t v0 = keys[base], v1 = keys[base + 1], v2 = keys[base + 2], v3 = keys[base + 3];
                        cx(v0, v2, up); cx(v1, v3, up); cx(v0, v1, up); cx(v2, v3, up);
                        keys[base] = v0; keys[base + 1] = v1; keys[base + 2] = v2; keys[base + 3] = v3;
                    } else {
                        uint32_t v0 = keys[base], v1 = keys[base + 1];
                        cx(v0, v1, up);
                        keys[base] = v0; keys[base + 1] = v1;
                    }
                }
                j = 0;
            }
        }
    }
    __syncthreads();
}

// The N1 output entries of one image pair: accepted edges sorted by (dist, k1), then the (0,0,int.MaxValue) tail
// (KeypointMatching.cs:38-66).  Inlined separately for sort keys in LDS and in the workspace: with ONE call on a pointer
// selected at run time the key accesses are flat_* instructions (LDS reached through the vector memory path); with two
// call sites each copy knows its address space and the LDS one sorts with ds_* instructions.
__device__ __forceinline__ void sort_and_emit(uint32_t *keys, uint32_t n2p, const PairWs &p, int n1o, int nacc, pgx_pair *__restrict__ o)
{
    const int tid = threadIdx.x, nth = blockDim.x;
    for (uint32_t i = tid; i < n2p; i += nth) {
        uint32_t k = PGX_KEY_NONE;
        if ((int)i < n1o && p.mk2[i] >= 0) k = ((uint32_t)p.md[i] << PGX_IDX_BITS) | i;
        keys[i] = k;
    }
    bitonic_sort_wg(keys, n2p);
    for (int e = tid; e < n1o; e += nth) {
        pgx_pair r;
        if (e < nacc) {
            const uint32_t k = keys[e];
            const uint32_t i = k & PGX_IDX_MASK;
            r.k1 = (int32_t)i; r.k2 = p.mk2[i]; r.dist = (int32_t)(k >> PGX_IDX_BITS);
        } else {
            r.k1 = 0; r.k2 = 0; r.dist = PGX_DIST_NONE;
        }
        o[e] = r;
    }
}

// ---- LDS-resident tail ---------------------------------------------------------------------
// When what is left of an image pair fits (<= TAIL_MAX rows and columns) the remaining rounds
// run out of LDS: residual descriptors, ascending index lists, best keys (local indices) and
// alive flags.  A row's best column stays valid until that column retires, so after a round only
// the rows (columns) whose best partner just retired are re-evaluated -- one wavefront per such
// row, lanes striding the alive columns, wave-wide min.  Tie-heavy data (hub descriptors, equal
// descriptors) needs many rounds of one or two acceptances; this keeps each of them ~1 us.
constexpr int TAIL_MAX = PGX_TAIL_MAX;

__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t t = __shfl_xor(v, o);
        v = t < v ? t : v;
    }
    return v;
}

// the same for a wavefront whose 64 lanes are ALL active: six DPP row operations and one v_readlane instead of six
// LDS-crossbar shuffles (each a ~60-cycle round trip); the result is wave-uniform
__device__ __forceinline__ uint32_t wave_min_u32_full(uint32_t v)
{
    auto mn = [](uint32_t a, uint32_t b) { return a < b ? a : b; };
    const int id = -1;
    v = mn(v, (uint32_t)__builtin_amdgcn_update_dpp(id, (int)v, 0xB1, 0xF, 0xF, false));  // quad_perm [1,0,3,2]
    v = mn(v, (uint32_t)__builtin_amdgcn_update_dpp(id, (int)v, 0x4E, 0xF, 0xF, false));  // quad_perm [2,3,0,1]
    v = mn(v, (uint32_t)__builtin_amdgcn_update_dpp(id, (int)v, 0x141, 0xF, 0xF, false)); // row_half_mirror
    v = mn(v, (uint32_t)__builtin_amdgcn_update_dpp(id, (int)v, 0x140, 0xF, 0xF, false)); // row_mirror
    v = mn(v, (uint32_t)__builtin_amdgcn_update_dpp(id, (int)v, 0x142, 0xA, 0xF, false)); // row_bcast:15 into rows 1, 3
    v = mn(v, (uint32_t)__builtin_amdgcn_update_dpp(id, (int)v, 0x143, 0xC, 0xF, false)); // row_bcast:31 into rows 2, 3
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

__host__ __device__ inline size_t tail_lds_words(int W)
{
    // rl, cl, rbest, cbest, rdl, cdl (6 x TAIL_MAX) + alive flags (2 x TAIL_MAX bytes) + 8 counters + descriptors
    return (size_t)6 * TAIL_MAX + (size_t)2 * TAIL_MAX / 4 + 8 + (size_t)2 * PGX_TAIL_FILL_MAX * W;
}

template <int WORDS>
__device__ void tail_rounds_lds(PairWs p, int parity, const uint32_t *__restrict__ dA, const uint32_t *__restrict__ dB,
                                int words_rt, uint32_t *lds, unsigned long long *dbg)
{
    const int W = WORDS > 0 ? WORDS : words_rt;
    const int tid = threadIdx.x, nth = blockDim.x, lane = tid & 63, wv = tid >> 6, nw = nth >> 6;
    const int R = p.cnt[CNT_N1], C = p.cnt[CNT_N2];
    const int Cs = (C + 7) & ~7, Rs = (R + 7) & ~7;       // row strides of the two cached matrices (16-B rows)
    uint16_t *D = p.dcache;                                // D[i][j], i < R, j < Cs
    uint16_t *DT = p.dcache + (size_t)PGX_TAIL_FILL_MAX * PGX_TAIL_FILL_MAX; // DT[j][i]
    uint32_t *rl = lds, *cl = rl + TAIL_MAX, *rbest = cl + TAIL_MAX, *cbest = rbest + TAIL_MAX;
    uint32_t *rdl = cbest + TAIL_MAX, *cdl = rdl + TAIL_MAX;
    uint8_t *ralive = reinterpret_cast<uint8_t *>(cdl + TAIL_MAX), *calive = ralive + TAIL_MAX;
    uint32_t *ctr = reinterpret_cast<uint32_t *>(calive + TAIL_MAX); // [0] dirty rows [1] dirty cols [2] accepted [3] alive rows [4] alive cols
    uint32_t *rdesc = ctr + 8, *cdesc = rdesc + (size_t)PGX_TAIL_FILL_MAX * W; // only used when this kernel fills D itself
    const uint32_t *rows = (parity ? p.rows1 : p.rows0), *cols = (parity ? p.cols1 : p.cols0);

    __syncthreads();
    for (int i = tid; i < TAIL_MAX; i += nth) { ralive[i] = 0; calive[i] = 0; }
    __syncthreads();
    for (int i = tid; i < R; i += nth) { rl[i] = rows[i]; rdl[i] = (uint32_t)i; ralive[i] = 1; rbest[i] = PGX_KEY_NONE; }
    for (int j = tid; j < C; j += nth) { cl[j] = cols[j]; cdl[j] = (uint32_t)j; calive[j] = 1; cbest[j] = PGX_KEY_NONE; }
    // word-major images (desc[w][k]) so that lanes reading neighbouring descriptors hit neighbouring banks
    for (int t = tid; t < R * W; t += nth) rdesc[(size_t)(t % W) * PGX_TAIL_FILL_MAX + t / W] = dA[(size_t)rows[t / W] * W + (t % W)];
    for (int t = tid; t < C * W; t += nth) cdesc[(size_t)(t % W) * PGX_TAIL_FILL_MAX + t / W] = dB[(size_t)cols[t / W] * W + (t % W)];
    if (tid == 0) { ctr[0] = (uint32_t)R; ctr[1] = (uint32_t)C; ctr[2] = 0; ctr[3] = 0; ctr[4] = 0; }
    __syncthreads();

    // one pass of xor+popcount fills both cached matrices (each entry computed twice so that
    // both are written with coalesced rows) and the first bests
    // (two adjacent entries per lane: one ds_read_b64 per word, one dword store per lane)
    auto fill = [&](const uint32_t *xdesc, int nx, const uint32_t *ydesc, int ny, int ystride, uint16_t *mat, uint32_t *bestout) {
        for (int i = wv; i < nx; i += nw) {
            uint32_t best = PGX_KEY_NONE;
            for (int j2 = lane * 2; j2 < ystride; j2 += 128) {
                uint32_t d0 = 0, d1 = 0;
                if (WORDS > 0) {
#pragma unroll
                    for (int w = 0; w < WORDS; w++) {
                        const uint32_t a = xdesc[(size_t)w * PGX_TAIL_FILL_MAX + i];
                        const uint2 b = *reinterpret_cast<const uint2 *>(ydesc + (size_t)w * PGX_TAIL_FILL_MAX + j2);
                        d0 += __popc(a ^ b.x);
                        d1 += __popc(a ^ b.y);
                    }
                } else {
                    for (int w = 0; w < W; w++) {
                        const uint32_t a = xdesc[(size_t)w * PGX_TAIL_FILL_MAX + i];
                        const uint2 b = *reinterpret_cast<const uint2 *>(ydesc + (size_t)w * PGX_TAIL_FILL_MAX + j2);
                        d0 += __popc(a ^ b.x);
                        d1 += __popc(a ^ b.y);
                    }
                }
                if (j2 < ny) {
                    const uint32_t key = (d0 << PGX_IDX_BITS) | (uint32_t)j2;
                    best = key < best ? key : best;
                } else d0 = 0xFFFFu;
                if (j2 + 1 < ny) {
                    const uint32_t key = (d1 << PGX_IDX_BITS) | (uint32_t)(j2 + 1);
                    best = key < best ? key : best;
                } else d1 = 0xFFFFu;
                *reinterpret_cast<uint32_t *>(mat + (size_t)i * ystride + j2) = d0 | (d1 << 16);
            }
            best = wave_min_u32(best);
            if (lane == 0) bestout[i] = best;
        }
    };
    fill(rdesc, R, cdesc, C, Cs, D, rbest);
    fill(cdesc, C, rdesc, R, Rs, DT, cbest);
    if (tid == 0) { ctr[0] = 0; ctr[1] = 0; }
    __syncthreads();

    // min over the alive entries of cached rows, 8 distances per 16-B load; a wavefront takes
    // four dirty rows at a time and issues all their loads before using any (the scan is bound
    // by L2 latency, not bandwidth)
    // Dirty rows (scanned in D against the alive columns) and dirty columns (in DT against the alive rows) go
    // through TOGETHER: a wavefront takes four of each and issues all sixteen 16-B loads before using any,
    // so a round costs one L2 round trip, not one per side.
    constexpr int NE = 3; // rows and columns per wavefront and pass (register budget: 128 at 1024 threads)
    auto scan_dirty = [&](int nrd, int ncd) {
        const int nmax = nrd > ncd ? nrd : ncd;
        for (int k0 = wv * NE; k0 < nmax; k0 += nw * NE) {
            int ir[NE], ic[NE];
            uint32_t br[NE], bc[NE];
#pragma unroll
            for (int u = 0; u < NE; u++) {
                ir[u] = (k0 + u < nrd) ? (int)rdl[k0 + u] : -1;
                ic[u] = (k0 + u < ncd) ? (int)cdl[k0 + u] : -1;
                br[u] = PGX_KEY_NONE; bc[u] = PGX_KEY_NONE;
            }
            const int nspan = C > R ? C : R;
            for (int q0 = 0; q0 < nspan; q0 += 1024) { // two 512-entry chunks per side in flight
                uint4 vr[NE][2], vc[NE][2];
                uint2 alr[2], alc[2];
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    const int j8 = q0 + lane * 8 + q * 512;
#pragma unroll
                    for (int u = 0; u < NE; u++) {
                        vr[u][q] = (ir[u] >= 0 && j8 < C) ? *reinterpret_cast<const uint4 *>(D + (size_t)ir[u] * Cs + j8) : make_uint4(~0u, ~0u, ~0u, ~0u);
                        vc[u][q] = (ic[u] >= 0 && j8 < R) ? *reinterpret_cast<const uint4 *>(DT + (size_t)ic[u] * Rs + j8) : make_uint4(~0u, ~0u, ~0u, ~0u);
                    }
                    alr[q] = (j8 < C) ? *reinterpret_cast<const uint2 *>(calive + j8) : make_uint2(0, 0);
                    alc[q] = (j8 < R) ? *reinterpret_cast<const uint2 *>(ralive + j8) : make_uint2(0, 0);
                }
                auto reduce = [&](const uint4 (&v)[NE][2], const uint2 (&al)[2], uint32_t (&best)[NE]) {
#pragma unroll
                    for (int u = 0; u < NE; u++) {
#pragma unroll
                        for (int q = 0; q < 2; q++) {
                            const int j8 = q0 + lane * 8 + q * 512;
                            const uint32_t dw[4] = {v[u][q].x, v[u][q].y, v[u][q].z, v[u][q].w};
#pragma unroll
                            for (int k = 0; k < 8; k++) {
                                const uint32_t d = (dw[k >> 1] >> (16 * (k & 1))) & 0xFFFFu;
                                const uint32_t a = ((k < 4 ? al[q].x : al[q].y) >> (8 * (k & 3))) & 0xFFu;
                                const uint32_t key = a ? ((d << PGX_IDX_BITS) | (uint32_t)(j8 + k)) : PGX_KEY_NONE;
                                best[u] = key < best[u] ? key : best[u];
                            }
                        }
                    }
                };
                reduce(vr, alr, br);
                reduce(vc, alc, bc);
            }
#pragma unroll
            for (int u = 0; u < NE; u++) {
                const uint32_t b1 = wave_min_u32(br[u]), b2 = wave_min_u32(bc[u]);
                if (lane == 0 && ir[u] >= 0) rbest[ir[u]] = b1;
                if (lane == 0 && ic[u] >= 0) cbest[ic[u]] = b2;
            }
        }
    };

    while (true) {
        const int nrd = (int)ctr[0], ncd = (int)ctr[1];
        if (tid == 0) { atomicAdd(&dbg[3], 1ull); atomicAdd(&dbg[4], (unsigned long long)nrd); atomicAdd(&dbg[5], (unsigned long long)ncd); }
        scan_dirty(nrd, ncd);
        __syncthreads();
        if (tid == 0) { ctr[0] = 0; ctr[1] = 0; ctr[3] = 0; ctr[4] = 0; }
        // accept mutual edges (every alive row points at an alive column here)
        for (int i = tid; i < R; i += nth) {
            if (!ralive[i]) continue;
            const uint32_t rk = rbest[i];
            const uint32_t j = rk & PGX_IDX_MASK;
            if ((cbest[j] & PGX_IDX_MASK) == (uint32_t)i) {
                const uint32_t oi = rl[i];
                p.mk2[oi] = (int32_t)cl[j];
                p.md[oi] = (int32_t)(rk >> PGX_IDX_BITS);
                ralive[i] = 0;
                calive[j] = 0;
                atomicAdd(&ctr[2], 1u);
            }
        }
        __syncthreads();
        // whoever lost its partner must look again
        for (int i = tid; i < R; i += nth) {
            if (!ralive[i]) continue;
            atomicAdd(&ctr[3], 1u);
            if (!calive[rbest[i] & PGX_IDX_MASK]) rdl[atomicAdd(&ctr[0], 1u)] = (uint32_t)i;
        }
        for (int j = tid; j < C; j += nth) {
            if (!calive[j]) continue;
            atomicAdd(&ctr[4], 1u);
            if (!ralive[cbest[j] & PGX_IDX_MASK]) cdl[atomicAdd(&ctr[1], 1u)] = (uint32_t)j;
        }
        __syncthreads();
        if (ctr[3] == 0 || ctr[4] == 0) break; // uniform
    }
    if (tid == 0) {
        p.cnt[CNT_NACC] += (int)ctr[2];
        p.cnt[CNT_N1] = (int)ctr[3];
        p.cnt[CNT_N2] = (int)ctr[4];
    }
    __syncthreads();
}

template <int WORDS>
__global__ __launch_bounds__(SEL_NT) void k_match_finish(uint32_t *ws, const uint32_t *__restrict__ desc,
                                                         const int32_t *__restrict__ pairlist, int S, int words,
                                                         pgx_pair *__restrict__ out, uint32_t lds_keys_cap,
                                                         int tail_in_lds, int *status)
{
    extern __shared__ uint32_t lds[];
    __shared__ uint32_t wsum[SEL_NT / 64];
    const int m = blockIdx.x;
    PairWs p = pair_ws(ws, m, S);
    if (p.cnt[CNT_DONE]) return; // finished by k_match_tail
    int parity = p.cnt[CNT_PARITY];
    const uint32_t *dA = desc + (size_t)pairlist[2 * m] * S * words;
    const uint32_t *dB = desc + (size_t)pairlist[2 * m + 1] * S * words;

    // rounds on the global lists while the residual is too large for LDS
    while (true) {
        const int n1 = p.cnt[CNT_N1], n2 = p.cnt[CNT_N2];
        if (n1 <= 0 || n2 <= 0) break; // uniform: cnt is only written behind barriers
        const int lim = PGX_TAIL_FILL_MAX; // the descriptors of the residual must fit LDS
        if (tail_in_lds && n1 <= lim && n2 <= lim) {
            tail_rounds_lds<WORDS>(p, parity, dA, dB, words, lds, pgx_dbg(status));
            break;
        }
        ham_rows_vs_cols<WORDS>(dA, (parity ? p.rows1 : p.rows0), n1, dB, (parity ? p.cols1 : p.cols0), 0, n2, 0, words, p.rowkey, false, lds);
        ham_rows_vs_cols<WORDS>(dB, (parity ? p.cols1 : p.cols0), n2, dA, (parity ? p.rows1 : p.rows0), 0, n1, 0, words, p.colkey, false, lds);
        select_compact_wg(p, parity, wsum);
        parity ^= 1;
    }
    __syncthreads();

    // emit: accepted edges sorted by (dist, k1), then the (0,0,int.MaxValue) tail
    const int n1o = p.cnt[CNT_N1_ORIG];
    const int nacc = p.cnt[CNT_NACC];
    pgx_pair *o = out + (size_t)m * S;
    const uint32_t n2p = (uint32_t)pow2_ge((size_t)(n1o > 1 ? n1o : 1));
    if (n2p <= lds_keys_cap) sort_and_emit(lds, n2p, p, n1o, nacc, o);
    else sort_and_emit(p.skeys, n2p, p, n1o, nacc, o);
}

} // namespace

// The residual byte matrix of an image pair (k_tail_rows* write it, k_match_gs reads it): rows start on 128-byte lines.
// The rows kernel is bound by its HBM writes, and a store instruction of it covers 16 rows x 64 bytes: with rows on line
// boundaries the two halves of a line come from neighbouring wavefronts of one workgroup and leave L2 as whole lines
// (tools/probe/write_pattern.hip: 120 -> 96 us for 385 MB in that pattern).
__device__ __forceinline__ int tail_row_stride(int C) { return (C + 127) & ~127; }
__device__ __forceinline__ uint8_t *tail_matrix(const PairWs &p)
{
    // pointer + offset (not an integer round trip): the result stays a global-memory pointer, so its loads and stores are
    // global_* instructions rather than flat_* ones (which also count against the LDS wait counter)
    uint8_t *b = reinterpret_cast<uint8_t *>(p.dcache);
    return b + ((128 - (reinterpret_cast<uintptr_t>(b) & 127)) & 127);
}

#include "k_match_tail.inc"
#include "k_match_mfma.inc"

// the M per-pair workspaces, then the order in which the per-pair finish takes the pairs (k_match_order)
size_t pgx_match_ws_bytes(int M, int stride) { return ((size_t)M * pair_ws_words(stride) + (size_t)M) * 4; }

template <int WORDS>
static void launch_rounds_valu(hipStream_t s, uint32_t *ws, const uint32_t *desc, const int32_t *pairlist,
                               const MatchPlan &plan)
{
    const int W = WORDS > 0 ? WORDS : plan.words;
    dim3 grid((plan.max_n + 255) / 256, (plan.max_n + VALU_CH - 1) / VALU_CH, plan.M * 2);
    const size_t shm = (size_t)256 * (W + 1) * 4;
    hipLaunchKernelGGL((k_ham_valu<WORDS>), grid, dim3(256), shm, s, ws, desc, pairlist, plan.stride, plan.words, plan.skip_below);
}

template <int WORDS>
static void launch_finish(hipStream_t s, uint32_t *ws, const uint32_t *desc, const int32_t *pairlist,
                          const MatchPlan &plan, pgx_pair *out, int *status)
{
    const int W = WORDS > 0 ? WORDS : plan.words;
    const size_t ham_words = (size_t)256 * (W + 1);
    const size_t n2p = pow2_ge((size_t)(plan.max_n > 1 ? plan.max_n : 1));
    const size_t key_cap = n2p <= 32768 ? n2p : 0; // <= 128 KiB of LDS for the sort
    const size_t tail_words = tail_lds_words(W);
    const int tail_in_lds = tail_words * 4 <= 140 * 1024 ? 1 : 0;
    size_t shm_words = ham_words > key_cap ? ham_words : key_cap;
    if (tail_in_lds && tail_words > shm_words) shm_words = tail_words;
    static bool attr_set = false; // per template instance: allow the largest carve once
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_match_finish<WORDS>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL((k_match_finish<WORDS>), dim3(plan.M), dim3(SEL_NT), shm_words * 4, s, ws, desc, pairlist,
                       plan.stride, plan.words, out, (uint32_t)key_cap, tail_in_lds, status);
}

void pgx_launch_match_wide(pgx_ctx *ctx, hipStream_t s, const uint32_t *d_desc, const int32_t *d_counts,
                           const int32_t *d_pairlist, const MatchPlan &plan, void *wsv, int *status, hipEvent_t gate)
{
    if (plan.M <= 0) return;
    uint32_t *ws = reinterpret_cast<uint32_t *>(wsv);
    {
        ProfScope ps(ctx, "match_init", s);
        hipLaunchKernelGGL(k_match_init, dim3((plan.stride + 255) / 256 > 64 ? 64 : (plan.stride + 255) / 256, plan.M),
                           dim3(256), 0, s, ws, d_counts, d_pairlist, plan.stride, plan.max_n, status);
    }
    // pgx_gate_match: the init kernel touches this context's workspace only, so it runs ahead of the gate (beside whatever the
    // other context still has on the chip) and the first distance round starts the moment the gate opens
    if (gate) (void)hipStreamWaitEvent(s, gate, 0);
    for (int r = 0; r < plan.rounds_mfma; r++) {
        {
            if (plan.words == 8) {   // 256-bit descriptors: the matrix pipe
                ProfScope ps(ctx, "ham_argmin", s, true);
                pgx_launch_ham_mfma(s, ws, d_desc, d_pairlist, plan, ps.a, ps.b, status);
            } else {                 // any other length: xor + popcount
                ProfScope ps(ctx, "ham_argmin", s);
                launch_rounds_valu<0>(s, ws, d_desc, d_pairlist, plan);
            }
        }
        {
            ProfScope ps(ctx, "match_select", s);
            hipLaunchKernelGGL(k_match_select, dim3(plan.M), dim3(SEL_NT), 0, s, ws, plan.stride,
                               reinterpret_cast<unsigned long long *>(status + 4) + (r < PGX_MAX_WIDE_ROUNDS ? r : PGX_MAX_WIDE_ROUNDS - 1), plan.skip_below);
        }
    }
}

void pgx_launch_match_rows(pgx_ctx *ctx, hipStream_t s, const uint32_t *d_desc, const int32_t *d_pairlist,
                           const MatchPlan &plan, void *wsv, int *status)
{
    if (plan.M <= 0 || plan.words != 8) return;
    uint32_t *ws = reinterpret_cast<uint32_t *>(wsv);
    ProfScope ps(ctx, "tail_rows", s);
    hipLaunchKernelGGL(k_tail_rows_fp4, dim3(PGX_TAIL_MAX / TM_ROWS, plan.M), dim3(256), 0, s, ws, d_desc, d_pairlist, plan.stride);
}

void pgx_launch_match_finish(pgx_ctx *ctx, hipStream_t s, const uint32_t *d_desc, const int32_t *d_pairlist,
                             const MatchPlan &plan, void *wsv, pgx_pair *d_out, int *status)
{
    if (plan.M <= 0) return;
    uint32_t *ws = reinterpret_cast<uint32_t *>(wsv);
    ProfScope ps(ctx, "match_finish", s);
    if (plan.words == 8) {
        const size_t n2p = pow2_ge((size_t)(plan.max_n > 1 ? plan.max_n : 1));
        const size_t key_cap = n2p <= (size_t)4 * PGX_TAIL_MAX ? n2p : 0; // sort keys over the finish's own LDS state (32 KiB), else in the workspace
        uint32_t *order = plan.M <= ORDER_MAX ? ws + (size_t)plan.M * pair_ws_words(plan.stride) : nullptr;
        if (order) hipLaunchKernelGGL(k_match_order, dim3(1), dim3(1024), 0, s, ws, plan.stride, plan.M, order);
        hipLaunchKernelGGL(k_match_gs, dim3(plan.M), dim3(plan.M >= GS_SMALL_FROM ? GS_NT : GS_NT_MAX), 0, s, ws, d_desc, d_pairlist, plan.stride, d_out,
                           (uint32_t)key_cap, status, order);
    } else launch_finish<0>(s, ws, d_desc, d_pairlist, plan, d_out, status);
}
