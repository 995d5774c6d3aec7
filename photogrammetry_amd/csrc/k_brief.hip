// k_brief.hip -- Keypoint.GetBriefDescriptor for gfx950: one wavefront per keypoint.
//
// Reference: ImageProcessing.Abstractions/Keypoint.cs:29-57.  For test pair p (table order):
// descriptor <<= 1; if either test point is outside [0,W)x[0,H) the bit stays 0; else bit =
// (K[c1] < K[c2]).  So pair p lands on BigInteger bit P-1-p.
//
// Lane l of chunk c evaluates pair 64c+l; a 64-bit ballot, bit-reversed, is the descriptor's
// bits [P-64(c+1), P-64c).  Gathers hit L2 (a 1080p grey image is 8.3 MB); per survivor
// 2*P*4 B of gathers and P/8 B written -- negligible next to the detect stream.
// The reference computes BRIEF for every raw hit and NMS then discards most of them; this
// path computes it for the survivors only, which gives the same descriptors.
#include "pgx_internal.h"

namespace {

constexpr int MAX_WORDS = 128; // P <= 4096

// P == 256 fast path: all four pair loads, then all eight gathers, are issued before any is used
// (the generic loop below waits for each 64-pair chunk in turn: 4 dependent L2 round trips).
__device__ __forceinline__ void brief_256(const float *__restrict__ g, int W, int H, int x, int y,
                                          const int4 *__restrict__ pairs, uint32_t *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    int4 pr[4];
#pragma unroll
    for (int c = 0; c < 4; c++) pr[c] = pairs[c * 64 + lane];
    float v1[4], v2[4];
    bool ok[4];
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const int x1 = x + pr[c].x, y1 = y + pr[c].y, x2 = x + pr[c].z, y2 = y + pr[c].w;
        ok[c] = x1 >= 0 && x1 < W && y1 >= 0 && y1 < H && x2 >= 0 && x2 < W && y2 >= 0 && y2 < H; // Keypoint.cs:39-45
        v1[c] = ok[c] ? g[(size_t)y1 * W + x1] : 0.f;
        v2[c] = ok[c] ? g[(size_t)y2 * W + x2] : 0.f;
    }
    unsigned long long rev[4];
#pragma unroll
    for (int c = 0; c < 4; c++) rev[c] = __brevll(__ballot(ok[c] && v1[c] < v2[c])); // :50; bits [192-64c, 256-64c)
    if (lane < 8) {
        const int q = lane >> 1; // 64-bit piece q of the descriptor comes from chunk 3 - q (selects, not a runtime index)
        const unsigned long long r = q == 0 ? rev[3] : (q == 1 ? rev[2] : (q == 2 ? rev[1] : rev[0]));
        out[lane] = (lane & 1) ? (uint32_t)(r >> 32) : (uint32_t)r;
    }
}

__device__ __forceinline__ void brief_one(const float *__restrict__ g, int W, int H, int x, int y,
                                          const int4 *__restrict__ pairs, int P, int words, uint32_t *wbuf /*LDS*/,
                                          uint32_t *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    for (int w = lane; w < words + 2; w += 64) wbuf[w] = 0;
    __builtin_amdgcn_wave_barrier();
    const int nchunk = (P + 63) / 64;
    for (int c = 0; c < nchunk; c++) {
        const int p = c * 64 + lane;
        bool bit = false;
        if (p < P) {
            const int4 pr = pairs[p];
            const int x1 = x + pr.x, y1 = y + pr.y;
            if (x1 >= 0 && x1 < W && y1 >= 0 && y1 < H) {           // Keypoint.cs:39-40
                const int x2 = x + pr.z, y2 = y + pr.w;
                if (x2 >= 0 && x2 < W && y2 >= 0 && y2 < H)          // :44-45
                    bit = g[(size_t)y1 * W + x1] < g[(size_t)y2 * W + x2]; // :50
            }
        }
        const unsigned long long rev = __brevll(__ballot(bit));
        if (lane == 0) {
            const int off = P - 64 * (c + 1); // bit position of rev's bit 0 (may be negative on the last chunk)
            unsigned long long v = rev;
            int o = off;
            if (o < 0) { v >>= -o; o = 0; }
            const int wi = o >> 5, sh = o & 31;
            wbuf[wi] |= (uint32_t)(v << sh);
            wbuf[wi + 1] |= (uint32_t)(sh ? (v >> (32 - sh)) : (v >> 32));
            if (sh) wbuf[wi + 2] |= (uint32_t)(v >> (64 - sh));
        }
    }
    __builtin_amdgcn_wave_barrier();
    for (int w = lane; w < words; w += 64) out[w] = wbuf[w];
}

// fused path: survivors come as indices (order) into the frame's raw list
__global__ __launch_bounds__(256) void k_brief_kept(const float *__restrict__ gray, int W, int H,
                                                    const uint32_t *__restrict__ raw_xy,
                                                    const int32_t *__restrict__ raw_score, int raw_cap,
                                                    const uint32_t *__restrict__ order,
                                                    const int32_t *__restrict__ n_kept, int kp_cap,
                                                    const int4 *__restrict__ pairs, int P, int words,
                                                    pgx_keypoint *__restrict__ kp_out, uint32_t *__restrict__ desc_out,
                                                    int32_t *__restrict__ counts_out, int nframes, int out_stride)
{
    __shared__ uint32_t wbuf[4][MAX_WORDS + 2];
    // XCD-aware block -> (frame, keypoint block) map: consecutive workgroup ids are dealt round-robin to
    // the 8 XCDs, so id % 8 picks the frame inside a group of 8 frames: all gathers of one frame then go
    // through ONE XCD's L2 instead of eight (speed only; any mapping is correct).
    const int nblk = (kp_cap + 3) / 4;
    const int F8 = (nframes / 8) * 8;
    int f, kb;
    {
        const int Lid = blockIdx.x;
        if (Lid < nblk * F8) {
            const int j = Lid >> 3;
            f = (j / nblk) * 8 + (Lid & 7);
            kb = j % nblk;
        } else {
            const int r = Lid - nblk * F8;
            f = F8 + r / nblk;
            kb = r % nblk;
        }
    }
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int k = kb * 4 + wv;
    const int nk = n_kept[f];
    if (kb == 0 && threadIdx.x == 0) counts_out[f] = nk;
    if (k >= nk) return; // wave-uniform
    const uint32_t ri = order[(size_t)f * kp_cap + k];
    const uint32_t xy = raw_xy[(size_t)f * raw_cap + ri];
    const int x = (int)(xy & 0xFFFFu), y = (int)(xy >> 16);
    const float *g = gray + (size_t)f * W * H;
    if (lane == 0) {
        pgx_keypoint kp;
        kp.x = x; kp.y = y; kp.fast_score = raw_score[(size_t)f * raw_cap + ri];
        kp.value = g[(size_t)y * W + x]; // Keypoint.cs:26
        kp_out[(size_t)f * out_stride + k] = kp;
    }
    if (P == 256) brief_256(g, W, H, x, y, pairs, desc_out + ((size_t)f * out_stride + k) * 8);
    else brief_one(g, W, H, x, y, pairs, P, words, wbuf[wv], desc_out + ((size_t)f * out_stride + k) * words);
}

__global__ __launch_bounds__(256) void k_brief_list(const float *__restrict__ gray, int W, int H,
                                                    const pgx_keypoint *__restrict__ kps, int n,
                                                    const int4 *__restrict__ pairs, int P, int words,
                                                    uint32_t *__restrict__ desc_out)
{
    __shared__ uint32_t wbuf[4][MAX_WORDS + 2];
    const int wv = threadIdx.x >> 6;
    const int k = blockIdx.x * 4 + wv;
    if (k >= n) return;
    if (P == 256) brief_256(gray, W, H, kps[k].x, kps[k].y, pairs, desc_out + (size_t)k * 8);
    else brief_one(gray, W, H, kps[k].x, kps[k].y, pairs, P, words, wbuf[wv], desc_out + (size_t)k * words);
}

} // namespace

void pgx_launch_brief(hipStream_t s, const float *gray, int F, int W, int H, const uint32_t *raw_xy,
                      const int32_t *raw_score, int raw_cap, const uint32_t *order, const int32_t *n_kept,
                      int kp_cap, const int32_t *pairs, int P, pgx_keypoint *kp_out, uint32_t *desc_out,
                      int32_t *counts_out, int out_stride)
{
    if (F <= 0 || kp_cap <= 0) return;
    const int words = (P + 31) / 32;
    hipLaunchKernelGGL(k_brief_kept, dim3(((kp_cap + 3) / 4) * F), dim3(256), 0, s, gray, W, H, raw_xy, raw_score,
                       raw_cap, order, n_kept, kp_cap, reinterpret_cast<const int4 *>(pairs), P, words, kp_out,
                       desc_out, counts_out, F, out_stride);
}

void pgx_launch_brief_list(hipStream_t s, const float *gray, int W, int H, const pgx_keypoint *kps, int n,
                           const int32_t *pairs, int P, uint32_t *desc_out)
{
    if (n <= 0) return;
    const int words = (P + 31) / 32;
    hipLaunchKernelGGL(k_brief_list, dim3((n + 3) / 4), dim3(256), 0, s, gray, W, H, kps, n,
                       reinterpret_cast<const int4 *>(pairs), P, words, desc_out);
}
