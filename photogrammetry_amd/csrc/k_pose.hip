// k_pose.hip -- CameraPoseEstimation on gfx950, batched over image pairs (BASELINE configs[4]).
//
// Reference: ImageProcessing/CameraPoseEstimation.cs
//   GetFundamentalMatrix        :26-94    RANSAC over `numSamples` random subsets of `numPairsPerSample` keypoint pairs
//   EstimateFundamentalMatrix   :204-250  normalised 8-point: rows [x1x2, x1y2, x1, y1x2, y1y2, y1, x2, y2, 1], last row of
//                                         VT of the SVD, F = T2^T * F0 * T1 (F0 filled COLUMN-major from that row)
//   CalculateTransformationMatrix :252-274 translation by -centroid; the scale is pow(2/msd, 1/2) with INTEGER 1/2 = 0,
//                                         i.e. always 1 (reproduced); T = translation * scaling
//   inlier test                 :67-77    (F * p2) . p1 <= threshold -- signed, and with the pair's points in the order
//                                         (Keypoint2, Keypoint1) (reproduced as written)
//   EstimateCameraPose          :96-202   E = K^T F K with the hard-coded K, SVD, the four (R, t) candidates with the
//                                         determinant sign fix, linear triangulation per pair (4x4 SVD), vote on z >= 0
//
// What cannot be reproduced bit for bit, and is therefore "parity unpinned" (DESIGN.md): the reference draws the
// subsets from an unseeded System.Random and takes its singular vectors from MathNet.Numerics 5.0.0's float SVD
// (sign and ordering conventions of a third-party library that is not in /root/reference).  Here the subsets come from a
// seeded counter-based generator (the seed is an ABI input, like the BRIEF pair table), the singular vectors from a
// float64 Jacobi eigen-solver of A^T A, and a null vector's sign is fixed by making its largest component positive.
// The oracle (oracle/pose_np.py) restates the same steps with numpy's SVD and the same sign rule.
//
// One thread per RANSAC sample (the work of a sample is a 9x9 symmetric eigenproblem plus one pass over the pair's
// match list); a second kernel per image pair picks the first best sample; a third does the pose.
#include "pgx_internal.h"

namespace {

__device__ __forceinline__ uint64_t splitmix64(uint64_t &s)
{
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// cyclic Jacobi on a symmetric N x N matrix (float64); on return A holds the eigenvalues on its diagonal and the
// columns of V the eigenvectors
template <int N>
__device__ void jacobi_eig(double (&A)[N][N], double (&V)[N][N])
{
#pragma unroll
    for (int i = 0; i < N; i++)
#pragma unroll
        for (int j = 0; j < N; j++) V[i][j] = i == j ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 40; sweep++) {
        double off = 0.0, diag = 0.0;
#pragma unroll
        for (int i = 0; i < N; i++) {
            diag += A[i][i] * A[i][i];
#pragma unroll
            for (int j = i + 1; j < N; j++) off += A[i][j] * A[i][j];
        }
        if (off <= 1e-30 * (diag + 1e-300)) break;
        // p, q and k unrolled: every index is a compile-time constant, so A and V live in registers (2 x 81 doubles for
        // N = 9; with run-time indices they sat in scratch and the solver was bound by scratch latency)
#pragma unroll
        for (int p = 0; p < N - 1; p++)
#pragma unroll
            for (int q = p + 1; q < N; q++) {
                const double apq = A[p][q];
                if (fabs(apq) < 1e-300) continue;
                const double theta = (A[q][q] - A[p][p]) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
#pragma unroll
                for (int k = 0; k < N; k++) {
                    const double akp = A[k][p], akq = A[k][q];
                    A[k][p] = c * akp - s * akq;
                    A[k][q] = s * akp + c * akq;
                }
#pragma unroll
                for (int k = 0; k < N; k++) {
                    const double apk = A[p][k], aqk = A[q][k];
                    A[p][k] = c * apk - s * aqk;
                    A[q][k] = s * apk + c * aqk;
                }
#pragma unroll
                for (int k = 0; k < N; k++) {
                    const double vkp = V[k][p], vkq = V[k][q];
                    V[k][p] = c * vkp - s * vkq;
                    V[k][q] = s * vkp + c * vkq;
                }
            }
    }
}

// eigenvector of the smallest eigenvalue of the symmetric matrix A (destroyed), sign: largest component positive
template <int N>
__device__ void smallest_eigvec(double (&A)[N][N], double (&v)[N])
{
    double V[N][N];
    jacobi_eig<N>(A, V);
    // selections instead of run-time indices (keeps A and V in registers): first smallest diagonal entry, its column
    double lmin = A[0][0];
#pragma unroll
    for (int i = 0; i < N; i++) v[i] = V[i][0];
#pragma unroll
    for (int j = 1; j < N; j++) {
        const bool take = A[j][j] < lmin;
        lmin = take ? A[j][j] : lmin;
#pragma unroll
        for (int i = 0; i < N; i++) v[i] = take ? V[i][j] : v[i];
    }
    double vbig = v[0];
#pragma unroll
    for (int i = 1; i < N; i++) vbig = fabs(v[i]) > fabs(vbig) ? v[i] : vbig;
    if (vbig < 0) {
#pragma unroll
        for (int i = 0; i < N; i++) v[i] = -v[i];
    }
}

struct PairView {
    const pgx_keypoint *kpa, *kpb;
    const pgx_pair *ml;
    int n;
};

__device__ __forceinline__ PairView pair_view(const pgx_keypoint *kp, const pgx_pair *matches, const int32_t *counts,
                                              const int32_t *pairlist, int m, int stride)
{
    PairView v;
    const int a = pairlist[2 * m], b = pairlist[2 * m + 1];
    v.kpa = kp + (size_t)a * stride;
    v.kpb = kp + (size_t)b * stride;
    v.ml = matches + (size_t)m * stride;
    int n = counts[a];
    v.n = n < 0 ? 0 : (n > stride ? stride : n);
    return v;
}

// sample record: [0..8] F row-major, [9] inlier count (as float bits of an int)
constexpr int REC = 10;

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_fund_samples(const pgx_keypoint *__restrict__ kp, const pgx_pair *__restrict__ matches,
                                                     const int32_t *__restrict__ counts, const int32_t *__restrict__ pairlist,
                                                     int stride, int n_samples, int P, float threshold, int rank_check,
                                                     uint64_t seed, float *__restrict__ rec)
{
    const int m = blockIdx.x, s = blockIdx.y * 64 + threadIdx.x;   // image pairs in grid.x: any number of them (grid.y holds 65535 at most)
    if (s >= n_samples) return;
    const PairView pv = pair_view(kp, matches, counts, pairlist, m, stride);
    float *out = rec + ((size_t)m * n_samples + s) * REC;
    int *outc = reinterpret_cast<int *>(out + 9);
    if (pv.n < P) { *outc = -1; return; } // CameraPoseEstimation.cs:31-32 (InvalidOperationException)
    // the subset: P distinct positions of the match list (OrderBy(random).Take(P), :42)
    uint64_t st = seed ^ ((uint64_t)(uint32_t)m << 32) ^ (uint64_t)(uint32_t)s * 0xD1B54A32D192ED03ull;
    int idx[64];
    for (int k = 0; k < P; k++) {
        while (true) {
            const int c = (int)(splitmix64(st) % (uint64_t)pv.n);
            bool dup = false;
            for (int j = 0; j < k; j++) dup |= idx[j] == c;
            if (!dup) { idx[k] = c; break; }
        }
    }
    // CalculateCentroid / CalculateTransformationMatrix (:252-288): float64 sums, float32 matrix entries, scale == 1
    double c1x = 0, c1y = 0, c2x = 0, c2y = 0;
    for (int k = 0; k < P; k++) {
        const pgx_pair e = pv.ml[idx[k]];
        c1x += pv.kpa[e.k1].x; c1y += pv.kpa[e.k1].y;
        c2x += pv.kpb[e.k2].x; c2y += pv.kpb[e.k2].y;
    }
    c1x /= P; c1y /= P; c2x /= P; c2y /= P;
    const float t1x = -(float)c1x, t1y = -(float)c1y, t2x = -(float)c2x, t2y = -(float)c2y;
    // A^T A of the P x 9 system (:221-236), rows in float32 like the reference's DenseMatrix
    double G[9][9];
#pragma unroll
    for (int i = 0; i < 9; i++)
#pragma unroll
        for (int j = 0; j < 9; j++) G[i][j] = 0.0;
    for (int k = 0; k < P; k++) {
        const pgx_pair e = pv.ml[idx[k]];
        const float x1 = (float)pv.kpa[e.k1].x + t1x, y1 = (float)pv.kpa[e.k1].y + t1y;
        const float x2 = (float)pv.kpb[e.k2].x + t2x, y2 = (float)pv.kpb[e.k2].y + t2y;
        const float r[9] = {x1 * x2, x1 * y2, x1, y1 * x2, y1 * y2, y1, x2, y2, 1.0f};
#pragma unroll
        for (int i = 0; i < 9; i++)
#pragma unroll
            for (int j = i; j < 9; j++) G[i][j] += (double)r[i] * (double)r[j];
    }
#pragma unroll
    for (int i = 0; i < 9; i++)
#pragma unroll
        for (int j = 0; j < i; j++) G[i][j] = G[j][i];
    double v[9];
    smallest_eigvec<9>(G, v);
    // F0 = DenseOfColumnMajor(3, 3, lastRow): F0[r][c] = v[3c + r]; F = T2^T * F0 * T1 (:238, :249)
    float F0[3][3], F[3][3];
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) F0[r][c] = (float)v[3 * c + r];
    // T = [[1,0,tx],[0,1,ty],[0,0,1]]:  (T2^T F0)[r][c] = F0[r][c] for r < 2, row 2 = t2x F0[0][c] + t2y F0[1][c] + F0[2][c]
    float M[3][3];
    for (int c = 0; c < 3; c++) {
        M[0][c] = F0[0][c];
        M[1][c] = F0[1][c];
        M[2][c] = t2x * F0[0][c] + t2y * F0[1][c] + F0[2][c];
    }
    for (int r = 0; r < 3; r++) {
        F[r][0] = M[r][0];
        F[r][1] = M[r][1];
        F[r][2] = M[r][0] * t1x + M[r][1] * t1y + M[r][2];
    }
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) out[3 * r + c] = F[r][c];
    if (rank_check) { // :46-51: only matrices of numerical rank 2 are scored (MathNet's Svd().Rank: singular values above eps(s_max) * 3)
        double B[3][3], Vd[3][3];
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
            double acc = 0;
            for (int k = 0; k < 3; k++) acc += (double)F[k][i] * (double)F[k][j];
            B[i][j] = acc;
        }
        jacobi_eig<3>(B, Vd);
        double smax = 0;
        for (int i = 0; i < 3; i++) { const double sv = sqrt(fmax(B[i][i], 0.0)); smax = sv > smax ? sv : smax; }
        const double tol = smax * 1.1920929e-7 * 3.0;
        int rank = 0;
        for (int i = 0; i < 3; i++) rank += sqrt(fmax(B[i][i], 0.0)) > tol ? 1 : 0;
        if (rank != 2) { *outc = -2; return; }
    }
    *outc = 0; // fitted: k_fund_score counts the inliers
}

// score of every fitted sample over ALL keypoint pairs of the list (:53-77), float32: one workgroup per (image pair, 256
// samples); the list's coordinates are gathered once per workgroup into LDS (1024 entries at a time) and every thread
// walks them as broadcast reads -- per-thread walks through global memory (three dependent loads per entry) took 174 ms
// for 3960 image pairs x 2000 samples x 4080 entries.
__global__ __launch_bounds__(256) void k_fund_score(const pgx_keypoint *__restrict__ kp, const pgx_pair *__restrict__ matches,
                                                    const int32_t *__restrict__ counts, const int32_t *__restrict__ pairlist,
                                                    int stride, int n_samples, float threshold, float *__restrict__ rec)
{
    constexpr int CH = 1024;
    __shared__ float4 s_xy[CH];
    const int m = blockIdx.x, s = blockIdx.y * 256 + threadIdx.x;
    const PairView pv = pair_view(kp, matches, counts, pairlist, m, stride);
    const bool active = s < n_samples;
    float *out = rec + ((size_t)m * n_samples + (active ? s : 0)) * REC;
    int *outc = reinterpret_cast<int *>(out + 9);
    const bool fitted = active && *outc == 0;
    float F[3][3];
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) F[r][c] = fitted ? out[3 * r + c] : 0.f;
    int good = 0;
    for (int e0 = 0; e0 < pv.n; e0 += CH) { // block-uniform
        const int cnt = pv.n - e0 < CH ? pv.n - e0 : CH;
        __syncthreads();
        for (int i = threadIdx.x; i < cnt; i += 256) {
            const pgx_pair pr = pv.ml[e0 + i];
            s_xy[i] = make_float4((float)pv.kpa[pr.k1].x, (float)pv.kpa[pr.k1].y, (float)pv.kpb[pr.k2].x, (float)pv.kpb[pr.k2].y);
        }
        __syncthreads();
        for (int i = 0; i < cnt; i++) {
            const float4 q = s_xy[i];
            const float x1 = q.x, y1 = q.y, x2 = q.z, y2 = q.w;
            // f.Multiply([x2, y2, 1]).DotProduct([x1, y1, 1])
            const float a0 = F[0][0] * x2 + F[0][1] * y2 + F[0][2];
            const float a1 = F[1][0] * x2 + F[1][1] * y2 + F[1][2];
            const float a2 = F[2][0] * x2 + F[2][1] * y2 + F[2][2];
            const float res = a0 * x1 + a1 * y1 + a2;
            good += res <= threshold ? 1 : 0;
        }
    }
    if (fitted) *outc = good;
}

// first sample with the largest count (`workingPairs.Count > bestSample.Count`, :79-84)
__global__ __launch_bounds__(256) void k_fund_pick(const float *__restrict__ rec, int n_samples, float *__restrict__ F_out,
                                                   int32_t *__restrict__ inliers, int32_t *__restrict__ best_sample)
{
    __shared__ unsigned long long best;
    const int m = blockIdx.x;
    if (threadIdx.x == 0) best = 0ull;
    __syncthreads();
    unsigned long long k = 0ull;
    for (int s = threadIdx.x; s < n_samples; s += 256) {
        const int c = *reinterpret_cast<const int *>(rec + ((size_t)m * n_samples + s) * REC + 9);
        if (c > 0) { // bestSample starts empty: a sample needs at least one inlier to replace it
            const unsigned long long key = ((unsigned long long)(uint32_t)c << 32) | (uint32_t)(0x7FFFFFFF - s);
            k = key > k ? key : k;
        }
    }
    atomicMax(&best, k);
    __syncthreads();
    if (threadIdx.x == 0) {
        if (best == 0ull) { // every sample skipped or without inliers: the reference throws (:88-89); the library reports -1
            inliers[m] = -1;
            best_sample[m] = -1;
            for (int i = 0; i < 9; i++) F_out[(size_t)m * 9 + i] = 0.f;
        } else {
            const int s = 0x7FFFFFFF - (int)(uint32_t)(best & 0xFFFFFFFFull);
            inliers[m] = (int32_t)(best >> 32);
            best_sample[m] = s;
            for (int i = 0; i < 9; i++) F_out[(size_t)m * 9 + i] = rec[((size_t)m * n_samples + s) * REC + i];
        }
    }
}

// ---- EstimateCameraPose (:96-202) -------------------------------------------------------------------------------
struct PoseCand { float R[3][3], t[3]; };

__device__ void mat3_mul(const float (&A)[3][3], const float (&B)[3][3], float (&C)[3][3])
{
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) C[i][j] = A[i][0] * B[0][j] + A[i][1] * B[1][j] + A[i][2] * B[2][j];
}

__device__ float det3(const float (&A)[3][3])
{
    return A[0][0] * (A[1][1] * A[2][2] - A[1][2] * A[2][1]) - A[0][1] * (A[1][0] * A[2][2] - A[1][2] * A[2][0]) +
           A[0][2] * (A[1][0] * A[2][1] - A[1][1] * A[2][0]);
}

__global__ __launch_bounds__(256) void k_pose(const pgx_keypoint *__restrict__ kp, const pgx_pair *__restrict__ matches,
                                              const int32_t *__restrict__ counts, const int32_t *__restrict__ pairlist, int stride,
                                              const float *__restrict__ F_in, float *__restrict__ Rt_out /*[M][12]*/,
                                              int32_t *__restrict__ votes /*[M][4]*/, int32_t *__restrict__ best_out,
                                              float *__restrict__ points /*[M][stride][3] or null*/)
{
    __shared__ PoseCand cand[4];
    __shared__ int cnt[4];
    __shared__ int bestc;
    const int m = blockIdx.x, tid = threadIdx.x;
    const PairView pv = pair_view(kp, matches, counts, pairlist, m, stride);
    const float K[3][3] = {{1000.f, 0.f, 1500.f}, {0.f, 1000.f, 2000.f}, {0.f, 0.f, 1.f}};          // :98-99
    const float Ki[3][3] = {{0.001f, 0.f, -1.5f}, {0.f, 0.001f, -2.f}, {0.f, 0.f, 1.f}};            // its inverse
    if (tid == 0) {
        float F[3][3], Kt[3][3], T[3][3], E[3][3];
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { F[i][j] = F_in[(size_t)m * 9 + 3 * i + j]; Kt[i][j] = K[j][i]; }
        mat3_mul(Kt, F, T);
        mat3_mul(T, K, E);                                                                              // :102
        // SVD of E from the eigen-decomposition of E^T E: E = U S V^T, singular values descending
        double B[3][3], V[3][3];
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
            double acc = 0;
            for (int k = 0; k < 3; k++) acc += (double)E[k][i] * (double)E[k][j];
            B[i][j] = acc;
        }
        jacobi_eig<3>(B, V);
        int ord[3] = {0, 1, 2};
        for (int a = 0; a < 2; a++) for (int b = a + 1; b < 3; b++) if (B[ord[b]][ord[b]] > B[ord[a]][ord[a]]) { const int t = ord[a]; ord[a] = ord[b]; ord[b] = t; }
        double Vs[3][3], U[3][3];
        for (int c = 0; c < 3; c++) {
            int big = 0;
            for (int r = 0; r < 3; r++) { Vs[r][c] = V[r][ord[c]]; if (fabs(Vs[r][c]) > fabs(Vs[big][c])) big = r; }
            if (Vs[big][c] < 0) for (int r = 0; r < 3; r++) Vs[r][c] = -Vs[r][c]; // sign rule: largest component positive
        }
        for (int c = 0; c < 2; c++) { // u_c = E v_c / s_c
            double u[3], n = 0;
            for (int r = 0; r < 3; r++) { u[r] = E[r][0] * Vs[0][c] + E[r][1] * Vs[1][c] + E[r][2] * Vs[2][c]; n += u[r] * u[r]; }
            n = sqrt(n);
            for (int r = 0; r < 3; r++) U[r][c] = n > 0 ? u[r] / n : (r == c ? 1.0 : 0.0);
        }
        // third left vector: u0 x u1 (E has rank 2 in exact arithmetic), sign rule as above
        U[0][2] = U[1][0] * U[2][1] - U[2][0] * U[1][1];
        U[1][2] = U[2][0] * U[0][1] - U[0][0] * U[2][1];
        U[2][2] = U[0][0] * U[1][1] - U[1][0] * U[0][1];
        {
            int big = 0;
            for (int r = 1; r < 3; r++) if (fabs(U[r][2]) > fabs(U[big][2])) big = r;
            if (U[big][2] < 0) for (int r = 0; r < 3; r++) U[r][2] = -U[r][2];
        }
        float Uf[3][3], VT[3][3];
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { Uf[i][j] = (float)U[i][j]; VT[i][j] = (float)Vs[j][i]; }
        const float W[3][3] = {{0.f, -1.f, 0.f}, {1.f, 0.f, 0.f}, {0.f, 0.f, 1.f}}, Wt[3][3] = {{0.f, 1.f, 0.f}, {-1.f, 0.f, 0.f}, {0.f, 0.f, 1.f}};
        float T1[3][3], R1[3][3], R2[3][3];
        mat3_mul(Uf, W, T1); mat3_mul(T1, VT, R1);                                                      // :112
        mat3_mul(Uf, Wt, T1); mat3_mul(T1, VT, R2);                                                     // :113
        const float s1 = det3(R1) > 0 ? 1.f : -1.f, s2 = det3(R2) > 0 ? 1.f : -1.f;                    // :115-116
        for (int c = 0; c < 4; c++) {                                                                   // :120-125
            const float sc = c < 2 ? s1 : s2, sg = (c & 1) ? -1.f : 1.f;
            for (int i = 0; i < 3; i++) {
                cand[c].t[i] = Uf[i][2] * sg * sc;
                for (int j = 0; j < 3; j++) cand[c].R[i][j] = (c < 2 ? R1[i][j] : R2[i][j]) * sc;
            }
            cnt[c] = 0;
        }
    }
    __syncthreads();
    // linear triangulation of every keypoint pair under every candidate (:143-180)
    for (int e = tid; e < pv.n; e += 256) {
        const pgx_pair pr = pv.ml[e];
        const float x1 = (float)pv.kpa[pr.k1].x, y1 = (float)pv.kpa[pr.k1].y;
        const float x2 = (float)pv.kpb[pr.k2].x, y2 = (float)pv.kpb[pr.k2].y;
        const float n1x = Ki[0][0] * x1 + Ki[0][2], n1y = Ki[1][1] * y1 + Ki[1][2];
        const float n2x = Ki[0][0] * x2 + Ki[0][2], n2y = Ki[1][1] * y2 + Ki[1][2];
        for (int c = 0; c < 4; c++) {
            const PoseCand &pc = cand[c];
            float D[4][4];
            // P1 = [I | 0], P2 = [R | t]
            D[0][0] = 1.f; D[0][1] = 0.f; D[0][2] = -n1x; D[0][3] = 0.f;                                   // P1.Row(0) - P1.Row(2) * n1x
            D[1][0] = 0.f; D[1][1] = -1.f; D[1][2] = n1y; D[1][3] = 0.f;                                   // P1.Row(2) * n1y - P1.Row(1)
            for (int j = 0; j < 3; j++) {
                D[2][j] = pc.R[0][j] - pc.R[2][j] * n2x;
                D[3][j] = pc.R[2][j] * n2y - pc.R[1][j];
            }
            D[2][3] = pc.t[0] - pc.t[2] * n2x;
            D[3][3] = pc.t[2] * n2y - pc.t[1];
            double G[4][4], X[4];
            for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) {
                double acc = 0;
                for (int k = 0; k < 4; k++) acc += (double)D[k][i] * (double)D[k][j];
                G[i][j] = acc;
            }
            smallest_eigvec<4>(G, X);                                                                   // V.Column(3) of the SVD (:170-171)
            const float sx = (float)(X[0] / X[3]), sy = (float)(X[1] / X[3]), sz = (float)(X[2] / X[3]); // :173
            const float pz = pc.R[2][0] * sx + pc.R[2][1] * sy + pc.R[2][2] * sz + pc.t[2];             // :174 (z of R * X + t)
            if (pz >= 0.f) atomicAdd(&cnt[c], 1);                                                        // :181-184
        }
    }
    __syncthreads();
    if (tid == 0) {
        int b = 0;
        for (int c = 1; c < 4; c++) if (cnt[c] > cnt[b]) b = c;                                          // IndexOf(Max) = first maximum (:197)
        bestc = b;
        best_out[m] = b;
        for (int c = 0; c < 4; c++) votes[(size_t)m * 4 + c] = cnt[c];
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Rt_out[(size_t)m * 12 + 3 * i + j] = cand[b].R[i][j];
        for (int i = 0; i < 3; i++) Rt_out[(size_t)m * 12 + 9 + i] = cand[b].t[i];
    }
    if (!points) return;
    __syncthreads();
    const PoseCand &pc = cand[bestc];
    for (int e = tid; e < pv.n; e += 256) { // the winning candidate's point cloud (Utils.CreatePointCloud input, :199)
        const pgx_pair pr = pv.ml[e];
        const float x1 = (float)pv.kpa[pr.k1].x, y1 = (float)pv.kpa[pr.k1].y;
        const float x2 = (float)pv.kpb[pr.k2].x, y2 = (float)pv.kpb[pr.k2].y;
        const float n1x = Ki[0][0] * x1 + Ki[0][2], n1y = Ki[1][1] * y1 + Ki[1][2];
        const float n2x = Ki[0][0] * x2 + Ki[0][2], n2y = Ki[1][1] * y2 + Ki[1][2];
        float D[4][4];
        D[0][0] = 1.f; D[0][1] = 0.f; D[0][2] = -n1x; D[0][3] = 0.f;
        D[1][0] = 0.f; D[1][1] = -1.f; D[1][2] = n1y; D[1][3] = 0.f;
        for (int j = 0; j < 3; j++) {
            D[2][j] = pc.R[0][j] - pc.R[2][j] * n2x;
            D[3][j] = pc.R[2][j] * n2y - pc.R[1][j];
        }
        D[2][3] = pc.t[0] - pc.t[2] * n2x;
        D[3][3] = pc.t[2] * n2y - pc.t[1];
        double G[4][4], X[4];
        for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) {
            double acc = 0;
            for (int k = 0; k < 4; k++) acc += (double)D[k][i] * (double)D[k][j];
            G[i][j] = acc;
        }
        smallest_eigvec<4>(G, X);
        const float sx = (float)(X[0] / X[3]), sy = (float)(X[1] / X[3]), sz = (float)(X[2] / X[3]);
        float *o = points + ((size_t)m * stride + e) * 3;
        o[0] = pc.R[0][0] * sx + pc.R[0][1] * sy + pc.R[0][2] * sz + pc.t[0];
        o[1] = pc.R[1][0] * sx + pc.R[1][1] * sy + pc.R[1][2] * sz + pc.t[1];
        o[2] = pc.R[2][0] * sx + pc.R[2][1] * sy + pc.R[2][2] * sz + pc.t[2];
    }
}

} // namespace

size_t pgx_pose_ws_bytes(int M, int n_samples) { return (size_t)M * n_samples * REC * 4; }

void pgx_launch_fundamental(hipStream_t s, const pgx_keypoint *kp, const pgx_pair *matches, const int32_t *counts,
                            const int32_t *pairlist, int M, int stride, int n_samples, int P, float threshold, int rank_check,
                            uint64_t seed, void *ws, float *F_out, int32_t *inliers, int32_t *best_sample)
{
    if (M <= 0 || n_samples <= 0) return;
    float *rec = reinterpret_cast<float *>(ws);
    hipLaunchKernelGGL(k_fund_samples, dim3(M, (n_samples + 63) / 64), dim3(64), 0, s, kp, matches, counts, pairlist, stride,
                       n_samples, P, threshold, rank_check, seed, rec);
    hipLaunchKernelGGL(k_fund_score, dim3(M, (n_samples + 255) / 256), dim3(256), 0, s, kp, matches, counts, pairlist, stride, n_samples,
                       threshold, rec);
    hipLaunchKernelGGL(k_fund_pick, dim3(M), dim3(256), 0, s, rec, n_samples, F_out, inliers, best_sample);
}

void pgx_launch_pose(hipStream_t s, const pgx_keypoint *kp, const pgx_pair *matches, const int32_t *counts,
                     const int32_t *pairlist, int M, int stride, const float *F_in, float *Rt_out, int32_t *votes,
                     int32_t *best, float *points)
{
    if (M <= 0) return;
    hipLaunchKernelGGL(k_pose, dim3(M), dim3(256), 0, s, kp, matches, counts, pairlist, stride, F_in, Rt_out, votes, best, points);
}
