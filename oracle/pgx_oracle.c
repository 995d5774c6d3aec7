/*
 * pgx_oracle.c -- CPU ORACLE (test infrastructure, NOT the product).  See pgx_oracle.h
 * for the pinning statement.  Every function cites the reference lines it restates
 * (paths relative to the reference root, dotnet_src/...).
 *
 * Written to be literal, not fast: same loop orders, same comparison operators,
 * same float widths.  Compiled with -ffp-contract=off and without -ffast-math.
 */
#include "pgx_oracle.h"

#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---- Matrix<T> addressing (Math/LinearAlgebra/Matrix.cs:44-76,194-209) --------------- */

/* Matrix.Get(int x, int y): rejects negatives / > ushort.MaxValue, then AssertInBounds. */
static int mat_in_bounds_int(int x, int y, int W, int H)
{
    if (x < 0 || y < 0 || x > 65535 || y > 65535) return 0; /* Matrix.cs:63-66 */
    return (y < H) && (x < W);                                /* Matrix.cs:198-202 */
}

/* ---- DeWarp.ApplyDistortionMat (ImageProcessing/DeWarp.cs:19-37) --------------------- */

int orc_apply_distortion(const uint16_t *rgba, int W, int H,
                         const int32_t *map_uv, int mapW, int mapH, uint16_t *out)
{
    if (W != mapW || H != mapH) return ORC_E_DIM; /* DeWarp.cs:22-23 */
    /* x outer, y inner, as DeWarp.cs:27-34; the first OOB source aborts (exception). */
    for (int x = 0; x < W; x++) {
        for (int y = 0; y < H; y++) {
            const int32_t *uv = map_uv + ((size_t)y * W + x) * 2;
            /* (ushort) casts are unchecked: value mod 65536 (DeWarp.cs:32). */
            unsigned su = (unsigned)(uint16_t)uv[0];
            unsigned sv = (unsigned)(uint16_t)uv[1];
            if (!(sv < (unsigned)H && su < (unsigned)W)) return ORC_E_OOB; /* Matrix.cs:72-76 */
            memcpy(out + ((size_t)y * W + x) * 4, rgba + ((size_t)sv * W + su) * 4, 8);
        }
    }
    return ORC_OK;
}

/* ---- Grayscale.FromRgba64 (Images.Abstractions/Pixels/Grayscale.cs:19-23) ------------ */

void orc_gray(const uint16_t *rgba, int W, int H, float *out)
{
    const float denom = (float)(3 * 65535);
    for (int y = 0; y < H; y++) {          /* Matrix.Convert: y outer, x inner (Matrix.cs:132-138) */
        for (int x = 0; x < W; x++) {
            const uint16_t *p = rgba + ((size_t)y * W + x) * 4;
            volatile float s = (float)p[0]; /* ((float)R + B + G): R, then B, then G */
            s = s + (float)p[2];
            s = s + (float)p[1];
            out[(size_t)y * W + x] = s / denom;
        }
    }
}

/* ---- KeypointDetection (ImageProcessing/KeypointDetection.cs) ------------------------ */

/* Table at :15-19, INCLUDING the last entry {-3, 1} (duplicate of entry 1).
 * FromRowMajorArray + Transpose (Matrix.cs:24-40,143-158) make T[idx,0]=entry[idx][0] (added
 * to x) and T[idx,1]=entry[idx][1] (added to y), see :84. */
static const int k_circle[16][2] = {
    {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}, {0, 3}, {1, 3}, {2, 2}, {3, 1}, {3, 0}, {3, -1},
    {2, -2}, {1, -3}, {0, -3}, {-1, -3}, {-2, -2}, {-3, 1}};
static const int k_mini[4][2] = {{-3, 0}, {0, 3}, {3, 0}, {0, -3}}; /* :21-22 */

int orc_in_threshold(float intensity, float test, float T)
{
    /* :137  testIntensity > intensity - T && testIntensity < intensity + T, all float32 */
    volatile float lo = intensity - T;
    volatile float hi = intensity + T;
    return (test > lo) && (test < hi);
}

int orc_is_potential_keypoint(const float *img, int W, int H, float intensity, int x, int y, float T)
{
    int num_inside = 0; /* :118 */
    for (int idx = 0; idx < 4; idx++) {
        int px = k_mini[idx][0] + x, py = k_mini[idx][1] + y;
        if (!mat_in_bounds_int(px, py, W, H)) return ORC_E_OOB;
        if (!orc_in_threshold(intensity, img[(size_t)py * W + px], T)) continue; /* :122-124 */
        if (num_inside > 0) return 0;                                          /* :126-127 */
        num_inside += 1;
    }
    return 1;
}

int orc_intensity_if_keypoint(const float *img, int W, int H, int x, int y, float T)
{
    if (!mat_in_bounds_int(x, y, W, H)) return ORC_E_OOB;
    float intensity = img[(size_t)y * W + x]; /* :67 */
    int pot = orc_is_potential_keypoint(img, W, H, intensity, x, y, T);
    if (pot < 0) return pot;
    if (!pot) return 0; /* :69-70 */

    int is_beginning = 1, num_beginning = 0, longest = 0, current = 0, num_fail = 0; /* :72-76 */
    for (int idx = 0; idx < 16; idx++) {
        int px = k_circle[idx][0] + x, py = k_circle[idx][1] + y;
        if (!mat_in_bounds_int(px, py, W, H)) return ORC_E_OOB;
        if (orc_in_threshold(intensity, img[(size_t)py * W + px], T)) { /* :83-84 */
            is_beginning = 0;
            if (current > longest) longest = current; /* :88 */
            current = 0;
            if (num_fail >= 4) return 0; /* :91-92 */
            num_fail += 1;
        } else {
            current += 1;
            if (is_beginning) num_beginning += 1; /* :99-102 */
        }
    }
    if (!is_beginning) current += num_beginning; /* :106-110 */
    if (current > longest) longest = current;    /* :112 */
    return longest < 12 ? 0 : longest;           /* :113 */
}

int orc_detect(const float *img, int W, int H, float T, orc_keypoint *out, int cap)
{
    int n = 0;
    /* :45-47  y outer [3, H-3), x inner [3, W-3) */
    for (int y = 3; y < H - 3; y++) {
        for (int x = 3; x < W - 3; x++) {
            int s = orc_intensity_if_keypoint(img, W, H, x, y, T);
            if (s > 0) {
                if (n < cap) {
                    out[n].x = x;
                    out[n].y = y;
                    out[n].fast_score = s;
                    out[n].value = img[(size_t)y * W + x]; /* Keypoint.cs:26 */
                }
                n++;
            }
        }
    }
    return n;
}

/* ---- Keypoint.GetBriefDescriptor (ImageProcessing.Abstractions/Keypoint.cs:29-57) ---- */

static void big_shl1(uint32_t *w, int words)
{
    uint32_t carry = 0;
    for (int i = 0; i < words; i++) {
        uint32_t nc = w[i] >> 31;
        w[i] = (w[i] << 1) | carry;
        carry = nc;
    }
}

void orc_brief(const float *img, int W, int H, int x, int y,
               const int32_t *pairs, int P, uint32_t *desc)
{
    int words = (P + 31) / 32;
    /* one spare word so the unbounded BigInteger never loses a bit while shifting */
    uint32_t *big = (uint32_t *)calloc((size_t)words + 1, sizeof(uint32_t));
    for (int p = 0; p < P; p++) {
        big_shl1(big, words + 1);                           /* :36  descriptor <<= 1 */
        int x1 = x + pairs[4 * p + 0], y1 = y + pairs[4 * p + 1]; /* :37 Coordinate.Add */
        if (!(x1 >= 0 && x1 < W && y1 >= 0 && y1 < H)) continue;  /* :39-40, Coordinate.cs:13-17 */
        int x2 = x + pairs[4 * p + 2], y2 = y + pairs[4 * p + 3];
        if (!(x2 >= 0 && x2 < W && y2 >= 0 && y2 < H)) continue;  /* :44-45 */
        float v1 = img[(size_t)y1 * W + x1];
        float v2 = img[(size_t)y2 * W + x2];
        if (v1 < v2) big[0] += 1; /* :50-53; the low bit is 0 after the shift, so += 1 sets it */
    }
    memcpy(desc, big, (size_t)words * sizeof(uint32_t));
    free(big);
}

/* ---- RedundantKeypointEliminator (ImageProcessing/RedundantKeypointEliminator.cs) ---- */

static double coordinate_distance(const orc_keypoint *a, const orc_keypoint *b)
{
    /* Utils.cs:49-51  Math.Sqrt(Math.Pow(dx,2) + Math.Pow(dy,2)) */
    return sqrt(pow((double)(b->x - a->x), 2.0) + pow((double)(b->y - a->y), 2.0));
}

int orc_nms(const orc_keypoint *kps, int n, int radius, int32_t *order_out)
{
    if (n <= 0) return 0;
    /* :21 OrderByDescending(FastScore) -- LINQ's sort is stable: insertion by score buckets. */
    int32_t *list = (int32_t *)malloc((size_t)n * sizeof(int32_t));
    int32_t *next = (int32_t *)malloc((size_t)n * sizeof(int32_t));
    int m = n;
    for (int i = 0; i < n; i++) list[i] = i;
    /* bottom-up merge sort, descending by score, ties keep input order (stable) */
    for (int width = 1; width < n; width *= 2) {
        for (int lo = 0; lo < n; lo += 2 * width) {
            int mid = lo + width < n ? lo + width : n;
            int hi = lo + 2 * width < n ? lo + 2 * width : n;
            int a = lo, b = mid, o = lo;
            while (a < mid && b < hi)
                next[o++] = (kps[list[b]].fast_score > kps[list[a]].fast_score) ? list[b++] : list[a++];
            while (a < mid) next[o++] = list[a++];
            while (b < hi) next[o++] = list[b++];
        }
        int32_t *t = list; list = next; next = t;
    }
    int accepted = 0;
    while (m > 0) { /* :24 */
        int head = list[0]; /* :26-28 */
        order_out[accepted++] = head;
        int m2 = 0;
        for (int k = 1; k < m; k++) { /* :31 Where(IsAcceptableDistance) keeps order */
            if (coordinate_distance(&kps[list[k]], &kps[head]) > (double)radius) /* :37-39 */
                next[m2++] = list[k];
        }
        int32_t *t = list; list = next; next = t;
        m = m2;
    }
    free(list);
    free(next);
    return accepted;
}

/* ---- KeypointMatching (ImageProcessing/KeypointMatching.cs) -------------------------- */

int orc_count_ones(const uint32_t *a, const uint32_t *b, int words)
{
    /* :71-82 Kernighan loop on the BigInteger a ^ b, one word at a time */
    int ones = 0;
    for (int w = 0; w < words; w++) {
        uint32_t v = a[w] ^ b[w];
        while (v != 0) { v &= (v - 1); ones += 1; }
    }
    return ones;
}

int orc_match(const uint32_t *desc1, int n1, const uint32_t *desc2, int n2, int words, orc_pair *out)
{
    if (n1 <= 0) return ORC_OK;          /* :38 loop never runs; empty list */
    if (n2 <= 0) return ORC_E_EMPTY;     /* :61 keypoints2[0] throws */
    /* :20-31 full table */
    int32_t *table = (int32_t *)malloc((size_t)n1 * n2 * sizeof(int32_t));
    for (int k1 = 0; k1 < n1; k1++)
        for (int k2 = 0; k2 < n2; k2++)
            table[(size_t)k1 * n2 + k2] =
                orc_count_ones(desc1 + (size_t)k1 * words, desc2 + (size_t)k2 * words, words);
    /* :34-35 HashSets enumerate ascending (inserted ascending, only removals follow). */
    int32_t *avail1 = (int32_t *)malloc((size_t)n1 * sizeof(int32_t));
    int32_t *avail2 = (int32_t *)malloc((size_t)n2 * sizeof(int32_t));
    int a1 = n1, a2 = n2;
    for (int i = 0; i < n1; i++) avail1[i] = i;
    for (int i = 0; i < n2; i++) avail2[i] = i;

    for (int emitted = 0; emitted < n1; emitted++) { /* :38 */
        int smallest = INT_MAX, sk1 = 0, sk2 = 0;    /* :40-42 */
        for (int i = 0; i < a1; i++) {
            const int32_t *row = table + (size_t)avail1[i] * n2;
            for (int j = 0; j < a2; j++) {
                int d = row[avail2[j]];
                if (smallest <= d) continue; /* :49-50 */
                smallest = d; sk1 = avail1[i]; sk2 = avail2[j];
            }
        }
        out[emitted].k1 = sk1; out[emitted].k2 = sk2; out[emitted].dist = smallest; /* :57-62 */
        /* :64-65 Remove (no-op when the element is absent) */
        for (int i = 0; i < a1; i++)
            if (avail1[i] == sk1) { memmove(avail1 + i, avail1 + i + 1, (size_t)(a1 - i - 1) * 4); a1--; break; }
        for (int j = 0; j < a2; j++)
            if (avail2[j] == sk2) { memmove(avail2 + j, avail2 + j + 1, (size_t)(a2 - j - 1) * 4); a2--; break; }
    }
    free(table); free(avail1); free(avail2);
    return ORC_OK;
}

int orc_match_sorted(const uint32_t *desc1, int n1, const uint32_t *desc2, int n2, int words, orc_pair *out)
{
    if (n1 <= 0) return ORC_OK;
    if (n2 <= 0) return ORC_E_EMPTY;
    const int maxd = words * 32;
    size_t total = (size_t)n1 * n2;
    uint16_t *dist = (uint16_t *)malloc(total * sizeof(uint16_t));
    size_t *count = (size_t *)calloc((size_t)maxd + 2, sizeof(size_t));
    for (int k1 = 0; k1 < n1; k1++)
        for (int k2 = 0; k2 < n2; k2++) {
            int d = 0;
            for (int w = 0; w < words; w++)
                d += __builtin_popcount(desc1[(size_t)k1 * words + w] ^ desc2[(size_t)k2 * words + w]);
            dist[(size_t)k1 * n2 + k2] = (uint16_t)d;
            count[d + 1]++;
        }
    for (int d = 0; d <= maxd; d++) count[d + 1] += count[d];
    /* stable counting sort of edge ids (already (k1,k2)-ascending) by distance */
    uint32_t *edges = (uint32_t *)malloc(total * sizeof(uint32_t));
    for (size_t e = 0; e < total; e++) edges[count[dist[e]]++] = (uint32_t)e;
    uint8_t *used1 = (uint8_t *)calloc((size_t)n1, 1), *used2 = (uint8_t *)calloc((size_t)n2, 1);
    int emitted = 0;
    for (size_t i = 0; i < total && emitted < n1 && emitted < n2; i++) {
        uint32_t e = edges[i];
        int k1 = (int)(e / (uint32_t)n2), k2 = (int)(e % (uint32_t)n2);
        if (used1[k1] || used2[k2]) continue;
        used1[k1] = used2[k2] = 1;
        out[emitted].k1 = k1; out[emitted].k2 = k2; out[emitted].dist = dist[e];
        emitted++;
    }
    for (; emitted < n1; emitted++) { /* columns exhausted: the C# loop emits (0,0,int.MaxValue) */
        out[emitted].k1 = 0; out[emitted].k2 = 0; out[emitted].dist = INT_MAX;
    }
    free(dist); free(count); free(edges); free(used1); free(used2);
    return ORC_OK;
}

/* ---- Utils.NextGaussianPair (ImageProcessing/Utils.cs:14-38), seeded ----------------- */

static uint64_t splitmix64(uint64_t *s)
{
    uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static double next_double(uint64_t *s) { return (double)(splitmix64(s) >> 11) * (1.0 / 9007199254740992.0); }

static void gaussian_coordinate(uint64_t *s, int sigma, int32_t *xy)
{
    double y1, y2, r2;
    do { /* :28-34; y1,y2 in [0,1) -- not [-1,1) -- so offsets are never negative */
        y1 = next_double(s);
        y2 = next_double(s);
        r2 = y1 * y1 + y2 * y2;
    } while (r2 >= 1);
    double sc = sqrt(-2 * log(r2) / r2);     /* :36 */
    xy[0] = (int32_t)(sc * y1 * sigma);      /* :37 (int) truncates toward zero */
    xy[1] = (int32_t)(sc * y2 * sigma);
}

void orc_gaussian_pairs(uint64_t seed, int sigma, int P, int32_t *out)
{
    uint64_t s = seed;
    for (int p = 0; p < P; p++) { /* :16 (NextGaussianCoordinate, NextGaussianCoordinate) */
        gaussian_coordinate(&s, sigma, out + 4 * p);
        gaussian_coordinate(&s, sigma, out + 4 * p + 2);
    }
}

/* ---- DeWarp.GetDistortionMatrix (ImageProcessing/DeWarp.cs:39-107) ------------------- */

static int32_t to_int_trunc(double v)
{
    /* C# (int)double: truncation; NaN / out of range yield 0x80000000 on x64 (cvttsd2si). */
    if (!(v > -2147483649.0 && v < 2147483648.0)) return INT32_MIN;
    return (int32_t)v;
}

static double pow_third(double n)
{
    double sgn = (n > 0) - (n < 0);
    return pow(fabs(n), 1.0 / 3.0) * sgn;
}

/* MathNet.Numerics 5.0.0 RootFinding.Cubic.RealRoots(a0, a1, a2) for x^3 + a2 x^2 + a1 x + a0,
 * restated from the published algorithm (the package is absent here: parity unpinned). */
static void cubic_real_roots(double a0, double a1, double a2, double r[3])
{
    double Q = (3 * a1 - a2 * a2) / 9.0;
    double R = (9.0 * a2 * a1 - 27 * a0 - 2 * a2 * a2 * a2) / 54.0;
    double Q3 = Q * Q * Q;
    double D = Q3 + R * R;
    double shift = -a2 / 3.0;
    r[0] = r[1] = r[2] = NAN;
    if (D >= 0) {
        double sqrtD = pow(D, 0.5);
        double S = pow_third(R + sqrtD);
        double T = pow_third(R - sqrtD);
        r[0] = shift + (S + T);
        if (D == 0) r[1] = shift - S;
    } else {
        double theta = acos(R / sqrt(-Q3));
        const double pi = 3.1415926535897932384626433832795;
        r[0] = 2.0 * sqrt(-Q) * cos(theta / 3.0) + shift;
        r[1] = 2.0 * sqrt(-Q) * cos((theta + 2.0 * pi) / 3.0) + shift;
        r[2] = 2.0 * sqrt(-Q) * cos((theta - 2.0 * pi) / 3.0) + shift;
    }
}

int orc_build_distortion_matrix(int W, int H, const double *k, int ncoef, int32_t *out)
{
    if (ncoef != 5) return ORC_E_BADARG; /* :46-48 */
    double x0 = W / 2.0, y0 = H / 2.0;   /* :53-54 */
    for (int u = 0; u < W; u++) {
        for (int v = 0; v < H; v++) {
            int x = to_int_trunc(u - x0); /* :60-61 */
            int y = to_int_trunc(v - y0);
            int rd2 = x * x + y * y;
            /* :65-86; the per-rd2 cache only memoises a pure function of rd2 */
            double rd = sqrt((double)rd2);
            double den = rd * k[4] - k[1];
            double b = (rd * k[3] - k[0]) / den;
            double c = (rd * k[2] - 1) / den;
            double d = rd / den;
            double roots[3], sorted[3];
            cubic_real_roots(d, c, b, roots); /* :76 */
            int n = 0;
            for (int i = 0; i < 3; i++) if (!isnan(roots[i])) sorted[n++] = roots[i]; /* :78 */
            if (n == 0) return ORC_E_BADARG; /* sortedRoots[0] on an empty list throws */
            for (int i = 1; i < n; i++)      /* :79 Sort */
                for (int j = i; j > 0 && sorted[j - 1] > sorted[j]; j--) {
                    double t = sorted[j]; sorted[j] = sorted[j - 1]; sorted[j - 1] = t;
                }
            double root = (n == 3) ? sorted[1] : sorted[0]; /* :82 */
            double theta = atan2((double)y, (double)x);     /* :93 */
            double xd = root * cos(theta), yd = root * sin(theta);
            out[((size_t)v * W + u) * 2 + 0] = to_int_trunc(xd + x0); /* :98-102 */
            out[((size_t)v * W + u) * 2 + 1] = to_int_trunc(yd + y0);
        }
    }
    return ORC_OK;
}
