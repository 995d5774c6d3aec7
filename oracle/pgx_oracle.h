/*
 * pgx_oracle.h -- CPU ORACLE (test infrastructure, NOT the product).
 *
 * A literal, single-threaded C restatement of the reference's hot path
 * (Takatsuka-Mark/Photogrammetry, dotnet_src/ImageProcessing), written from the
 * C# sources with every quirk kept.  It exists so that tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() can check the HIP path.
 * Nothing under photogrammetry_amd/ may call, link or import it.
 *
 * PARITY PINNING
 *   pinned   : IsPotentialKeypoint / GetIntensityValueIfKeypoint against the three
 *              xUnit known answers (ImageProcessing.Tests/KeypointDetectionTests.cs:10-50)
 *              and Matrix indexer/transposition semantics (LinearAlgebra.Tests/MatrixTests.cs:41-73).
 *   pinned by a C# OUTPUT (locations): Detect (a4) on 15pt_star.png -- the blue mask of the reference's
 *              data/feature_detection_test/output/dotnet_keypoints_backup.bmp (written by the older flow of
 *              Photogrammetry/Program.cs:116-149, T = 0.2) equals, pixel for pixel, the union of the 10 x 10 squares
 *              (ResultBuilders.cs:41-54) at this oracle's 126 raw hits; 106 hits own a mask pixel no other hit covers
 *              and one single non-hit position could be added unnoticed.  The eliminator's (a7) survivors at
 *              r = (int)(451 * 0.015) sit one per connected component of that mask (30 of 30): a consistency check,
 *              the image shows every raw hit, not the survivors.  tests/test_oracle.py, tests/golden/make_golden.py.
 *   unpinned : BRIEF, NMS order, matching, dewarp -- the reference holds no test, golden file or
 *              numeric output for them and its C#/.NET 8 toolchain is absent here, so for those
 *              stages this oracle is "parity unpinned": it is cross-checked only against an
 *              independently written numpy/Python twin (oracle/oracle_np.py) and hand-derived cases.
 *   third-party, absent: MathNet.Numerics 5.0.0 Cubic.RealRoots (DeWarp.cs:76) is restated from
 *              its published algorithm in orc_build_distortion_matrix -- parity unpinned.
 *
 * Image layout everywhere: row-major [H][W]; pixel (x, y) = column x, row y, which is
 * the reference's Matrix<T>[x, y] (Matrix.cs:44-76).
 */
#ifndef PGX_ORACLE_H
#define PGX_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_OK            0
#define ORC_E_DIM        -1   /* ArgumentException            DeWarp.cs:22-23            */
#define ORC_E_OOB        -2   /* IndexOutOfRangeException     Matrix.cs:63-66,204-209    */
#define ORC_E_EMPTY      -3   /* ArgumentOutOfRangeException  KeypointMatching.cs:61     */
#define ORC_E_CAPACITY   -4
#define ORC_E_BADARG     -5   /* ArgumentException            DeWarp.cs:46-48            */

typedef struct { int32_t x, y, fast_score; float value; } orc_keypoint;
typedef struct { int32_t k1, k2, dist; } orc_pair;

/* DeWarp.ApplyDistortionMat<Rgba64>  (DeWarp.cs:19-37).  map = [H][W][2] (U,V). */
int orc_apply_distortion(const uint16_t *rgba, int W, int H,
                         const int32_t *map_uv, int mapW, int mapH, uint16_t *out);

/* Grayscale.FromRgba64 through Matrix.Convert  (Grayscale.cs:19-23, Converters.cs:15-22). */
void orc_gray(const uint16_t *rgba, int W, int H, float *out);

/* KeypointDetection.InThreshold (KeypointDetection.cs:135-138). */
int orc_in_threshold(float intensity, float test, float T);
/* KeypointDetection.IsPotentialKeypoint (:116-133): 1/0, or ORC_E_OOB. */
int orc_is_potential_keypoint(const float *img, int W, int H, float intensity, int x, int y, float T);
/* KeypointDetection.GetIntensityValueIfKeypoint (:65-114): score 12..16, 0 for null, or ORC_E_OOB. */
int orc_intensity_if_keypoint(const float *img, int W, int H, int x, int y, float T);
/* KeypointDetection.Detect (:42-63) without the BRIEF ctor work: raster-order list.
 * Returns the count (may exceed cap; only the first cap are written). */
int orc_detect(const float *img, int W, int H, float T, orc_keypoint *out, int cap);

/* Keypoint.GetBriefDescriptor (Keypoint.cs:29-57).  pairs = [P][4] (x1,y1,x2,y2).
 * desc = ceil(P/32) little-endian words of the BigInteger: pair p lands on bit P-1-p. */
void orc_brief(const float *img, int W, int H, int x, int y,
               const int32_t *pairs, int P, uint32_t *desc);

/* RedundantKeypointEliminator.EliminateRedundantKeypoints (:16-35).
 * order_out[k] = index (into kps) of the k-th accepted keypoint.  Returns the accepted count. */
int orc_nms(const orc_keypoint *kps, int n, int radius, int32_t *order_out);

/* KeypointMatching.MatchKeypoints (:14-69), literal Theta(N^3) loop.  words = ceil(P/32).
 * out has n1 entries.  ORC_E_EMPTY when n2 == 0 < n1. */
int orc_match(const uint32_t *desc1, int n1, const uint32_t *desc2, int n2, int words, orc_pair *out);
/* Same result by the sorted-scan formulation (sort all edges by (dist,k1,k2), scan);
 * used to cross-check orc_match and to make expectations at sizes the literal loop cannot. */
int orc_match_sorted(const uint32_t *desc1, int n1, const uint32_t *desc2, int n2, int words, orc_pair *out);
/* KeypointMatching.CountOnes (:71-82) on a multi-word value. */
int orc_count_ones(const uint32_t *a, const uint32_t *b, int words);

/* Utils.NextGaussianPair (Utils.cs:14-38) on a SEEDED stream (the reference's is unseeded,
 * SURVEY D6): splitmix64 -> double in [0,1).  out = [P][4]. */
void orc_gaussian_pairs(uint64_t seed, int sigma, int P, int32_t *out);

/* DeWarp.GetDistortionMatrix (DeWarp.cs:39-107); k has ncoef entries (must be 5). out = [H][W][2]. */
int orc_build_distortion_matrix(int W, int H, const double *k, int ncoef, int32_t *out);

#ifdef __cplusplus
}
#endif
#endif
