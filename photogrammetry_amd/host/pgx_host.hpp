// pgx_host.hpp -- compiled-host mirror of the reference's ImageProcessing classes over the C ABI
// (include/pgx.h).  The reference's host is C#/.NET 8, whose toolchain is absent here, so the host
// side above the ABI is written in C++ with the SAME class names, method names, argument meaning and
// error behaviour as the C# (exceptions below stand for the .NET types); the literal C# binding is
// integration/csharp/*.cs.  Header-only; link with libpgx.so.
//
//   DeWarp.GetDistortionMatrix / ApplyDistortionMat           ImageProcessing/DeWarp.cs:19-107
//   Grayscale.FromRgba64 (through Matrix.Convert)             Images.Abstractions/Pixels/Grayscale.cs:19-23
//   KeypointDetection.Detect                                  ImageProcessing/KeypointDetection.cs:42-63
//   RedundantKeypointEliminator.EliminateRedundantKeypoints   ImageProcessing/RedundantKeypointEliminator.cs:16-35
//   KeypointMatching.MatchKeypoints                           ImageProcessing/KeypointMatching.cs:14-69
#pragma once

#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/pgx.h"

namespace pgx {

struct ArgumentException : std::invalid_argument { using std::invalid_argument::invalid_argument; };
struct IndexOutOfRangeException : std::out_of_range { using std::out_of_range::out_of_range; };
struct ArgumentOutOfRangeException : std::out_of_range { using std::out_of_range::out_of_range; };
struct PgxException : std::runtime_error { using std::runtime_error::runtime_error; };

// Matrix<T> stand-in: row-major storage, (x, y) accessors like the reference's indexer (Matrix.cs:44-76)
template <class T> struct Matrix {
    int Width = 0, Height = 0;
    std::vector<T> data;
    Matrix() = default;
    Matrix(int w, int h, T fill = T()) : Width(w), Height(h), data((size_t)w * h, fill) {}
    T &operator()(int x, int y)
    {
        if (x < 0 || y < 0 || x >= Width || y >= Height) throw IndexOutOfRangeException("Matrix index");
        return data[(size_t)y * Width + x];
    }
    const T &operator()(int x, int y) const { return const_cast<Matrix *>(this)->operator()(x, y); }
};

struct Rgba64 { uint16_t R, G, B, A; };                 // Images.Abstractions/Pixels/Rgba64.cs:3-9
struct Uv { int32_t U, V; };                            // Images.Abstractions/Pixels/Uv.cs:3-7
struct Coordinate { int X, Y; };                        // Math/LinearAlgebra/Coordinate.cs:3-6
using GaussianPair = std::pair<Coordinate, Coordinate>; // KeypointDetection.cs:26

struct Keypoint {                                       // ImageProcessing.Abstractions/Keypoint.cs:11-15
    Coordinate Coordinate_{};
    int FastScore = 0;
    std::vector<uint32_t> BriefDescriptor;              // little-endian words of the BigInteger
    float Value = 0.f;
};
struct KeypointPair { int Keypoint1, Keypoint2, Distance; }; // indices into the two lists (KeypointPair.cs:3-8)

class Context {
  public:
    explicit Context(int device = 0)
    {
        if (pgx_ctx_create(device, &c_) != PGX_OK) throw PgxException("pgx_ctx_create: no usable gfx950 device");
    }
    ~Context() { pgx_ctx_destroy(c_); }
    Context(const Context &) = delete;
    Context &operator=(const Context &) = delete;
    pgx_ctx *get() const { return c_; }
    void check(int rc) const
    {
        if (rc == PGX_OK) return;
        const std::string msg = pgx_last_error(c_);
        switch (rc) {
        case PGX_E_DIM_MISMATCH:
        case PGX_E_BADARG: throw ArgumentException(msg);
        case PGX_E_OOB_SOURCE: throw IndexOutOfRangeException(msg);
        case PGX_E_EMPTY_SET: throw ArgumentOutOfRangeException(msg);
        default: throw PgxException(msg);
        }
    }

  private:
    pgx_ctx *c_ = nullptr;
};

struct DeWarpOptions { int Width, Height; std::vector<double> DistortionCoefficients; };  // Options/DeWarpOptions.cs
struct KeypointDetectionOptions { float Threshold; int GaussianStandardDeviation; int NumGaussianPairs; };
struct RedundantKeypointEliminationOptions { int SuppressionRadius; };

class DeWarp {
  public:
    DeWarp(Context &ctx, DeWarpOptions o) : ctx_(ctx), o_(std::move(o)) {}
    Matrix<Uv> GetDistortionMatrix() const
    {
        if (o_.DistortionCoefficients.size() != 5) throw ArgumentException("You must pass exactly 5 distortion coefficients");
        Matrix<Uv> m(o_.Width, o_.Height);
        if (pgx_build_dewarp_map(o_.Width, o_.Height, o_.DistortionCoefficients.data(), 5,
                                 reinterpret_cast<int32_t *>(m.data.data())) != PGX_OK)
            throw ArgumentException("GetDistortionMatrix");
        return m;
    }
    // the same table built on the device and left there (no upload); ApplyDistortionMatOnDevice then uses it
    void InitializeOnDevice() const
    {
        if (o_.DistortionCoefficients.size() != 5) throw ArgumentException("You must pass exactly 5 distortion coefficients");
        ctx_.check(pgx_set_dewarp_coeffs(ctx_.get(), o_.Width, o_.Height, o_.DistortionCoefficients.data(), 5));
    }
    Matrix<Uv> GetDeviceDistortionMatrix() const
    {
        Matrix<Uv> m(o_.Width, o_.Height);
        ctx_.check(pgx_get_dewarp_map(ctx_.get(), reinterpret_cast<int32_t *>(m.data.data()), o_.Width, o_.Height));
        return m;
    }
    Matrix<Rgba64> ApplyDistortionMat(const Matrix<Rgba64> &image, const Matrix<Uv> &distortionMatrix) const
    {
        ctx_.check(pgx_set_dewarp_map(ctx_.get(), reinterpret_cast<const int32_t *>(distortionMatrix.data.data()),
                                      distortionMatrix.Width, distortionMatrix.Height));
        Matrix<Rgba64> out(image.Width, image.Height);
        ctx_.check(pgx_dewarp(ctx_.get(), reinterpret_cast<const uint16_t *>(image.data.data()), image.Width, image.Height,
                              reinterpret_cast<uint16_t *>(out.data.data())));
        return out;
    }

  private:
    Context &ctx_;
    DeWarpOptions o_;
};

struct Grayscale {
    static Matrix<float> FromRgba64(Context &ctx, const Matrix<Rgba64> &image)
    {
        Matrix<float> out(image.Width, image.Height);
        ctx.check(pgx_gray(ctx.get(), reinterpret_cast<const uint16_t *>(image.data.data()), image.Width, image.Height,
                           out.data.data()));
        return out;
    }
};

class KeypointDetection {
  public:
    // gaussianKeypairs is explicit because the reference draws it from an unseeded Random
    KeypointDetection(Context &ctx, KeypointDetectionOptions o, const std::vector<GaussianPair> &gaussianKeypairs)
        : ctx_(ctx), o_(o)
    {
        std::vector<int32_t> flat;
        for (const auto &p : gaussianKeypairs) { flat.push_back(p.first.X); flat.push_back(p.first.Y); flat.push_back(p.second.X); flat.push_back(p.second.Y); }
        ctx_.check(pgx_set_brief_pairs(ctx_.get(), flat.data(), (int)gaussianKeypairs.size()));
        words_ = ((int)gaussianKeypairs.size() + 31) / 32;
    }
    static std::vector<GaussianPair> MakeGaussianKeypairs(uint64_t seed, const KeypointDetectionOptions &o)
    {
        std::vector<int32_t> flat((size_t)o.NumGaussianPairs * 4);
        pgx_make_brief_pairs(seed, o.GaussianStandardDeviation, o.NumGaussianPairs, flat.data());
        std::vector<GaussianPair> out;
        for (int p = 0; p < o.NumGaussianPairs; p++)
            out.push_back({{flat[4 * p], flat[4 * p + 1]}, {flat[4 * p + 2], flat[4 * p + 3]}});
        return out;
    }
    std::vector<Keypoint> Detect(const Matrix<float> &image) const
    {
        ctx_.check(pgx_set_detect_params(ctx_.get(), o_.Threshold, 0));
        int n = 0;
        std::vector<pgx_keypoint> raw((size_t)image.Width * image.Height + 1);
        ctx_.check(pgx_fast(ctx_.get(), image.data.data(), image.Width, image.Height, raw.data(), (int)raw.size(), &n));
        raw.resize(n);
        std::vector<uint32_t> desc((size_t)n * words_);
        ctx_.check(pgx_brief(ctx_.get(), image.data.data(), image.Width, image.Height, raw.data(), n, desc.data()));
        std::vector<Keypoint> out(n);
        for (int i = 0; i < n; i++) {
            out[i].Coordinate_ = {raw[i].x, raw[i].y};
            out[i].FastScore = raw[i].fast_score;
            out[i].Value = raw[i].value;
            out[i].BriefDescriptor.assign(desc.begin() + (size_t)i * words_, desc.begin() + (size_t)(i + 1) * words_);
        }
        return out;
    }

  private:
    Context &ctx_;
    KeypointDetectionOptions o_;
    int words_ = 0;
};

class RedundantKeypointEliminator {
  public:
    RedundantKeypointEliminator(Context &ctx, RedundantKeypointEliminationOptions o) : ctx_(ctx), r_(o.SuppressionRadius) {}
    std::vector<Keypoint> EliminateRedundantKeypoints(const std::vector<Keypoint> &keypoints, int width, int height) const
    {
        std::vector<pgx_keypoint> in(keypoints.size());
        for (size_t i = 0; i < keypoints.size(); i++)
            in[i] = {keypoints[i].Coordinate_.X, keypoints[i].Coordinate_.Y, keypoints[i].FastScore, keypoints[i].Value};
        std::vector<int32_t> order(keypoints.size() + 1);
        int n = 0;
        ctx_.check(pgx_set_detect_params(ctx_.get(), 0.f, r_));
        ctx_.check(pgx_nms(ctx_.get(), in.data(), (int)in.size(), width, height, order.data(), &n));
        std::vector<Keypoint> out;
        for (int k = 0; k < n; k++) out.push_back(keypoints[order[k]]);
        return out;
    }

  private:
    Context &ctx_;
    int r_;
};

class KeypointMatching {
  public:
    explicit KeypointMatching(Context &ctx) : ctx_(ctx) {}
    std::vector<KeypointPair> MatchKeypoints(const std::vector<Keypoint> &keypoints1, const std::vector<Keypoint> &keypoints2) const
    {
        const int n1 = (int)keypoints1.size(), n2 = (int)keypoints2.size();
        const int words = n1 ? (int)keypoints1[0].BriefDescriptor.size() : (n2 ? (int)keypoints2[0].BriefDescriptor.size() : 8);
        std::vector<uint32_t> d1((size_t)n1 * words), d2((size_t)n2 * words);
        for (int i = 0; i < n1; i++) std::copy(keypoints1[i].BriefDescriptor.begin(), keypoints1[i].BriefDescriptor.end(), d1.begin() + (size_t)i * words);
        for (int i = 0; i < n2; i++) std::copy(keypoints2[i].BriefDescriptor.begin(), keypoints2[i].BriefDescriptor.end(), d2.begin() + (size_t)i * words);
        std::vector<pgx_pair> out((size_t)n1 + 1);
        ctx_.check(pgx_match(ctx_.get(), d1.data(), n1, d2.data(), n2, words, out.data()));
        std::vector<KeypointPair> res(n1);
        for (int i = 0; i < n1; i++) res[i] = {out[i].k1, out[i].k2, out[i].dist};
        return res;
    }

    // The same for many image pairs in one call (pgx_match_batch): frames[f] = a keypoint list, pairs = (frame a, frame b).
    // No reference counterpart (TestService.cs:96 matches one pair); same result per pair as MatchKeypoints.
    std::vector<std::vector<KeypointPair>> MatchKeypointsBatch(const std::vector<std::vector<Keypoint>> &frames,
                                                               const std::vector<std::pair<int, int>> &pairs) const
    {
        const int F = (int)frames.size(), M = (int)pairs.size();
        int words = 8;
        for (const auto &f : frames) if (!f.empty()) { words = (int)f[0].BriefDescriptor.size(); break; }
        std::vector<std::vector<uint32_t>> packed(F);
        std::vector<const uint32_t *> ptrs(F ? F : 1, nullptr);
        std::vector<int32_t> counts(F ? F : 1, 0), pl(2 * (size_t)(M ? M : 1), 0);
        for (int f = 0; f < F; f++) {
            counts[f] = (int32_t)frames[f].size();
            packed[f].resize((size_t)counts[f] * words);
            for (int i = 0; i < counts[f]; i++) {
                if ((int)frames[f][i].BriefDescriptor.size() != words)
                    throw ArgumentException("MatchKeypointsBatch: descriptors of different lengths");
                std::copy(frames[f][i].BriefDescriptor.begin(), frames[f][i].BriefDescriptor.end(), packed[f].begin() + (size_t)i * words);
            }
            ptrs[f] = packed[f].data();
        }
        size_t total = 0;
        for (int m = 0; m < M; m++) {
            if (pairs[m].first < 0 || pairs[m].first >= F || pairs[m].second < 0 || pairs[m].second >= F)
                throw ArgumentException("MatchKeypointsBatch: a pair names a frame outside the list");
            pl[2 * m] = pairs[m].first; pl[2 * m + 1] = pairs[m].second; total += frames[pairs[m].first].size();
        }
        std::vector<pgx_pair> out(total + 1);
        std::vector<int64_t> offs((size_t)M + 1, 0);
        ctx_.check(pgx_match_batch(ctx_.get(), ptrs.data(), counts.data(), F, words, pl.data(), M, out.data(), offs.data()));
        std::vector<std::vector<KeypointPair>> res(M);
        for (int m = 0; m < M; m++)
            for (int64_t i = offs[m]; i < offs[m + 1]; i++) res[m].push_back({out[i].k1, out[i].k2, out[i].dist});
        return res;
    }

  private:
    Context &ctx_;
};

} // namespace pgx
