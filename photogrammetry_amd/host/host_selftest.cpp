// host_selftest.cpp -- drives the compiled-host mirror (pgx_host.hpp) through the whole chain on a
// deterministic image pair and prints every result as text; tests/test_gpu_host_mirror.py rebuilds the
// same inputs in numpy and checks this output against the CPU oracle.  Reads like the reference's
// TestService.TestKeypointMatching (Photogrammetry/TestService.cs:80-96).
#include <cstdio>
#include <cstring>

#include "pgx_host.hpp"

using namespace pgx;

static uint32_t lcg(uint32_t &s) { s = s * 1664525u + 1013904223u; return s >> 8; }

static Matrix<Rgba64> make_image(int W, int H, uint32_t seed, int shift)
{
    // blocky random image: 4x4-pixel cells of 4 grey levels, shifted right by `shift` pixels
    Matrix<Rgba64> img(W, H);
    const int cw = (W + shift) / 4 + 2;
    std::vector<uint16_t> cells((size_t)cw * (H / 4 + 2));
    for (auto &c : cells) c = (uint16_t)((lcg(seed) & 3u) * 21845u);
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            const uint16_t v = cells[(size_t)(y / 4) * cw + (x + shift) / 4];
            img(x, y) = Rgba64{v, v, v, 65535};
        }
    return img;
}

int main()
{
    const int W = 240, H = 180;
    try {
        Context ctx(0);
        DeWarp deWarp(ctx, DeWarpOptions{W, H, {3e-4, 1e-7, 0, 0, 0}});
        const Matrix<Uv> map = deWarp.GetDistortionMatrix();
        const KeypointDetectionOptions kdo{0.1f, 20, 256};
        KeypointDetection detection(ctx, kdo, KeypointDetection::MakeGaussianKeypairs(5, kdo));
        RedundantKeypointEliminator eliminator(ctx, RedundantKeypointEliminationOptions{5});
        KeypointMatching matching(ctx);

        std::vector<Keypoint> kept[2];
        for (int k = 0; k < 2; k++) {
            const Matrix<Rgba64> image = make_image(W, H, 77u, k * 8);
            const Matrix<Rgba64> dewarped = deWarp.ApplyDistortionMat(image, map);
            const Matrix<float> gray = Grayscale::FromRgba64(ctx, dewarped);
            const std::vector<Keypoint> raw = detection.Detect(gray);
            kept[k] = eliminator.EliminateRedundantKeypoints(raw, W, H);
            std::printf("image %d raw %zu kept %zu\n", k, raw.size(), kept[k].size());
            for (const Keypoint &p : kept[k]) {
                uint32_t vbits;
                std::memcpy(&vbits, &p.Value, 4);
                std::printf("kp %d %d %d %08x", p.Coordinate_.X, p.Coordinate_.Y, p.FastScore, vbits);
                for (uint32_t w : p.BriefDescriptor) std::printf(" %08x", w);
                std::printf("\n");
            }
        }
        const std::vector<KeypointPair> pairs = matching.MatchKeypoints(kept[0], kept[1]);
        for (const KeypointPair &p : pairs) std::printf("pair %d %d %d\n", p.Keypoint1, p.Keypoint2, p.Distance);
        // both directions and a self match in one batched call (pgx_match_batch)
        const std::vector<std::vector<KeypointPair>> lists = matching.MatchKeypointsBatch({kept[0], kept[1]}, {{0, 1}, {1, 0}, {1, 1}});
        for (size_t m = 0; m < lists.size(); m++)
            for (const KeypointPair &p : lists[m]) std::printf("batch %zu %d %d %d\n", m, p.Keypoint1, p.Keypoint2, p.Distance);

        // error behaviour of the reference, mapped back to exception types
        int errors = 0;
        try { deWarp.ApplyDistortionMat(make_image(W - 1, H, 1u, 0), map); } catch (const ArgumentException &) { errors |= 1; }
        Matrix<Uv> bad = map;
        bad(3, 3) = Uv{-1, 2};
        try { deWarp.ApplyDistortionMat(make_image(W, H, 1u, 0), bad); } catch (const IndexOutOfRangeException &) { errors |= 2; }
        try { matching.MatchKeypoints(kept[0], {}); } catch (const ArgumentOutOfRangeException &) { errors |= 4; }
        try { DeWarp(ctx, DeWarpOptions{W, H, {1, 2, 3}}).GetDistortionMatrix(); } catch (const ArgumentException &) { errors |= 8; }
        std::printf("exceptions %d\n", errors);
        return errors == 15 ? 0 : 2;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "host_selftest failed: %s\n", e.what());
        return 1;
    }
}
