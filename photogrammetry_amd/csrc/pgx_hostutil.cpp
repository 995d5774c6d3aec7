// pgx_hostutil.cpp -- init-time host helpers of libpgx.so (no GPU work).
//
// pgx_make_brief_pairs : Utils.NextGaussianPair / NextGaussianCoordinate
//                        (ImageProcessing/Utils.cs:14-38) on a seeded stream.  The reference
//                        draws from an unseeded System.Random, so its table cannot be
//                        reproduced; the formula (Marsaglia polar on y1,y2 in [0,1), truncation)
//                        is kept so synthetic tables have the reference's shape: all offsets >= 0.
// pgx_build_dewarp_map : DeWarp.GetDistortionMatrix (ImageProcessing/DeWarp.cs:39-107) in float64.
//                        MathNet.Numerics 5.0.0 Cubic.RealRoots (third party, absent) is restated
//                        from its published algorithm: parity unpinned, see DESIGN.md.
#include <cmath>
#include <cstdint>
#include <limits>

#include "../../include/pgx.h"

namespace {

struct SplitMix64 {
    uint64_t s;
    uint64_t next()
    {
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    double next_double() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
};

void gaussian_coordinate(SplitMix64 &rng, int sigma, int32_t *xy)
{
    double y1, y2, r2;
    do {
        y1 = rng.next_double();
        y2 = rng.next_double();
        r2 = y1 * y1 + y2 * y2;
    } while (r2 >= 1);
    const double s = std::sqrt(-2 * std::log(r2) / r2);
    xy[0] = (int32_t)(s * y1 * sigma);
    xy[1] = (int32_t)(s * y2 * sigma);
}

int32_t trunc_to_int(double v)
{
    if (!(v > -2147483649.0 && v < 2147483648.0)) return std::numeric_limits<int32_t>::min();
    return (int32_t)v;
}

double cbrt_signed(double n)
{
    const double sgn = (n > 0) - (n < 0);
    return std::pow(std::fabs(n), 1.0 / 3.0) * sgn;
}

// real roots of x^3 + a2 x^2 + a1 x + a0; unused slots are NaN
void real_roots(double a0, double a1, double a2, double r[3])
{
    const double Q = (3 * a1 - a2 * a2) / 9.0;
    const double R = (9.0 * a2 * a1 - 27 * a0 - 2 * a2 * a2 * a2) / 54.0;
    const double Q3 = Q * Q * Q;
    const double D = Q3 + R * R;
    const double shift = -a2 / 3.0;
    const double nan = std::numeric_limits<double>::quiet_NaN();
    r[0] = r[1] = r[2] = nan;
    if (D >= 0) {
        const double sqrtD = std::pow(D, 0.5);
        const double S = cbrt_signed(R + sqrtD), T = cbrt_signed(R - sqrtD);
        r[0] = shift + (S + T);
        if (D == 0) r[1] = shift - S;
    } else {
        const double pi = 3.1415926535897932384626433832795;
        const double theta = std::acos(R / std::sqrt(-Q3));
        const double m = 2.0 * std::sqrt(-Q);
        r[0] = m * std::cos(theta / 3.0) + shift;
        r[1] = m * std::cos((theta + 2.0 * pi) / 3.0) + shift;
        r[2] = m * std::cos((theta - 2.0 * pi) / 3.0) + shift;
    }
}

} // namespace

extern "C" int pgx_make_brief_pairs(uint64_t seed, int sigma, int P, int32_t *out)
{
    if (!out || P < 0) return PGX_E_BADARG;
    SplitMix64 rng{seed};
    for (int p = 0; p < P; p++) {
        gaussian_coordinate(rng, sigma, out + 4 * p);
        gaussian_coordinate(rng, sigma, out + 4 * p + 2);
    }
    return PGX_OK;
}

extern "C" int pgx_build_dewarp_map(int W, int H, const double *k, int ncoeffs, int32_t *out)
{
    if (!k || !out || W <= 0 || H <= 0) return PGX_E_BADARG;
    if (ncoeffs != 5) return PGX_E_BADARG; // DeWarp.cs:46-48
    const double x0 = W / 2.0, y0 = H / 2.0;
    for (int v = 0; v < H; v++) {
        for (int u = 0; u < W; u++) {
            const int x = trunc_to_int(u - x0), y = trunc_to_int(v - y0);
            const int rd2 = x * x + y * y;
            const double rd = std::sqrt((double)rd2);
            const double den = rd * k[4] - k[1];
            const double b = (rd * k[3] - k[0]) / den, c = (rd * k[2] - 1) / den, d = rd / den;
            double roots[3], kept[3];
            real_roots(d, c, b, roots);
            int n = 0;
            for (double r : roots) if (!std::isnan(r)) kept[n++] = r;
            if (n == 0) return PGX_E_BADARG;
            for (int i = 1; i < n; i++)
                for (int j = i; j > 0 && kept[j - 1] > kept[j]; j--) { double t = kept[j]; kept[j] = kept[j - 1]; kept[j - 1] = t; }
            const double root = n == 3 ? kept[1] : kept[0];
            const double theta = std::atan2((double)y, (double)x);
            out[((size_t)v * W + u) * 2 + 0] = trunc_to_int(root * std::cos(theta) + x0);
            out[((size_t)v * W + u) * 2 + 1] = trunc_to_int(root * std::sin(theta) + y0);
        }
    }
    return PGX_OK;
}
