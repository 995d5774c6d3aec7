// k_image.hip -- fused DeWarp.ApplyDistortionMat + Grayscale.FromRgba64 for gfx950.
//
// Reference: ImageProcessing/DeWarp.cs:19-37 (integer gather through the Matrix<Uv> table) and
// Images.Abstractions/Pixels/Grayscale.cs:19-23 (K = ((float)R + B + G) / (3*65535), float32).
// HBM-bound: per pixel and frame 8 B gathered source (4 B for 8-bit sources, widened in registers) + 4 B grey, plus 8 B of map per pixel and group of
// FB = 4 frames (the map is the same for every frame of a batch): 14 B/px/frame; FAST re-reads the 4 B.
// Layout: every thread owns 4 consecutive output pixels of FB frames -> two 16-B map loads, then per
// frame four 8-B gathers and one 16-B grey store.
#include "pgx_internal.h"

namespace {

__device__ __forceinline__ float gray_of(uint2 px)
{
    // px.x = R | G<<16, px.y = B | A<<16.  Sum order R, B, G as the C# writes it; the sum is
    // exact (< 2^24) and the division is one correctly rounded float32 divide.
    float r = (float)(px.x & 0xFFFFu), g = (float)(px.x >> 16), b = (float)(px.y & 0xFFFFu);
    float s = __fadd_rn(__fadd_rn(r, b), g);
    return __fdiv_rn(s, 196605.0f);
}

constexpr int FB = 4; // frames per thread: one map read serves FB frames

// 8-bit sources (PGX_SRC_RGBA8): a channel c becomes c * 257 = c | c << 8, the scaling an 8-bit image gets when it is
// loaded as Rgba64 (LocalImageReader.cs:22 via ImageSharp; 255 -> 65535).  Bytes r,g,b,a -> words r,r,g,g | b,b,a,a.
__device__ __forceinline__ uint2 widen8(uint32_t v)
{
    return make_uint2(__builtin_amdgcn_perm(v, v, 0x01010000u), __builtin_amdgcn_perm(v, v, 0x03030202u));
}

template <bool SRC8> struct SrcPix { using type = uint2; };
template <> struct SrcPix<true> { using type = uint32_t; };
template <bool SRC8> __device__ __forceinline__ uint2 load_px(const typename SrcPix<SRC8>::type *p);
template <> __device__ __forceinline__ uint2 load_px<false>(const uint2 *p) { return *p; }
template <> __device__ __forceinline__ uint2 load_px<true>(const uint32_t *p) { return widen8(*p); }

template <bool HAS_MAP, bool WRITE_RGBA, bool SRC8>
__global__ __launch_bounds__(256) void k_dewarp_gray(const typename SrcPix<SRC8>::type *__restrict__ rgba,
                                                     const int2 *__restrict__ map, int W, int H, int F,
                                                     float *__restrict__ gray, uint2 *__restrict__ dewarped, int *status)
{
    using Px = typename SrcPix<SRC8>::type;
    const size_t npix = (size_t)W * H;
    const int f0 = blockIdx.y * FB;
    const int nf = F - f0 < FB ? F - f0 : FB; // block-uniform
    const size_t ngroups = (npix + 3) / 4;
    // 16-byte accesses at frame f's base need f * npix to be a multiple of 4 pixels: odd frame sizes in a batch
    // take the scalar path (block-uniform; the same for 4-byte source pixels, where 4 pixels are one 16-byte load)
    const bool vec = (npix & 3) == 0 || F == 1;
    bool oob = false;
    for (size_t grp = (size_t)blockIdx.x * blockDim.x + threadIdx.x; grp < ngroups;
         grp += (size_t)gridDim.x * blockDim.x) {
        const size_t i0 = grp * 4;
        if (vec && i0 + 3 < npix) {
            size_t so[4]; // source offsets of the four output pixels, the same in every frame
            bool ok[4];
            if (HAS_MAP) {
                const int4 m01 = *reinterpret_cast<const int4 *>(map + i0);
                const int4 m23 = *reinterpret_cast<const int4 *>(map + i0 + 2);
                const int us[4] = {m01.x, m01.z, m23.x, m23.z};
                const int vs[4] = {m01.y, m01.w, m23.y, m23.w};
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    // (ushort) casts are unchecked in the C#: value mod 65536, then the bounds assert
                    const unsigned su = (unsigned)us[k] & 0xFFFFu, sv = (unsigned)vs[k] & 0xFFFFu;
                    ok[k] = su < (unsigned)W && sv < (unsigned)H;
                    oob |= !ok[k];
                    so[k] = ok[k] ? (size_t)sv * W + su : 0;
                }
            }
            uint2 px[FB][4];
#pragma unroll
            for (int fb = 0; fb < FB; fb++) {
                if (fb >= nf) break;
                const Px *src = rgba + (size_t)(f0 + fb) * npix;
                if (HAS_MAP) {
#pragma unroll
                    for (int k = 0; k < 4; k++) px[fb][k] = load_px<SRC8>(src + so[k]);
                } else if constexpr (SRC8) {
                    const uint4 a = *reinterpret_cast<const uint4 *>(src + i0);
                    px[fb][0] = widen8(a.x); px[fb][1] = widen8(a.y); px[fb][2] = widen8(a.z); px[fb][3] = widen8(a.w);
                } else {
                    const uint4 a = *reinterpret_cast<const uint4 *>(src + i0);
                    const uint4 b = *reinterpret_cast<const uint4 *>(src + i0 + 2);
                    px[fb][0] = make_uint2(a.x, a.y); px[fb][1] = make_uint2(a.z, a.w);
                    px[fb][2] = make_uint2(b.x, b.y); px[fb][3] = make_uint2(b.z, b.w);
                }
            }
#pragma unroll
            for (int fb = 0; fb < FB; fb++) {
                if (fb >= nf) break;
                if (HAS_MAP) {
#pragma unroll
                    for (int k = 0; k < 4; k++) if (!ok[k]) px[fb][k] = make_uint2(0, 0);
                }
                if (gray) {
                    const float4 g4 = make_float4(gray_of(px[fb][0]), gray_of(px[fb][1]), gray_of(px[fb][2]), gray_of(px[fb][3]));
                    *reinterpret_cast<float4 *>(gray + (size_t)(f0 + fb) * npix + i0) = g4;
                }
                if (WRITE_RGBA) {
                    uint2 *dout = dewarped + (size_t)(f0 + fb) * npix;
                    *reinterpret_cast<uint4 *>(dout + i0) = make_uint4(px[fb][0].x, px[fb][0].y, px[fb][1].x, px[fb][1].y);
                    *reinterpret_cast<uint4 *>(dout + i0 + 2) = make_uint4(px[fb][2].x, px[fb][2].y, px[fb][3].x, px[fb][3].y);
                }
            }
        } else {
            for (size_t i = i0; i < npix && i < i0 + 4; i++) {
                size_t o = i;
                bool okk = true;
                if (HAS_MAP) {
                    const int2 m = map[i];
                    const unsigned su = (unsigned)m.x & 0xFFFFu, sv = (unsigned)m.y & 0xFFFFu;
                    okk = su < (unsigned)W && sv < (unsigned)H;
                    oob |= !okk;
                    o = okk ? (size_t)sv * W + su : 0;
                }
                for (int fb = 0; fb < nf; fb++) {
                    const uint2 p = okk ? load_px<SRC8>(rgba + (size_t)(f0 + fb) * npix + o) : make_uint2(0, 0);
                    if (gray) gray[(size_t)(f0 + fb) * npix + i] = gray_of(p);
                    if (WRITE_RGBA) dewarped[(size_t)(f0 + fb) * npix + i] = p;
                }
            }
        }
    }
    if (HAS_MAP && oob) atomicOr(status, (int)PGX_ST_OOB_SOURCE);
}

// ---- DeWarp.GetDistortionMatrix on the device ------------------------------------------
// Reference: ImageProcessing/DeWarp.cs:39-107, same float64 formulas as the host builder
// (pgx_hostutil.cpp): x = (int)(u - W/2.0), rd = sqrt(x^2+y^2), cubic r^3 + b r^2 + c r + d with
// MathNet's Cubic.RealRoots restated, middle root of three else the smallest, theta = atan2(y, x),
// (U, V) = ((int)(root cos theta + W/2.0), (int)(root sin theta + H/2.0)).  One thread per pixel; the
// reference caches the root per r^2, which changes nothing in the values.  The device's libm (pow, acos,
// cos, sin, atan2) is not bit-identical to the host's, so a truncation can land on the other side of an
// integer: parity with the host table is "equal, except +-1 at isolated pixels" (tested), unpinned
// against the reference like the host builder itself.
__device__ __forceinline__ int32_t trunc_to_int_dev(double v)
{
    if (!(v > -2147483649.0 && v < 2147483648.0)) return INT32_MIN;
    return (int32_t)v;
}

__device__ __forceinline__ double cbrt_signed_dev(double n)
{
    const double sgn = (double)((n > 0) - (n < 0));
    return pow(fabs(n), 1.0 / 3.0) * sgn;
}

__global__ __launch_bounds__(256) void k_dewarp_map(int W, int H, double k0, double k1, double k2, double k3, double k4,
                                                    int2 *__restrict__ map, int *status)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)W * H) return;
    const int v = (int)(i / W), u = (int)(i - (size_t)v * W);
    const double x0 = W / 2.0, y0 = H / 2.0;
    const int x = trunc_to_int_dev(u - x0), y = trunc_to_int_dev(v - y0);
    const int rd2 = x * x + y * y;
    const double rd = sqrt((double)rd2);
    const double den = rd * k4 - k1;
    const double a2 = (rd * k3 - k0) / den, a1 = (rd * k2 - 1) / den, a0 = rd / den;
    const double Q = (3 * a1 - a2 * a2) / 9.0;
    const double R = (9.0 * a2 * a1 - 27 * a0 - 2 * a2 * a2 * a2) / 54.0;
    const double Q3 = Q * Q * Q;
    const double D = Q3 + R * R;
    const double shift = -a2 / 3.0;
    const double nan = __longlong_as_double(0x7FF8000000000000ll);
    double r0 = nan, r1 = nan, r2 = nan;
    if (D >= 0) {
        const double sqrtD = pow(D, 0.5);
        const double S = cbrt_signed_dev(R + sqrtD), T = cbrt_signed_dev(R - sqrtD);
        r0 = shift + (S + T);
        if (D == 0) r1 = shift - S;
    } else {
        const double pi = 3.1415926535897932384626433832795;
        const double theta = acos(R / sqrt(-Q3));
        const double m = 2.0 * sqrt(-Q);
        r0 = m * cos(theta / 3.0) + shift;
        r1 = m * cos((theta + 2.0 * pi) / 3.0) + shift;
        r2 = m * cos((theta - 2.0 * pi) / 3.0) + shift;
    }
    // drop NaN, sort ascending, middle of three else the smallest (DeWarp.cs:78-82)
    double kept[3];
    int n = 0;
    if (!isnan(r0)) kept[n++] = r0;
    if (!isnan(r1)) kept[n++] = r1;
    if (!isnan(r2)) kept[n++] = r2;
    if (n == 0) { atomicOr(status, (int)PGX_ST_INTERNAL); map[i] = make_int2(INT32_MIN, INT32_MIN); return; }
    if (n >= 2 && kept[0] > kept[1]) { const double t = kept[0]; kept[0] = kept[1]; kept[1] = t; }
    if (n == 3) {
        if (kept[1] > kept[2]) { const double t = kept[1]; kept[1] = kept[2]; kept[2] = t; }
        if (kept[0] > kept[1]) { const double t = kept[0]; kept[0] = kept[1]; kept[1] = t; }
    }
    const double root = n == 3 ? kept[1] : kept[0];
    const double theta = atan2((double)y, (double)x);
    map[i] = make_int2(trunc_to_int_dev(root * cos(theta) + x0), trunc_to_int_dev(root * sin(theta) + y0));
}

} // namespace

void pgx_launch_dewarp_gray(hipStream_t s, const void *rgba, int src8, const int32_t *map_uv, int F, int W, int H,
                            float *gray, uint16_t *dewarped, int *status)
{
    if (F <= 0 || W <= 0 || H <= 0) return;
    const size_t ngroups = ((size_t)W * H + 3) / 4;
    unsigned gx = (unsigned)((ngroups + 255) / 256);
    // About 512 workgroups per launch (two per CU), grid-stride beyond: a streaming kernel needs that many to keep HBM busy and
    // no more -- with one workgroup per 1024 pixels (32 400 of them for 64 frames of 1920x1080) the launch was 5 % slower alone,
    // and beside another job's distance kernel the flood of small workgroups took every wave slot that came free (two jobs in
    // flight: 7.19 -> 7.12 ms per bench step with the bounded grid).
    const unsigned ny = (unsigned)((F + FB - 1) / FB);
    const unsigned cap = 512u / ny > 32u ? 512u / ny : 32u;
    if (gx > cap) gx = cap;
    dim3 grid(gx, (unsigned)((F + FB - 1) / FB)), block(256);
    const int2 *map = reinterpret_cast<const int2 *>(map_uv);
    uint2 *dw = reinterpret_cast<uint2 *>(dewarped);
#define PGX_DG(M, WR, S8) hipLaunchKernelGGL((k_dewarp_gray<M, WR, S8>), grid, block, 0, s, \
                                             reinterpret_cast<const SrcPix<S8>::type *>(rgba), map, W, H, F, gray, dw, status)
    if (src8) {
        if (map) { if (dw) PGX_DG(true, true, true); else PGX_DG(true, false, true); }
        else { if (dw) PGX_DG(false, true, true); else PGX_DG(false, false, true); }
    } else {
        if (map) { if (dw) PGX_DG(true, true, false); else PGX_DG(true, false, false); }
        else { if (dw) PGX_DG(false, true, false); else PGX_DG(false, false, false); }
    }
#undef PGX_DG
}

void pgx_launch_dewarp_map(hipStream_t s, int W, int H, const double *k, int32_t *map_uv, int *status)
{
    const size_t npix = (size_t)W * H;
    hipLaunchKernelGGL(k_dewarp_map, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, s, W, H, k[0], k[1], k[2], k[3], k[4],
                       reinterpret_cast<int2 *>(map_uv), status);
}
