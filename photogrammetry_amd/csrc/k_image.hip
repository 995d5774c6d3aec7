// k_image.hip -- fused DeWarp.ApplyDistortionMat + Grayscale.FromRgba64 for gfx950.
//
// Reference: ImageProcessing/DeWarp.cs:19-37 (integer gather through the Matrix<Uv> table) and
// Images.Abstractions/Pixels/Grayscale.cs:19-23 (K = ((float)R + B + G) / (3*65535), float32).
// HBM-bound: per pixel 8 B map + 8 B gathered source + 4 B grey (20 B; FAST re-reads the 4 B).
// Layout: every thread owns 4 consecutive output pixels -> two 16-B map loads, four 8-B
// gathers, one 16-B grey store; the map is shared by all frames of a batch (blockIdx.y).
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

template <bool HAS_MAP, bool WRITE_RGBA>
__global__ __launch_bounds__(256) void k_dewarp_gray(const uint2 *__restrict__ rgba, const int2 *__restrict__ map,
                                                     int W, int H, float *__restrict__ gray,
                                                     uint2 *__restrict__ dewarped, int *status)
{
    const size_t npix = (size_t)W * H;
    const size_t f = blockIdx.y;
    const uint2 *src = rgba + f * npix;
    float *gout = gray ? gray + f * npix : nullptr;
    uint2 *dout = WRITE_RGBA ? dewarped + f * npix : nullptr;
    const size_t ngroups = (npix + 3) / 4;
    bool oob = false;
    for (size_t grp = (size_t)blockIdx.x * blockDim.x + threadIdx.x; grp < ngroups;
         grp += (size_t)gridDim.x * blockDim.x) {
        const size_t i0 = grp * 4;
        uint2 px[4];
        if (i0 + 3 < npix) {
            if (HAS_MAP) {
                const int4 m01 = *reinterpret_cast<const int4 *>(map + i0);
                const int4 m23 = *reinterpret_cast<const int4 *>(map + i0 + 2);
                const int us[4] = {m01.x, m01.z, m23.x, m23.z};
                const int vs[4] = {m01.y, m01.w, m23.y, m23.w};
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    // (ushort) casts are unchecked in the C#: value mod 65536, then the bounds assert
                    unsigned su = (unsigned)us[k] & 0xFFFFu, sv = (unsigned)vs[k] & 0xFFFFu;
                    bool ok = su < (unsigned)W && sv < (unsigned)H;
                    oob |= !ok;
                    px[k] = ok ? src[(size_t)sv * W + su] : make_uint2(0, 0);
                }
            } else {
                const uint4 a = *reinterpret_cast<const uint4 *>(src + i0);
                const uint4 b = *reinterpret_cast<const uint4 *>(src + i0 + 2);
                px[0] = make_uint2(a.x, a.y); px[1] = make_uint2(a.z, a.w);
                px[2] = make_uint2(b.x, b.y); px[3] = make_uint2(b.z, b.w);
            }
            if (gout) {
                float4 g4 = make_float4(gray_of(px[0]), gray_of(px[1]), gray_of(px[2]), gray_of(px[3]));
                *reinterpret_cast<float4 *>(gout + i0) = g4;
            }
            if (WRITE_RGBA) {
                *reinterpret_cast<uint4 *>(dout + i0) = make_uint4(px[0].x, px[0].y, px[1].x, px[1].y);
                *reinterpret_cast<uint4 *>(dout + i0 + 2) = make_uint4(px[2].x, px[2].y, px[3].x, px[3].y);
            }
        } else {
            for (size_t i = i0; i < npix; i++) {
                uint2 p;
                if (HAS_MAP) {
                    int2 m = map[i];
                    unsigned su = (unsigned)m.x & 0xFFFFu, sv = (unsigned)m.y & 0xFFFFu;
                    bool ok = su < (unsigned)W && sv < (unsigned)H;
                    oob |= !ok;
                    p = ok ? src[(size_t)sv * W + su] : make_uint2(0, 0);
                } else {
                    p = src[i];
                }
                if (gout) gout[i] = gray_of(p);
                if (WRITE_RGBA) dout[i] = p;
            }
        }
    }
    if (HAS_MAP && oob) atomicOr(status, (int)PGX_ST_OOB_SOURCE);
}

} // namespace

void pgx_launch_dewarp_gray(hipStream_t s, const uint16_t *rgba, const int32_t *map_uv, int F, int W, int H,
                            float *gray, uint16_t *dewarped, int *status)
{
    if (F <= 0 || W <= 0 || H <= 0) return;
    const size_t ngroups = ((size_t)W * H + 3) / 4;
    unsigned gx = (unsigned)((ngroups + 255) / 256);
    if (gx > 4096u) gx = 4096u; // >> 256 CUs, grid-stride beyond
    dim3 grid(gx, (unsigned)F), block(256);
    const uint2 *src = reinterpret_cast<const uint2 *>(rgba);
    const int2 *map = reinterpret_cast<const int2 *>(map_uv);
    uint2 *dw = reinterpret_cast<uint2 *>(dewarped);
    if (map) {
        if (dw) hipLaunchKernelGGL((k_dewarp_gray<true, true>), grid, block, 0, s, src, map, W, H, gray, dw, status);
        else hipLaunchKernelGGL((k_dewarp_gray<true, false>), grid, block, 0, s, src, map, W, H, gray, dw, status);
    } else {
        if (dw) hipLaunchKernelGGL((k_dewarp_gray<false, true>), grid, block, 0, s, src, map, W, H, gray, dw, status);
        else hipLaunchKernelGGL((k_dewarp_gray<false, false>), grid, block, 0, s, src, map, W, H, gray, dw, status);
    }
}
