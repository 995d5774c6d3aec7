// Probe: HBM write bandwidth of the residual-rows kernel's store pattern.  Every wavefront writes 16-byte pieces per lane;
//   mode 0: one contiguous stream over the whole buffer (plain fill)
//   mode 1: the byte-matrix pattern: per "pair" a region with a 4 MiB stride of which 1.5 MiB are used, a store
//           instruction covers 16 rows x 64 bytes (rows 1232 bytes apart)
//   mode 2: the same regions, but a store instruction covers 1 KiB contiguous
//   mode 3: mode 1 with the rows padded to a multiple of 128 bytes (1280)
// build: hipcc -O3 --offload-arch=gfx950 write_pattern.hip -o write_pattern
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(256) void k_write(uint4 *buf, size_t bytes_per_pair_used, size_t pair_stride, int npairs, int Cs)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint4 v = make_uint4(threadIdx.x, blockIdx.x, 3, 4);
    unsigned char *base = reinterpret_cast<unsigned char *>(buf);
    if (MODE == 0) {
        const size_t total = bytes_per_pair_used * npairs;
        for (size_t off = ((size_t)blockIdx.x * 256 + threadIdx.x) * 16; off < total; off += (size_t)gridDim.x * 256 * 16)
            *reinterpret_cast<uint4 *>(base + off) = v;
    } else {
        const int R = 1220;
        // blockIdx.y = pair, blockIdx.x = row block of 256 rows; wave = 64-column block inside a 256-column block
        const int m = blockIdx.y, rb = blockIdx.x * 256;
        unsigned char *D = base + (size_t)m * pair_stride;
        if (rb >= R) return;
        for (int cb = 0; cb < Cs; cb += 256)
            for (int rt = 0; rt < 8; rt++)
                for (int k = 0; k < 2; k++) {
                    if (MODE == 1) {
                        const int rr = k * 16 + (lane >> 2), seg = lane & 3;
                        const int ii = rb + rt * 32 + rr, j0 = cb + wv * 64 + seg * 16;
                        if (ii < R && j0 + 16 <= 1232) *reinterpret_cast<uint4 *>(D + (size_t)ii * Cs + j0) = v;
                    } else {
                        const size_t piece = ((((size_t)blockIdx.x * 8 + rt) * 5 + (cb >> 8)) * 4 + wv) * 2 + k;
                        const size_t off = piece * 1024 + lane * 16;
                        if (off + 16 <= bytes_per_pair_used) *reinterpret_cast<uint4 *>(D + off) = v;
                    }
                }
    }
}

int main()
{
    const int npairs = 256;
    const size_t used = 1232 * 1220, stride = 4u << 20;
    uint4 *buf;
    hipMalloc(&buf, stride * npairs);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 4; mode++) {
        float best = 1e9f;
        for (int rep = 0; rep < 5; rep++) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k_write<0>, dim3(4096), dim3(256), 0, 0, buf, used, stride, npairs, 1232);
            else if (mode == 1) hipLaunchKernelGGL(k_write<1>, dim3(8, npairs), dim3(256), 0, 0, buf, used, stride, npairs, 1232);
            else if (mode == 3) hipLaunchKernelGGL(k_write<1>, dim3(8, npairs), dim3(256), 0, 0, buf, used, stride, npairs, 1280);
            else hipLaunchKernelGGL(k_write<2>, dim3(8, npairs), dim3(256), 0, 0, buf, used, stride, npairs, 1232);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("mode %d: %.1f us for %.0f MB = %.2f TB/s\n", mode, best * 1e3, used * npairs / 1e6, used * npairs / (best * 1e-3) / 1e12);
    }
    return 0;
}
