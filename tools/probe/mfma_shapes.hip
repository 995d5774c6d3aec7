// Probe: how much independent vector work rides beside back-to-back FP4 MFMAs of the two shapes gfx950 offers at the same rate,
//   v_mfma_scale_f32_32x32x64_f8f6f4  (8 passes, reads and writes 16 accumulator registers per instruction)
//   v_mfma_scale_f32_16x16x128_f8f6f4 (4 passes, reads and writes  4 accumulator registers per instruction)
// per "tile" = the work of k_ham_fp4's loop step (3 row tiles x 32 columns x 256 bits = 12 or 24 instructions, 384 matrix-pipe
// cycles either way), two waves per SIMD, NV independent vector instructions per tile that touch no accumulator.
// Build: hipcc -O3 --offload-arch=gfx950 mfma_shapes.hip -o mfma_shapes ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
using i32x8 = int __attribute__((ext_vector_type(8)));
using f32x16 = float __attribute__((ext_vector_type(16)));
using f32x4 = float __attribute__((ext_vector_type(4)));

template <int SHAPE, int NV>
__global__ __launch_bounds__(256, 2) void loop(const uint32_t *__restrict__ desc, int ntile, int *out, long long *cyc)
{
    const int tid = threadIdx.x;
    int a[3][4][4], b[4][4];
    for (int t = 0; t < 3; t++) for (int s = 0; s < 4; s++) for (int q = 0; q < 4; q++) a[t][s][q] = (int)(desc[(tid * 7 + t * 16 + s * 4 + q) & 4095] & 0x88888888u) | 0x22222222;
    for (int s = 0; s < 4; s++) for (int q = 0; q < 4; q++) b[s][q] = a[0][s][q] ^ 0x08080808;
    int x[16], y[16];
    for (int g = 0; g < 16; g++) { x[g] = tid * 31 + g; y[g] = tid * 17 - g; }
    int sum = 0;
    long long t0, t1;
    if (SHAPE == 32) {
        f32x16 acc[3];
        for (int t = 0; t < 3; t++) for (int g = 0; g < 16; g++) acc[t][g] = (float)g;
        t0 = wall_clock64();
        for (int ct = 0; ct < ntile; ct++) {
#pragma unroll
            for (int s = 0; s < 4; s++)
#pragma unroll
                for (int t = 0; t < 3; t++) {
                    const i32x8 av = {a[t][s][0], a[t][s][1], a[t][s][2], a[t][s][3], 0, 0, 0, 0};
                    const i32x8 bv = {b[s][0], b[s][1], b[s][2], b[s][3], 0, 0, 0, 0};
                    acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, acc[t], 4, 4, 0, 127, 0, 127);
                }
#pragma unroll
            for (int k = 0; k < NV; k++) { x[k & 15] = max(x[k & 15], y[(k * 5 + 3) & 15] + ct); y[k & 15] ^= x[(k + 7) & 15]; }
        }
        t1 = wall_clock64();
        for (int t = 0; t < 3; t++) for (int g = 0; g < 16; g++) sum += __float_as_int(acc[t][g]);
    } else {
        // 3 row tiles x 32 columns = 6 x 2 blocks of 16 x 16, two k-steps of 128 bits each: 24 instructions, 12 accumulators of 4 registers
        f32x4 acc[12];
        for (int t = 0; t < 12; t++) for (int g = 0; g < 4; g++) acc[t][g] = (float)g;
        t0 = wall_clock64();
        for (int ct = 0; ct < ntile; ct++) {
#pragma unroll
            for (int s = 0; s < 2; s++)
#pragma unroll
                for (int t = 0; t < 12; t++) {
                    const i32x8 av = {a[t % 3][2 * s + t / 6][0], a[t % 3][2 * s + t / 6][1], a[t % 3][2 * s + t / 6][2], a[t % 3][2 * s + t / 6][3], 0, 0, 0, 0};
                    const i32x8 bv = {b[2 * s + (t & 1)][0], b[2 * s + (t & 1)][1], b[2 * s + (t & 1)][2], b[2 * s + (t & 1)][3], 0, 0, 0, 0};
                    acc[t] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, acc[t], 4, 4, 0, 127, 0, 127);
                }
#pragma unroll
            for (int k = 0; k < NV; k++) { x[k & 15] = max(x[k & 15], y[(k * 5 + 3) & 15] + ct); y[k & 15] ^= x[(k + 7) & 15]; }
        }
        t1 = wall_clock64();
        for (int t = 0; t < 12; t++) for (int g = 0; g < 4; g++) sum += __float_as_int(acc[t][g]);
    }
    for (int g = 0; g < 16; g++) sum += x[g] + y[g];
    out[blockIdx.x * blockDim.x + tid] = sum;
    if (tid == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

int main()
{
    uint32_t *dD; int *dO; long long *dC, hC;
    hipMalloc(&dD, 4096 * 32 + 64); hipMemset(dD, 0x5a, 4096 * 32 + 64);
    hipMalloc(&dO, 512 * 256 * 4); hipMalloc(&dC, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int ntile = 4096;
    auto run = [&](const char *name, auto kern, int nv, int blocks) {
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, dD, ntile, dO, dC); hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, dD, ntile, dO, dC);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(&hC, dC, 8, hipMemcpyDeviceToHost);
        const double ops = (double)ntile * blocks * 4 * 12 * 2.0 * 32 * 32 * 64;
        printf("%-12s V = %3d  %d waves/SIMD: %7.3f ms  %6.1f ns per tile and wave  %.2f of 10 POP/s\n", name, nv * 3, blocks / 256, ms, ms * 1e6 / ntile,
               ops / (ms * 1e-3) / 1e16);
    };
    for (int blocks : {256, 512}) {
        run("32x32x64", loop<32, 0>, 0, blocks);  run("16x16x128", loop<16, 0>, 0, blocks);
        run("32x32x64", loop<32, 16>, 16, blocks); run("16x16x128", loop<16, 16>, 16, blocks);
        run("32x32x64", loop<32, 24>, 24, blocks); run("16x16x128", loop<16, 24>, 24, blocks);
        run("32x32x64", loop<32, 32>, 32, blocks); run("16x16x128", loop<16, 32>, 32, blocks);
        run("32x32x64", loop<32, 40>, 40, blocks); run("16x16x128", loop<16, 40>, 40, blocks);
        run("32x32x64", loop<32, 48>, 48, blocks); run("16x16x128", loop<16, 48>, 48, blocks);
    }
    return 0;
}
