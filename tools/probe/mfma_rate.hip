// Probe: issue rate of v_mfma_scale_f32_32x32x64_f8f6f4 (fp4 operands) against v_mfma_i32_32x32x32_i8, with 1, 2 and 4
// independent accumulator chains per wave; one wave per SIMD (grid = 1 block of 256 threads per CU is enough: per-wave cycles).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
using i32x8 = int __attribute__((ext_vector_type(8)));
using i32x4 = int __attribute__((ext_vector_type(4)));
using f32x16 = float __attribute__((ext_vector_type(16)));
using i32x16 = int __attribute__((ext_vector_type(16)));

template <int NCH>
__global__ void fp4_rate(float *out, long long *cyc, int iters)
{
    i32x8 a = {0x22222222, 0x2a2a2a2a, 0x22aa22aa, 0x2222aaaa, 0, 0, 0, 0}, b = a;
    f32x16 acc[NCH];
    for (int c = 0; c < NCH; c++) for (int g = 0; g < 16; g++) acc[c][g] = (float)(c + g);
    const long long t0 = clock64();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int c = 0; c < NCH; c++) acc[c] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc[c], 4, 4, 0, 127, 0, 127);
    }
    const long long t1 = clock64();
    float s = 0;
    for (int c = 0; c < NCH; c++) for (int g = 0; g < 16; g++) s += acc[c][g];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

// the distance kernel's tile step without its maxima: 4 words expanded to 16 fp4 dwords (28 VALU), 2 chains x 4 MFMAs.
// AHEAD = true: the fragments of the NEXT step are expanded while this step's MFMAs run (one step of software pipelining).
template <bool AHEAD>
__global__ void fp4_step_rate(float *out, long long *cyc, int iters, const uint32_t *src)
{
    int km = (int)0x88888888, kc = 0x22222222;
    asm volatile("" : "+v"(km), "+v"(kc));
    i32x8 a0 = {0x22222222, 0x2a2a2a2a, 0x22aa22aa, 0x2222aaaa, 0, 0, 0, 0}, a1 = {0x2a2a2a2a, 0x22222222, 0x2222aaaa, 0x22aa22aa, 0, 0, 0, 0};
    f32x16 c0, c1;
    for (int g = 0; g < 16; g++) { c0[g] = (float)g; c1[g] = (float)(g + 1); }
    uint32_t w[4] = {src[threadIdx.x & 63], src[64 + (threadIdx.x & 63)], src[128 + (threadIdx.x & 63)], src[192 + (threadIdx.x & 63)]};
    auto expand = [&](int (&b)[4][4], uint32_t salt) {
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const uint32_t x = w[s] ^ salt;
            b[s][0] = (int)((x & (uint32_t)km) | (uint32_t)kc);
            b[s][1] = (int)(((x << 1) & (uint32_t)km) | (uint32_t)kc);
            b[s][2] = (int)(((x << 2) & (uint32_t)km) | (uint32_t)kc);
            b[s][3] = (int)(((x << 3) & (uint32_t)km) | (uint32_t)kc);
        }
    };
    int b[4][4];
    expand(b, 0);
    const long long t0 = clock64();
    for (int i = 0; i < iters; i++) {
        int bn[4][4];
        if (AHEAD) expand(bn, (uint32_t)i + 1);
        else if (i) expand(b, (uint32_t)i);
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const i32x8 bv = {b[s][0], b[s][1], b[s][2], b[s][3], 0, 0, 0, 0};
            c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a0, bv, c0, 4, 4, 0, 127, 0, 127);
            c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a1, bv, c1, 4, 4, 0, 127, 0, 127);
        }
        if (AHEAD) {
#pragma unroll
            for (int s = 0; s < 4; s++)
#pragma unroll
                for (int q = 0; q < 4; q++) b[s][q] = bn[s][q];
        }
    }
    const long long t1 = clock64();
    float sum = 0;
    for (int g = 0; g < 16; g++) sum += c0[g] + c1[g];
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int NCH>
__global__ void i8_rate(int *out, long long *cyc, int iters)
{
    i32x4 a = {0x40404040, 0x40c040c0, 0x4040c0c0, 0x40404040}, b = a;
    i32x16 acc[NCH];
    for (int c = 0; c < NCH; c++) for (int g = 0; g < 16; g++) acc[c][g] = c + g;
    const long long t0 = clock64();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int c = 0; c < NCH; c++) acc[c] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[c], 0, 0, 0);
    }
    const long long t1 = clock64();
    int s = 0;
    for (int c = 0; c < NCH; c++) for (int g = 0; g < 16; g++) s += acc[c][g];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

int main()
{
    float *dO; long long *dC, hC; const int iters = 4000;
    hipMalloc(&dO, 256 * 256 * 8 * 4); hipMalloc(&dC, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char *name, auto launch, int nch, double ops_per_mfma) {
        launch(); hipDeviceSynchronize();
        hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(&hC, dC, 8, hipMemcpyDeviceToHost);
        const double n = (double)iters * nch;
        // grid: 256 CUs x 4 waves = 1024 waves, one per SIMD
        printf("%-22s chains %d: %.1f clock64 ticks per MFMA (wave 0), %.3f ms for %g MFMAs per wave -> %.2f POP/s chip-wide\n", name, nch, hC / n, ms, n,
               n * 1024 * ops_per_mfma / (ms * 1e-3) / 1e15);
    };
    run("fp4 32x32x64", [&] { fp4_rate<1><<<256, 256>>>(dO, dC, iters); }, 1, 2.0 * 32 * 32 * 64);
    run("fp4 32x32x64", [&] { fp4_rate<2><<<256, 256>>>(dO, dC, iters); }, 2, 2.0 * 32 * 32 * 64);
    run("fp4 32x32x64", [&] { fp4_rate<4><<<256, 256>>>(dO, dC, iters); }, 4, 2.0 * 32 * 32 * 64);
    run("i8 32x32x32", [&] { i8_rate<1><<<256, 256>>>((int *)dO, dC, iters); }, 1, 2.0 * 32 * 32 * 32);
    run("i8 32x32x32", [&] { i8_rate<2><<<256, 256>>>((int *)dO, dC, iters); }, 2, 2.0 * 32 * 32 * 32);
    run("i8 32x32x32", [&] { i8_rate<4><<<256, 256>>>((int *)dO, dC, iters); }, 4, 2.0 * 32 * 32 * 32);
    uint32_t *dS; hipMalloc(&dS, 1024); hipMemset(dS, 0x5a, 1024);
    run("fp4 step, in order", [&] { fp4_step_rate<false><<<256, 512>>>(dO, dC, iters, dS); }, 8, 2.0 * 2 * 32 * 32 * 64);
    run("fp4 step, 1 ahead", [&] { fp4_step_rate<true><<<256, 512>>>(dO, dC, iters, dS); }, 8, 2.0 * 2 * 32 * 32 * 64);
    // two waves per SIMD (512 threads per CU)
    run("fp4, 2 waves/SIMD", [&] { fp4_rate<2><<<256, 512>>>(dO, dC, iters); }, 2, 2.0 * 2 * 32 * 32 * 64);
    run("i8, 2 waves/SIMD", [&] { i8_rate<2><<<256, 512>>>((int *)dO, dC, iters); }, 2, 2.0 * 2 * 32 * 32 * 32);
    return 0;
}
