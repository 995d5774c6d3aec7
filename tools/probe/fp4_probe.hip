// Probe: v_mfma_scale_f32_32x32x64_f8f6f4 with fp4 (e2m1) operands as a +-1 dot product.
// One wave: A = 32 rows x 256 bits, B = 32 cols x 256 bits; acc = 4096 * dot + C exactly?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
using i32x8 = int __attribute__((ext_vector_type(8)));
using f32x16 = float __attribute__((ext_vector_type(16)));

// 32 bits of a descriptor word -> 32 fp4 codes in 4 dwords: dword q takes bits {4i + 3 - q}; set -> -1 (0xA), clear -> +1 (0x2)
__device__ __forceinline__ void expand4(uint32_t w, int *o)
{
    o[0] = (int)((w & 0x88888888u) | 0x22222222u);
    o[1] = (int)(((w << 1) & 0x88888888u) | 0x22222222u);
    o[2] = (int)(((w << 2) & 0x88888888u) | 0x22222222u);
    o[3] = (int)(((w << 3) & 0x88888888u) | 0x22222222u);
}

__global__ void probe(const uint32_t *A, const uint32_t *B, float *out, int scale_a)
{
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    f32x16 acc;
    for (int g = 0; g < 16; g++) acc[g] = (float)((1 << 22) + (127 << 7) + 100 * g + lane % 7); // C input: something recognisable
    for (int s = 0; s < 4; s++) {
        i32x8 a = {0, 0, 0, 0, 0, 0, 0, 0}, b = {0, 0, 0, 0, 0, 0, 0, 0};
        int t[4];
        expand4(A[r * 8 + 2 * s + h], t); a[0] = t[0]; a[1] = t[1]; a[2] = t[2]; a[3] = t[3];
        expand4(B[r * 8 + 2 * s + h], t); b[0] = t[0]; b[1] = t[1]; b[2] = t[2]; b[3] = t[3];
        acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 4, 4, 0, scale_a, 0, 127);
    }
    for (int g = 0; g < 16; g++) out[lane * 16 + g] = acc[g];
}

int main()
{
    uint32_t hA[32 * 8], hB[32 * 8];
    srand(7);
    for (int i = 0; i < 256; i++) { hA[i] = (uint32_t)rand() ^ ((uint32_t)rand() << 16); hB[i] = (uint32_t)rand() ^ ((uint32_t)rand() << 16); }
    // rows/cols with every distance range: column c = row c with (8 c) bits flipped
    for (int c = 0; c < 32; c++) for (int w = 0; w < 8; w++) { hB[c * 8 + w] = hA[c * 8 + w]; }
    for (int c = 0; c < 32; c++) for (int k = 0; k < 8 * c; k++) hB[c * 8 + (k >> 5)] ^= 1u << (k & 31);
    uint32_t *dA, *dB; float *dO; float hO[64 * 16];
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dO, sizeof hO);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    for (int sa : {127, 139, 140}) {
        probe<<<1, 64>>>(dA, dB, dO, sa);
        hipMemcpy(hO, dO, sizeof hO, hipMemcpyDeviceToHost);
        const double scale = sa == 127 ? 1.0 : (sa == 139 ? 4096.0 : 8192.0);
        int bad = 0;
        for (int lane = 0; lane < 64; lane++)
            for (int g = 0; g < 16; g++) {
                const int col = lane & 31, row = (g & 3) + 8 * (g >> 2) + 4 * (lane >> 5);
                int ham = 0;
                for (int w = 0; w < 8; w++) ham += __builtin_popcount(hA[row * 8 + w] ^ hB[col * 8 + w]);
                const double expect = scale * (256 - 2 * ham) + ((1 << 22) + (127 << 7) + 100 * g + lane % 7);
                if ((double)hO[lane * 16 + g] != expect) { if (bad < 5) printf("scale_a=%d lane %d g %d: got %.1f expect %.1f (ham %d)\n", sa, lane, g, hO[lane * 16 + g], expect, ham); bad++; }
            }
        printf("scale_a=%d: %d mismatches of 1024\n", sa, bad);
    }
    return 0;
}
