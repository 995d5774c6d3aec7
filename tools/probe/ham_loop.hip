// Probe: what holds the MFMA issue rate of k_ham_fp4's column-tile loop below the back-to-back rate?
// The loop is rebuilt here from toggled parts -- column fetch (global, through an LDS index), fp4 expansion, NCH chains of 4
// MFMAs with a FRESH C input per tile, the row-side maxima, the column-side maximum + LDS atomic -- and timed per column tile
// at 1 and 2 waves per SIMD.  Build: hipcc -O3 --offload-arch=gfx950 ham_loop.hip -o ham_loop ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
using i32x8 = int __attribute__((ext_vector_type(8)));
using f32x16 = float __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void expand_fp4(uint32_t w, int km, int kc, int &d0, int &d1, int &d2, int &d3)
{
    d0 = (int)((w & (uint32_t)km) | (uint32_t)kc);
    d1 = (int)(((w << 1) & (uint32_t)km) | (uint32_t)kc);
    d2 = (int)(((w << 2) & (uint32_t)km) | (uint32_t)kc);
    d3 = (int)(((w << 3) & (uint32_t)km) | (uint32_t)kc);
}

// FETCH: column words from global memory through an LDS index (else a register rotation); EXPAND: the 28 vector ops;
// ROW: rbest = max(rbest, acc) per element; COL: max tree over the 16 accumulators + one LDS atomic per tile;
// FRESHC: every chain starts from the C vector (else it accumulates in place, like tools/probe/mfma_rate.hip)
template <int NCH, bool FETCH, bool EXPAND, bool ROW, bool COL, bool FRESHC>
__global__ __launch_bounds__(256, 2) void loop(const uint32_t *__restrict__ desc, int ntile, int *out, long long *cyc)
{
    __shared__ uint32_t colidx[4096 + 256];
    __shared__ int colbest[4096];
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
    for (int c = tid; c < 4096 + 256; c += blockDim.x) colidx[c] = (uint32_t)((c * 7 + blockIdx.x) & 4095);
    for (int c = tid; c < 4096; c += blockDim.x) colbest[c] = -(1 << 30);
    int km = (int)0x88888888, kc = 0x22222222;
    asm volatile("" : "+v"(km), "+v"(kc));
    int afr[NCH][4][4];
    for (int t = 0; t < NCH; t++) {
        const uint4 w = *reinterpret_cast<const uint4 *>(desc + (size_t)((tid + 37 * t) & 4095) * 8 + 4 * h);
        expand_fp4(w.x, km, kc, afr[t][0][0], afr[t][0][1], afr[t][0][2], afr[t][0][3]);
        expand_fp4(w.y, km, kc, afr[t][1][0], afr[t][1][1], afr[t][1][2], afr[t][1][3]);
        expand_fp4(w.z, km, kc, afr[t][2][0], afr[t][2][1], afr[t][2][2], afr[t][2][3]);
        expand_fp4(w.w, km, kc, afr[t][3][0], afr[t][3][1], afr[t][3][2], afr[t][3][3]);
    }
    f32x16 cc, acc[NCH];
    int rbest[NCH][16];
    for (int g = 0; g < 16; g++) {
        cc[g] = (float)((1 << 22) + ((127 << 7) | (127 - g)));
        for (int t = 0; t < NCH; t++) { rbest[t][g] = 0; acc[t][g] = cc[g]; }
    }
    __syncthreads();
    auto fetch = [&](int ct) -> uint4 { return *reinterpret_cast<const uint4 *>(desc + (size_t)(colidx[(ct & 127) * 32 + r] & 4095u) * 8 + 4 * h); };   // the tile index wraps: 128 tiles of 32 columns
    uint4 w0 = fetch(0), w1 = fetch(1), w2 = fetch(2), w3 = fetch(3);
    int b[4][4];
    expand_fp4(w0.x, km, kc, b[0][0], b[0][1], b[0][2], b[0][3]);
    expand_fp4(w0.y, km, kc, b[1][0], b[1][1], b[1][2], b[1][3]);
    expand_fp4(w0.z, km, kc, b[2][0], b[2][1], b[2][2], b[2][3]);
    expand_fp4(w0.w, km, kc, b[3][0], b[3][1], b[3][2], b[3][3]);
    auto step = [&](int ct, uint4 &slot) {
        if (EXPAND) {
            expand_fp4(slot.x, km, kc, b[0][0], b[0][1], b[0][2], b[0][3]);
            expand_fp4(slot.y, km, kc, b[1][0], b[1][1], b[1][2], b[1][3]);
            expand_fp4(slot.z, km, kc, b[2][0], b[2][1], b[2][2], b[2][3]);
            expand_fp4(slot.w, km, kc, b[3][0], b[3][1], b[3][2], b[3][3]);
        }
        if (FETCH) slot = fetch(ct + 4);
        else { slot.x += 0x01010101u; slot.y ^= slot.x; }
        int cm = 0;
#pragma unroll
        for (int t = 0; t < NCH; t++) {
            if (FRESHC) acc[t] = cc;
#pragma unroll
            for (int s4 = 0; s4 < 4; s4++) {
                const i32x8 av = {afr[t][s4][0], afr[t][s4][1], afr[t][s4][2], afr[t][s4][3], 0, 0, 0, 0};
                const i32x8 bv = {b[s4][0], b[s4][1], b[s4][2], b[s4][3], 0, 0, 0, 0};
                acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, acc[t], 4, 4, 0, 140, 0, 127);
            }
            if (ROW) {
#pragma unroll
                for (int g = 0; g < 16; g++) rbest[t][g] = max(rbest[t][g], __float_as_int(acc[t][g]));
            }
            if (COL) {
                int m = __float_as_int(acc[t][0]);
#pragma unroll
                for (int g = 1; g < 16; g++) m = max(m, __float_as_int(acc[t][g]));
                cm = max(cm, m - 32 * t);
            }
        }
        if (COL) atomicMax(&colbest[(ct & 127) * 32 + r], cm);
        if (FRESHC) {
#pragma unroll
            for (int g = 0; g < 16; g++) cc[g] -= 128.0f;
        }
    };
    const long long t0 = wall_clock64();
    int ct = 0;
    for (; ct + 4 <= ntile; ct += 4) {
        step(ct, w0);
        step(ct + 1, w1);
        step(ct + 2, w2);
        step(ct + 3, w3);
    }
    const long long t1 = wall_clock64();
    int s = 0;
    for (int t = 0; t < NCH; t++) for (int g = 0; g < 16; g++) s += rbest[t][g] + __float_as_int(acc[t][g]);
    out[blockIdx.x * blockDim.x + tid] = s + colbest[tid] + (int)w0.x + (int)w1.y + (int)w2.x + (int)w3.y + b[0][0];
    if (tid == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

// in-place MFMA chains (results consumed after the loop only) beside NV vector instructions per tile that do NOT touch the
// accumulators (v_max on private registers): does a wave co-issue that much vector work with its own MFMAs at all?
template <int NCH, int NV>
__global__ __launch_bounds__(256, 2) void loop_indep(const uint32_t *__restrict__ desc, int ntile, int *out, long long *cyc)
{
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
    int km = (int)0x88888888, kc = 0x22222222;
    asm volatile("" : "+v"(km), "+v"(kc));
    int afr[NCH][4][4], b[4][4];
    for (int t = 0; t < NCH; t++) {
        const uint4 w = *reinterpret_cast<const uint4 *>(desc + (size_t)((tid + 37 * t) & 4095) * 8 + 4 * h);
        expand_fp4(w.x, km, kc, afr[t][0][0], afr[t][0][1], afr[t][0][2], afr[t][0][3]);
        expand_fp4(w.y, km, kc, afr[t][1][0], afr[t][1][1], afr[t][1][2], afr[t][1][3]);
        expand_fp4(w.z, km, kc, afr[t][2][0], afr[t][2][1], afr[t][2][2], afr[t][2][3]);
        expand_fp4(w.w, km, kc, afr[t][3][0], afr[t][3][1], afr[t][3][2], afr[t][3][3]);
    }
    for (int s4 = 0; s4 < 4; s4++) for (int q = 0; q < 4; q++) b[s4][q] = afr[0][s4][q] ^ 0x08080808;
    f32x16 acc[NCH];
    int x[16], y[16];
    for (int g = 0; g < 16; g++) { x[g] = tid * 31 + g; y[g] = tid * 17 - g; for (int t = 0; t < NCH; t++) acc[t][g] = (float)g; }
    const long long t0 = wall_clock64();
    for (int ct = 0; ct < ntile; ct++) {
#pragma unroll
        for (int s4 = 0; s4 < 4; s4++)
#pragma unroll
            for (int t = 0; t < NCH; t++) {
                const i32x8 av = {afr[t][s4][0], afr[t][s4][1], afr[t][s4][2], afr[t][s4][3], 0, 0, 0, 0};
                const i32x8 bv = {b[s4][0], b[s4][1], b[s4][2], b[s4][3], 0, 0, 0, 0};
                acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, acc[t], 4, 4, 0, 140, 0, 127);
            }
#pragma unroll
        for (int k = 0; k < NV; k++) { x[k & 15] = max(x[k & 15], y[(k * 5 + 3) & 15] + ct); y[k & 15] ^= x[(k + 7) & 15]; }
    }
    const long long t1 = wall_clock64();
    int s = 0;
    for (int t = 0; t < NCH; t++) for (int g = 0; g < 16; g++) s += __float_as_int(acc[t][g]) + x[g] + y[g];
    out[blockIdx.x * blockDim.x + tid] = s;
    if (tid == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

using i32x4 = int __attribute__((ext_vector_type(4)));
using i32x16 = int __attribute__((ext_vector_type(16)));
__device__ __forceinline__ int max16i(const i32x16 &v)
{
    const int m0 = max(max(v[0], v[1]), v[2]), m1 = max(max(v[3], v[4]), v[5]), m2 = max(max(v[6], v[7]), v[8]);
    const int m3 = max(max(v[9], v[10]), v[11]), m4 = max(max(v[12], v[13]), v[14]);
    return max(max(max(m0, m1), max(m2, m3)), max(m4, v[15]));
}
// the shipped kernel's loop, source form for source form; SHARE: column fragments expanded once per workgroup through LDS
template <int RT, bool SHARE>
__global__ __launch_bounds__(256, 2) void loop_kernel(const uint32_t *__restrict__ desc, int ntile, int *out, long long *cyc)
{
    __shared__ uint32_t colidx[4096 + 256];
    __shared__ int colbest[4096];
    __shared__ i32x4 bring[2][4][4][64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, r = lane & 31, h = lane >> 5;
    for (int c = tid; c < 4096 + 256; c += blockDim.x) colidx[c] = (uint32_t)((c * 7 + blockIdx.x) & 4095);
    for (int c = tid; c < 4096; c += blockDim.x) colbest[c] = -(1 << 30);
    int km = (int)0x88888888, kc = 0x22222222;
    asm volatile("" : "+v"(km), "+v"(kc));
    int afr[RT][4][4];
    for (int t = 0; t < RT; t++) {
        const uint4 w = *reinterpret_cast<const uint4 *>(desc + (size_t)((tid + 37 * t) & 4095) * 8 + 4 * h);
        expand_fp4(w.x, km, kc, afr[t][0][0], afr[t][0][1], afr[t][0][2], afr[t][0][3]);
        expand_fp4(w.y, km, kc, afr[t][1][0], afr[t][1][1], afr[t][1][2], afr[t][1][3]);
        expand_fp4(w.z, km, kc, afr[t][2][0], afr[t][2][1], afr[t][2][2], afr[t][2][3]);
        expand_fp4(w.w, km, kc, afr[t][3][0], afr[t][3][1], afr[t][3][2], afr[t][3][3]);
    }
    f32x16 cc;
    int rbest[RT][16];
    for (int g = 0; g < 16; g++) {
        cc[g] = (float)((1 << 22) + ((127 << 7) | (127 - g)));
        for (int t = 0; t < RT; t++) rbest[t][g] = 0;
    }
    __syncthreads();
    auto fetch = [&](int ct) -> uint4 { return *reinterpret_cast<const uint4 *>(desc + (size_t)(colidx[(ct & 127) * 32 + r] & 4095u) * 8 + 4 * h); };
    const int kwv = (127 - wv) << 7;
    auto body = [&](int ct, const int (&b)[4][4]) {
        int cm = 0;
#pragma unroll
        for (int t = 0; t < RT; t++) {
            f32x16 acc = cc;
#pragma unroll
            for (int s4 = 0; s4 < 4; s4++) {
                const i32x8 av = {afr[t][s4][0], afr[t][s4][1], afr[t][s4][2], afr[t][s4][3], 0, 0, 0, 0};
                const i32x8 bv = {b[s4][0], b[s4][1], b[s4][2], b[s4][3], 0, 0, 0, 0};
                acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, acc, 4, 4, 0, 140, 0, 127);
            }
            i32x16 ai;
#pragma unroll
            for (int g = 0; g < 16; g++) ai[g] = __float_as_int(acc[g]);
#pragma unroll
            for (int g = 0; g < 16; g++) rbest[t][g] = max(rbest[t][g], ai[g]);
            cm = max(cm, __float_as_int(__int_as_float(max16i(ai)) - (float)(32 * t)));
        }
        const int cmi = (int)__int_as_float(cm) - (1 << 22);
        atomicMax(&colbest[(ct & 127) * 32 + r], (cmi & (int)0xFFFFC07F) | kwv);
#pragma unroll
        for (int g = 0; g < 16; g++) cc[g] -= 128.0f;
    };
    const long long t0 = wall_clock64();
    if (!SHARE) {
        uint4 w0 = fetch(0), w1 = fetch(1), w2 = fetch(2), w3 = fetch(3);
        auto step = [&](int ct, uint4 &slot) {
            int b[4][4];
            expand_fp4(slot.x, km, kc, b[0][0], b[0][1], b[0][2], b[0][3]);
            expand_fp4(slot.y, km, kc, b[1][0], b[1][1], b[1][2], b[1][3]);
            expand_fp4(slot.z, km, kc, b[2][0], b[2][1], b[2][2], b[2][3]);
            expand_fp4(slot.w, km, kc, b[3][0], b[3][1], b[3][2], b[3][3]);
            slot = fetch(ct + 4);
            body(ct, b);
        };
        for (int ct = 0; ct + 4 <= ntile; ct += 4) { step(ct, w0); step(ct + 1, w1); step(ct + 2, w2); step(ct + 3, w3); }
    } else {
        auto stage = [&](int half, const uint4 &w) {
            int f[4][4];
            expand_fp4(w.x, km, kc, f[0][0], f[0][1], f[0][2], f[0][3]);
            expand_fp4(w.y, km, kc, f[1][0], f[1][1], f[1][2], f[1][3]);
            expand_fp4(w.z, km, kc, f[2][0], f[2][1], f[2][2], f[2][3]);
            expand_fp4(w.w, km, kc, f[3][0], f[3][1], f[3][2], f[3][3]);
#pragma unroll
            for (int s4 = 0; s4 < 4; s4++) { const i32x4 v = {f[s4][0], f[s4][1], f[s4][2], f[s4][3]}; bring[half][wv][s4][lane] = v; }
        };
        uint4 wn = fetch(wv);
        stage(0, wn);
        wn = fetch(4 + wv);
        for (int g = 0; g * 4 < ntile; g++) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            stage((g + 1) & 1, wn);
            wn = fetch((g + 2) * 4 + wv);
#pragma unroll
            for (int q = 0; q < 4; q++) {
                int b[4][4];
#pragma unroll
                for (int s4 = 0; s4 < 4; s4++) { const i32x4 v = bring[g & 1][q][s4][lane]; b[s4][0] = v[0]; b[s4][1] = v[1]; b[s4][2] = v[2]; b[s4][3] = v[3]; }
                body(g * 4 + q, b);
            }
        }
    }
    const long long t1 = wall_clock64();
    int s = 0;
    for (int t = 0; t < RT; t++) for (int g = 0; g < 16; g++) s += rbest[t][g];
    out[blockIdx.x * blockDim.x + tid] = s + colbest[tid] + (int)__float_as_int(cc[3]);
    if (tid == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int NCH, bool SCHED>
__global__ __launch_bounds__(256, 2) void loop_pipe(const uint32_t *__restrict__ desc, int ntile, int *out, long long *cyc)
{
    __shared__ uint32_t colidx[4096 + 256];
    __shared__ int colbest[4096];
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
    for (int c = tid; c < 4096 + 256; c += blockDim.x) colidx[c] = (uint32_t)((c * 7 + blockIdx.x) & 4095);
    for (int c = tid; c < 4096; c += blockDim.x) colbest[c] = -(1 << 30);
    int km = (int)0x88888888, kc = 0x22222222;
    asm volatile("" : "+v"(km), "+v"(kc));
    int afr[NCH][4][4];
    for (int t = 0; t < NCH; t++) {
        const uint4 w = *reinterpret_cast<const uint4 *>(desc + (size_t)((tid + 37 * t) & 4095) * 8 + 4 * h);
        expand_fp4(w.x, km, kc, afr[t][0][0], afr[t][0][1], afr[t][0][2], afr[t][0][3]);
        expand_fp4(w.y, km, kc, afr[t][1][0], afr[t][1][1], afr[t][1][2], afr[t][1][3]);
        expand_fp4(w.z, km, kc, afr[t][2][0], afr[t][2][1], afr[t][2][2], afr[t][2][3]);
        expand_fp4(w.w, km, kc, afr[t][3][0], afr[t][3][1], afr[t][3][2], afr[t][3][3]);
    }
    f32x16 cc, accA[NCH], accB[NCH];
    int rbest[NCH][16];
    for (int g = 0; g < 16; g++) {
        cc[g] = (float)((1 << 22) + ((127 << 7) | (127 - g)));
        for (int t = 0; t < NCH; t++) rbest[t][g] = 0;
    }
    __syncthreads();
    auto fetch = [&](int ct) -> uint4 { return *reinterpret_cast<const uint4 *>(desc + (size_t)(colidx[(ct & 127) * 32 + r] & 4095u) * 8 + 4 * h); };
    uint4 w0 = fetch(0), w1 = fetch(1), w2 = fetch(2), w3 = fetch(3);
    // one pipeline stage: MFMAs of the tile in `slot` into accN, maxima of accC (tile ct)
    auto stage = [&](f32x16 (&accN)[NCH], const f32x16 (&accC)[NCH], uint4 &slot, int ct) {
        int b[4][4];
        expand_fp4(slot.x, km, kc, b[0][0], b[0][1], b[0][2], b[0][3]);
        expand_fp4(slot.y, km, kc, b[1][0], b[1][1], b[1][2], b[1][3]);
        expand_fp4(slot.z, km, kc, b[2][0], b[2][1], b[2][2], b[2][3]);
        expand_fp4(slot.w, km, kc, b[3][0], b[3][1], b[3][2], b[3][3]);
        slot = fetch(ct + 5);
#pragma unroll
        for (int t = 0; t < NCH; t++) accN[t] = cc;
#pragma unroll
        for (int s4 = 0; s4 < 4; s4++)
#pragma unroll
            for (int t = 0; t < NCH; t++) {
                const i32x8 av = {afr[t][s4][0], afr[t][s4][1], afr[t][s4][2], afr[t][s4][3], 0, 0, 0, 0};
                const i32x8 bv = {b[s4][0], b[s4][1], b[s4][2], b[s4][3], 0, 0, 0, 0};
                accN[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, accN[t], 4, 4, 0, 140, 0, 127);
            }
        int cm = 0;
#pragma unroll
        for (int t = 0; t < NCH; t++) {
#pragma unroll
            for (int g = 0; g < 16; g++) rbest[t][g] = max(rbest[t][g], __float_as_int(accC[t][g]));
            int m = __float_as_int(accC[t][0]);
#pragma unroll
            for (int g = 1; g < 16; g++) m = max(m, __float_as_int(accC[t][g]));
            cm = max(cm, m - 32 * t);
        }
        atomicMax(&colbest[(ct & 127) * 32 + r], cm);
#pragma unroll
        for (int g = 0; g < 16; g++) cc[g] -= 128.0f;
        if (SCHED) { // 4 * NCH MFMAs, each followed by a share of the vector work
#pragma unroll
            for (int k = 0; k < 4 * NCH; k++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                       // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, NCH == 3 ? 9 : 8, 0);        // vector ALU
            }
        }
    };
    for (int t = 0; t < NCH; t++) accA[t] = cc;
    const long long t0 = wall_clock64();
    for (int ct = 0; ct + 4 <= ntile; ct += 4) {
        stage(accB, accA, w1, ct);
        stage(accA, accB, w2, ct + 1);
        stage(accB, accA, w3, ct + 2);
        stage(accA, accB, w0, ct + 3);
    }
    const long long t1 = wall_clock64();
    int s = 0;
    for (int t = 0; t < NCH; t++) for (int g = 0; g < 16; g++) s += rbest[t][g] + __float_as_int(accA[t][g]);
    out[blockIdx.x * blockDim.x + tid] = s + colbest[tid] + (int)w0.x + (int)w1.y + (int)w2.x + (int)w3.y;
    if (tid == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

int main()
{
    uint32_t *dD; int *dO; long long *dC, hC;
    hipMalloc(&dD, 4096 * 32 + 64); hipMemset(dD, 0x5a, 4096 * 32 + 64);
    hipMalloc(&dO, 512 * 256 * 4); hipMalloc(&dC, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int ntile = 4096;   // tiles per wave (the index wraps): long enough to drown prologue and epilogue
    auto run = [&](const char *name, auto kern, int nch, int blocks) {
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, dD, ntile, dO, dC); hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, dD, ntile, dO, dC);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(&hC, dC, 8, hipMemcpyDeviceToHost);
        const double mfmas = (double)ntile * nch * 4 * blocks * 4;      // per launch
        printf("%-58s %d chains, %d waves/SIMD: %7.3f ms  %6.1f ns per tile and wave (wave 0: %5.0f)  %5.2f POP/s = %.2f of 10\n", name, nch,
               blocks / 256, ms, ms * 1e6 / ntile, hC * 10.0 / ntile, mfmas * 2.0 * 32 * 32 * 64 / (ms * 1e-3) / 1e15,
               mfmas * 2.0 * 32 * 32 * 64 / (ms * 1e-3) / 1e16);
    };
    for (int blocks : {256, 512}) {
        run("MFMA only, in place", loop<3, false, false, false, false, false>, 3, blocks);
        run("MFMA only, fresh C per tile", loop<3, false, false, false, false, true>, 3, blocks);
        run("+ expansion", loop<3, false, true, false, false, true>, 3, blocks);
        run("+ expansion + fetch", loop<3, true, true, false, false, true>, 3, blocks);
        run("+ expansion + fetch + row maxima", loop<3, true, true, true, false, true>, 3, blocks);
        run("+ expansion + fetch + column maximum/atomic", loop<3, true, true, false, true, true>, 3, blocks);
        run("everything (the kernel's loop)", loop<3, true, true, true, true, true>, 3, blocks);
        run("everything, 2 chains", loop<2, true, true, true, true, true>, 2, blocks);
        run("everything, pipelined over tiles, 2 chains", loop_pipe<2, false>, 2, blocks);
        run("everything, pipelined, 2 chains, sched_group_barrier", loop_pipe<2, true>, 2, blocks);
        run("the shipped loop, source for source (3 row tiles)", loop_kernel<3, false>, 3, blocks);
        run("the shipped loop + LDS-shared expansion", loop_kernel<3, true>, 3, blocks);
        run("in-place MFMAs + 24 x 3 independent vector ops", loop_indep<3, 24>, 3, blocks);
        run("in-place MFMAs + 28 x 3 independent vector ops", loop_indep<3, 28>, 3, blocks);
        run("in-place MFMAs + 32 x 3 independent vector ops", loop_indep<3, 32>, 3, blocks);
        run("in-place MFMAs + 36 x 3 independent vector ops", loop_indep<3, 36>, 3, blocks);
        run("in-place MFMAs + 40 x 3 independent vector ops", loop_indep<3, 40>, 3, blocks);
        run("in-place MFMAs + 48 x 3 independent vector ops", loop_indep<3, 48>, 3, blocks);
        run("in-place MFMAs, 2 chains + 32 x 3 independent", loop_indep<2, 32>, 2, blocks);
    }
    return 0;
}
