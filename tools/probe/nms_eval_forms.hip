// Probe: the two ways of writing the NMS neighbour evaluation (k_nms.hip) on random per-lane inputs with no concurrency.
// FORM A: per-lane predicated loads + nested ternaries on the score code (the form that produced wrong survivors on
// gfx950); FORM B: unconditional loads + branch-free level masks (shipped).  Any disagreement is a code-generation
// problem, not a race.
// build: hipcc -O3 --offload-arch=gfx950 nms_eval_forms.hip -o nms_eval_forms
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <random>

constexpr int RR = 2, ND = 5, NO = 25;

struct In { const unsigned long long *an, *dm, *p0, *p1, *p2; const uint32_t *me; const unsigned long long *needm; };

__device__ __forceinline__ unsigned long long ld(const unsigned long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <int FORM>
__global__ __launch_bounds__(256) void k_eval(In in, int n, uint32_t *out)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    const bool live = t < n;
    const int tt = live ? t : 0;
    const uint32_t me = in.me[tt];
    const int code = (int)(me >> 28), uy = (int)((me >> 3) & 7);
    const unsigned long long rows_below = (1ull << (8 * uy)) - 1ull, row_mine = 0xFFull << (8 * uy);
    const unsigned long long s5 = code == 5 ? ~0ull : 0ull, s4 = code == 4 ? ~0ull : 0ull, s3 = code == 3 ? ~0ull : 0ull,
                             s2 = code == 2 ? ~0ull : 0ull, s1 = code == 1 ? ~0ull : 0ull;
    const unsigned long long needm = live ? in.needm[tt] : 0ull;
    const size_t base = (size_t)tt * NO;
    bool blocked = false;
#pragma unroll 1
    for (int j = 0; j < ND; j++) {
        const int dy = j - RR, g0 = j * ND;
        const unsigned long long gm = blocked ? 0ull : (needm >> g0) & ((1ull << ND) - 1ull);
        if (!__any(gm != 0ull)) continue;
        unsigned long long an[ND], dm[ND], p0[ND], p1[ND], p2[ND];
        if (FORM == 0) {
#pragma unroll
            for (int k = 0; k < ND; k++) {
                an[k] = 0ull; dm[k] = 0ull; p0[k] = 0ull; p1[k] = 0ull; p2[k] = 0ull;
                if ((gm >> k) & 1ull) {
                    an[k] = ld(in.an + base + g0 + k);
                    dm[k] = in.dm[base + g0 + k];
                    p0[k] = in.p0[base + g0 + k]; p1[k] = in.p1[base + g0 + k]; p2[k] = in.p2[base + g0 + k];
                }
            }
            const unsigned long long Eall = dy < 0 ? ~0ull : 0ull, Erow = dy == 0 ? rows_below : 0ull;
#pragma unroll
            for (int k = 0; k < ND; k++) {
                const int dx = k - RR;
                const unsigned long long E = Eall | Erow | ((dy == 0 && dx < 0) ? row_mine : 0ull);
                const unsigned long long hi = code == 5 ? 0ull : (code == 4 ? (p2[k] & p0[k]) : (code == 3 ? p2[k] : (code == 2 ? (p2[k] | (p1[k] & p0[k])) : (p2[k] | p1[k]))));
                const unsigned long long same = code == 5 ? (p2[k] & p0[k]) : (code == 4 ? (p2[k] & ~p0[k]) : (code == 3 ? (~p2[k] & p1[k] & p0[k]) : (code == 2 ? (~p2[k] & p1[k] & ~p0[k]) : (~p2[k] & ~p1[k]))));
                blocked = blocked || ((an[k] & dm[k] & (hi | (same & E))) != 0ull);
            }
        } else {
#pragma unroll
            for (int k = 0; k < ND; k++) {
                an[k] = ld(in.an + base + g0 + k);
                dm[k] = in.dm[base + g0 + k];
                p0[k] = in.p0[base + g0 + k]; p1[k] = in.p1[base + g0 + k]; p2[k] = in.p2[base + g0 + k];
            }
            unsigned long long hit = 0ull;
#pragma unroll
            for (int k = 0; k < ND; k++) {
                const int dx = k - RR;
                const unsigned long long E = dy < 0 ? ~0ull : (dy > 0 ? 0ull : (rows_below | (dx < 0 ? row_mine : 0ull)));
                const unsigned long long ge5 = p2[k] & p0[k], ge4 = p2[k], ge3 = p2[k] | (p1[k] & p0[k]), ge2 = p2[k] | p1[k];
                const unsigned long long above = (ge5 & s4) | (ge4 & s3) | (ge3 & s2) | (ge2 & s1);
                const unsigned long long mine = (ge5 & s5) | (ge4 & s4) | (ge3 & s3) | (ge2 & s2) | s1;
                hit |= an[k] & dm[k] & (above | (mine & E)) & (((gm >> k) & 1ull) ? ~0ull : 0ull);
            }
            blocked = blocked || hit != 0ull;
        }
    }
    if (live) out[t] = blocked ? 1u : 0u;
}

int main()
{
    const int n = 1 << 18;
    std::mt19937_64 rng(12345);
    std::vector<unsigned long long> an((size_t)n * NO), dm((size_t)n * NO), p0((size_t)n * NO), p1((size_t)n * NO), p2((size_t)n * NO), need(n);
    std::vector<uint32_t> me(n);
    auto sparse = [&](int keep_bits) { unsigned long long v = 0; for (int i = 0; i < keep_bits; i++) v |= 1ull << (rng() & 63); return v; };
    unsigned long long *d_an, *d_dm, *d_p0, *d_p1, *d_p2, *d_need; uint32_t *d_me, *d_o0, *d_o1;
    hipMalloc(&d_an, an.size() * 8); hipMalloc(&d_dm, an.size() * 8); hipMalloc(&d_p0, an.size() * 8); hipMalloc(&d_p1, an.size() * 8);
    hipMalloc(&d_p2, an.size() * 8); hipMalloc(&d_need, (size_t)n * 8); hipMalloc(&d_me, (size_t)n * 4); hipMalloc(&d_o0, (size_t)n * 4); hipMalloc(&d_o1, (size_t)n * 4);
    long long total_bad = 0, total_blocked = 0;
    for (int rep = 0; rep < 20; rep++) {
        for (int t = 0; t < n; t++) {
            const int r = (int)(rng() % 100);
            int code = 1 + (int)(rng() % 5);
            if (r < 10) code = (int)(rng() % 16); // passengers with any code
            me[t] = ((uint32_t)code << 28) | (uint32_t)(rng() & 0x3F);
            unsigned long long nm = 0;
            const int nneed = r < 10 ? 0 : (r < 60 ? 1 : (int)(rng() % 5));
            for (int i = 0; i < nneed; i++) { int o = (int)(rng() % NO); if (o != 12) nm |= 1ull << o; }
            need[t] = nm;
            for (int o = 0; o < NO; o++) {
                const size_t i = (size_t)t * NO + o;
                an[i] = sparse(1 + (int)(rng() % 3)); dm[i] = (rng() % 4) ? ~0ull : sparse(20);
                p0[i] = rng(); p1[i] = rng(); p2[i] = rng() & rng();
            }
        }
        hipMemcpy(d_an, an.data(), an.size() * 8, hipMemcpyHostToDevice); hipMemcpy(d_dm, dm.data(), an.size() * 8, hipMemcpyHostToDevice);
        hipMemcpy(d_p0, p0.data(), an.size() * 8, hipMemcpyHostToDevice); hipMemcpy(d_p1, p1.data(), an.size() * 8, hipMemcpyHostToDevice);
        hipMemcpy(d_p2, p2.data(), an.size() * 8, hipMemcpyHostToDevice); hipMemcpy(d_need, need.data(), (size_t)n * 8, hipMemcpyHostToDevice);
        hipMemcpy(d_me, me.data(), (size_t)n * 4, hipMemcpyHostToDevice);
        In in{d_an, d_dm, d_p0, d_p1, d_p2, d_me, d_need};
        hipLaunchKernelGGL(k_eval<0>, dim3((n + 255) / 256), dim3(256), 0, 0, in, n, d_o0);
        hipLaunchKernelGGL(k_eval<1>, dim3((n + 255) / 256), dim3(256), 0, 0, in, n, d_o1);
        std::vector<uint32_t> o0(n), o1(n);
        hipMemcpy(o0.data(), d_o0, (size_t)n * 4, hipMemcpyDeviceToHost); hipMemcpy(o1.data(), d_o1, (size_t)n * 4, hipMemcpyDeviceToHost);
        for (int t = 0; t < n; t++) { total_bad += o0[t] != o1[t]; total_blocked += o1[t]; }
    }
    printf("lanes evaluated %lld, blocked %lld, forms disagree on %lld\n", 20LL * n, total_blocked, total_bad);
    return 0;
}
