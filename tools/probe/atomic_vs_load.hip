// Probe: can a 64-bit load observe anything but the old or the new value of a word that another workgroup (other XCD)
// is changing with a 64-bit atomic AND?  Words start as ~0; writers AND word i down to the single bit (i % 64); readers
// poll every word with agent-scope (sc1) loads, plain loads and 0-OR atomics.  Any third value is a violation.
// build: hipcc -O3 --offload-arch=gfx950 atomic_vs_load.hip -o atomic_vs_load
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(256) void k_fill(unsigned long long *a, int n) { int i = blockIdx.x * 256 + threadIdx.x; if (i < n) a[i] = ~0ull; }

template <int MODE> // reader's access: 0 sc1 load, 1 plain volatile load, 2 atomic OR 0, 3 sc1 load by a few lanes of the wave only
__global__ __launch_bounds__(256) void k_race(unsigned long long *a, int n, unsigned int *viol, unsigned int *seen_new)
{
    const int g = blockIdx.x, G = gridDim.x;
    if (g & 1) { // writer: its slice, one AND per word
        const int W = G / 2, w = g / 2, per = (n + W - 1) / W;
        for (int i = w * per + threadIdx.x; i < (w + 1) * per && i < n; i += 256) atomicAnd(a + i, 1ull << (i & 63));
    } else { // reader: sweep everything a few times, starting at a different place
        unsigned int bad = 0, nw = 0;
        for (int pass = 0; pass < 4; pass++)
            for (int t = threadIdx.x; t < n; t += 256) {
                const int i = (t + g * 977) % n;
                unsigned long long v = ~0ull;
                if (MODE == 3) { if (((threadIdx.x + pass * 7 + t / 256) % 23) == 0) v = __hip_atomic_load(a + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
                else if (MODE == 0) v = __hip_atomic_load(a + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else if (MODE == 1) v = *(volatile unsigned long long *)(a + i);
                else v = atomicOr(a + i, 0ull);
                const unsigned long long k = 1ull << (i & 63);
                if (v != ~0ull && v != k) bad++;
                nw += v == k;
            }
        if (bad) atomicAdd(viol, bad);
        if (nw) atomicAdd(seen_new, nw);
    }
}

int main()
{
    const int n = 1 << 16, G = 2048;
    unsigned long long *a; unsigned int *c;
    hipMalloc(&a, (size_t)n * 8); hipMalloc(&c, 8);
    for (int mode = 0; mode < 4; mode++) {
        hipMemset(c, 0, 8);
        for (int rep = 0; rep < 50; rep++) {
            hipLaunchKernelGGL(k_fill, dim3((n + 255) / 256), dim3(256), 0, 0, a, n);
            if (mode == 0) hipLaunchKernelGGL(k_race<0>, dim3(G), dim3(256), 0, 0, a, n, c, c + 1);
            else if (mode == 1) hipLaunchKernelGGL(k_race<1>, dim3(G), dim3(256), 0, 0, a, n, c, c + 1);
            else if (mode == 3) hipLaunchKernelGGL(k_race<3>, dim3(G), dim3(256), 0, 0, a, n, c, c + 1);
            else hipLaunchKernelGGL(k_race<2>, dim3(G), dim3(256), 0, 0, a, n, c, c + 1);
        }
        unsigned int h[2];
        hipMemcpy(h, c, 8, hipMemcpyDeviceToHost);
        printf("reader mode %d: violations %u (reads that saw the new value: %u)\n", mode, h[0], h[1]);
    }
    return 0;
}
