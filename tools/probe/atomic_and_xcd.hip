// Probe: are agent-scope 64-bit atomic ANDs from workgroups on different XCDs coherent with each other and with sc1 / plain
// loads of the same lines?  Workgroup g clears bit (g % 64) of every word; afterwards every word must be 0.
// build: hipcc -O3 --offload-arch=gfx950 atomic_and_xcd.hip -o atomic_and_xcd ; run: ./atomic_and_xcd
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void k_fill(unsigned long long *a, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) a[i] = ~0ull; }

template <int MODE> // 0: atomics only; 1: + sc1 loads of other words; 2: + plain loads of other words; 3: both
__global__ __launch_bounds__(256) void k_and(unsigned long long *a, int n, unsigned long long *sink)
{
    const unsigned long long clr = ~(1ull << (blockIdx.x & 63));
    unsigned long long acc = 0;
    for (int i = threadIdx.x; i < n; i += 256) {
        const int j = (i * 37 + blockIdx.x * 101) % n;
        if (MODE & 1) acc += __hip_atomic_load(a + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (MODE & 2) acc += a[(j + 17) % n];
        atomicAnd(a + i, clr);
    }
    if (acc == 0x1234567ull) *sink = acc;
}

int main()
{
    const int n = 1 << 14, G = 1024; // 128 KB of words, 1024 workgroups (16 per bit)
    unsigned long long *a, *sink;
    hipMalloc(&a, n * 8); hipMalloc(&sink, 8);
    std::vector<unsigned long long> h(n);
    for (int mode = 0; mode < 4; mode++) {
        int bad_total = 0;
        for (int rep = 0; rep < 20; rep++) {
            hipLaunchKernelGGL(k_fill, dim3((n + 255) / 256), dim3(256), 0, 0, a, n);
            switch (mode) {
            case 0: hipLaunchKernelGGL(k_and<0>, dim3(G), dim3(256), 0, 0, a, n, sink); break;
            case 1: hipLaunchKernelGGL(k_and<1>, dim3(G), dim3(256), 0, 0, a, n, sink); break;
            case 2: hipLaunchKernelGGL(k_and<2>, dim3(G), dim3(256), 0, 0, a, n, sink); break;
            default: hipLaunchKernelGGL(k_and<3>, dim3(G), dim3(256), 0, 0, a, n, sink); break;
            }
            hipMemcpy(h.data(), a, n * 8, hipMemcpyDeviceToHost);
            int bad = 0;
            for (int i = 0; i < n; i++) bad += h[i] != 0ull;
            bad_total += bad;
        }
        printf("mode %d: words with lost updates over 20 runs: %d\n", mode, bad_total);
    }
    return 0;
}
