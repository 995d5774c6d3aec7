// Probe: after a kernel boundary, can an agent-scope (sc1) load or a plain load still return a value the location had two
// kernels ago?  Pattern of the NMS workspace: K1 plain-stores fresh words from every workgroup, K2 reads words of other
// rows with sc1 loads (and a few atomic ANDs that change nothing), K3 zeroes everything with sc1 stores from one
// workgroup per slab.  A zero (or a word of an older repetition) seen by K2 is a stale read.
// build: hipcc -O3 --offload-arch=gfx950 stale_sc1.hip -o stale_sc1
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ unsigned long long word(int i, int rep) { return ((unsigned long long)(rep + 1) << 40) | (unsigned long long)(i + 1); }

__global__ __launch_bounds__(256) void k1_fill(unsigned long long *a, int n, int rep)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) a[i] = word(i, rep);
}

template <int ATOM>
__global__ __launch_bounds__(256) void k2_read(unsigned long long *a, int n, int rep, int pitch, unsigned int *stale)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    unsigned int bad = 0;
#pragma unroll
    for (int dy = -2; dy <= 2; dy++) {
        const int j = i + dy * pitch;
        if (j < 0 || j >= n) continue;
        const unsigned long long v = __hip_atomic_load(a + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bad += v != word(j, rep);
    }
    if (ATOM && (i & 7) == 0) atomicAnd(a + ((i + 3 * pitch) % n), ~0ull);
    if (bad) atomicAdd(stale, bad);
}

__global__ __launch_bounds__(1024) void k3_zero(unsigned long long *a, int n, int slab)
{
    const int base = blockIdx.x * slab;
    for (int i = threadIdx.x; i < slab && base + i < n; i += 1024) __hip_atomic_store(a + base + i, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

int main()
{
    const int pitch = 244, slab = 244 * 139, F = 8, n = slab * F; // 8 frames of 1080p cell grids
    unsigned long long *a; unsigned int *stale;
    hipMalloc(&a, (size_t)n * 8); hipMalloc(&stale, 4);
    hipMemset(a, 0, (size_t)n * 8);
    for (int atom = 0; atom < 2; atom++) {
        hipMemset(stale, 0, 4);
        for (int rep = 0; rep < 200; rep++) {
            hipLaunchKernelGGL(k1_fill, dim3((n + 255) / 256), dim3(256), 0, 0, a, n, rep);
            if (atom) hipLaunchKernelGGL(k2_read<1>, dim3((n + 255) / 256), dim3(256), 0, 0, a, n, rep, pitch, stale);
            else hipLaunchKernelGGL(k2_read<0>, dim3((n + 255) / 256), dim3(256), 0, 0, a, n, rep, pitch, stale);
            hipLaunchKernelGGL(k3_zero, dim3(F), dim3(1024), 0, 0, a, n, slab);
        }
        unsigned int h = 0;
        hipMemcpy(&h, stale, 4, hipMemcpyDeviceToHost);
        printf("atomics %d: stale words seen over 200 repetitions: %u\n", atom, h);
    }
    return 0;
}
