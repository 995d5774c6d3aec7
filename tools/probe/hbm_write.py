"""Probe: plain HBM write and copy rates on this GPU (what a store-bound kernel can hope for)."""
import time, torch
dev = "cuda:0"
n = 3 * 1024**3
a = torch.empty(n, dtype=torch.uint8, device=dev)
b = torch.empty(n, dtype=torch.uint8, device=dev)
for name, fn, byts in (("fill (write only)", lambda: a.fill_(7), n), ("copy (read + write)", lambda: b.copy_(a), 2 * n)):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print("%s: %.2f ms for %.1f GB = %.2f TB/s" % (name, dt * 1e3, byts / 1e9, byts / dt / 1e12))
