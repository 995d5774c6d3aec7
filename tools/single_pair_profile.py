import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
import photogrammetry_amd as pg
import bench
from photogrammetry_amd import synth
eng = pg.Engine(0)
eng.set_brief_pairs(pg.make_brief_pairs(0, 50, 256)); eng.set_detect_params(0.1, 16); eng.set_capacity(1 << 18, 4096)
eng.set_dewarp_map(pg.build_dewarp_map(1920, 1080, [3e-4, 1e-7, 0, 0, 0]))
f0 = bench.base_frame(1920, 1080, 4321); f1 = synth.shift_frame(f0, 37, 11)
d0 = eng.detect(f0, capacity=4096)[1]; d1 = eng.detect(f1, capacity=4096)[1]
eng.match(d0, d1)
eng.profile_reset(); eng.profile_enable(True)
t0 = time.perf_counter()
for _ in range(20): eng.match(d0, d1)
dt = (time.perf_counter() - t0) / 20
eng.profile_enable(False)
print("ms per call", dt * 1e3, len(d0), len(d1))
for k in bench.KERNEL_GROUPS:
    n, ms = eng.profile_get(k)
    if n: print(k, n, round(ms / 20, 4))
