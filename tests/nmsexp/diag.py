"""Developer diagnostic for the NMS round kernel: runs the repeated 8-frame detect of
tests/test_gpu_sequence.py against the library named by PGX_LIB and lists every survivor the oracle does not keep,
with the kept point that should have suppressed it (cell offset, whether the GPU kept it too) and the value the
experiment builds of build_variants.sh stamp into the score field.  The oracle is the checker only."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import photogrammetry_amd as pg
from oracle import cref
from photogrammetry_amd import synth

T, DEV = np.float32(0.1), "cuda:0"
radius = int(os.environ.get("RADIUS", "16"))
W, H, F, CAP = 1920, 1080, 8, 8192
e = pg.Engine(0)
pairs = pg.make_brief_pairs(0, 50, 256)
dmap = pg.build_dewarp_map(W, H, [3e-4, 1e-7, 0, 0, 0])
e.set_brief_pairs(pairs); e.set_detect_params(T, radius); e.set_capacity(1 << 18, CAP); e.set_dewarp_map(dmap)
base = synth.make_frame(W, H, seed=4321, n_shapes=20000)
d_base = torch.from_numpy(base).to(DEV)
d_frames = torch.empty((F, H, W, 4), dtype=torch.uint16, device=DEV)
for i in range(F):
    d_frames.view(torch.int64)[i] = torch.roll(d_base.view(torch.int64), shifts=(i % H, (3 * i) % W), dims=(0, 1))
d_kp = torch.zeros((F, CAP, 4), dtype=torch.int32, device=DEV)
d_desc = torch.zeros((F, CAP, 8), dtype=torch.int32, device=DEV)
d_cnt = torch.zeros((F,), dtype=torch.int32, device=DEV)
d_nraw = torch.zeros((F,), dtype=torch.int32, device=DEV)
torch.cuda.synchronize()
frames_h = d_frames.cpu().numpy()
expect, raws = [], []
for f in range(F):
    g = cref.gray(cref.apply_distortion(frames_h[f], dmap))
    raw = cref.detect(g, T)
    raws.append(raw)
    expect.append(raw[cref.nms(raw, radius)])
nbad = 0
for rep in range(int(os.environ.get("REPS", "6"))):
    e.detect_batch_dev(d_frames.data_ptr(), F, W, H, d_kp.data_ptr(), d_desc.data_ptr(), d_cnt.data_ptr(), d_nraw.data_ptr(), CAP)
    e.check_status()
    cnt = d_cnt.cpu().numpy(); kp = d_kp.cpu().numpy()
    for f in range(F):
        kept = expect[f]
        rnd = {(int(x), int(y)): int(s) >> 5 for x, y, s in kp[f, :cnt[f], :3]}  # experiment builds stamp the lane's need mask
        got = [(int(x), int(y), int(s) & 31) for x, y, s in kp[f, :cnt[f], :3]]
        exp = [(int(x), int(y), int(s)) for x, y, s in zip(kept["x"], kept["y"], kept["fast_score"])]
        if got == exp:
            continue
        nbad += 1
        sg, se = set(got), set(exp)
        extra, missing = sorted(sg - se), sorted(se - sg)
        dups = len(got) - len(sg)
        print(f"rep {rep} frame {f}: got {len(got)} exp {len(exp)} dups {dups} extra {len(extra)} missing {len(missing)}")
        for (x, y, s) in extra:
            # who should have suppressed it: accepted (expected) points within r that are better (higher score, or equal and raster-earlier)
            sup = [(ex, ey, es) for (ex, ey, es) in exp if (ex - x) ** 2 + (ey - y) ** 2 <= radius * radius and (es > s or (es == s and (ey, ex) < (y, x)))]
            print(f"  extra ({x},{y}) s={s} stamp={rnd.get((x, y), 0)} cell=({x>>3},{y>>3}) in-cell=({x&7},{y&7}); suppressors:",
                  [(ex, ey, es, (ex >> 3) - (x >> 3), (ey >> 3) - (y >> 3), "got" if (ex, ey, es) in sg else "NOTGOT") for ex, ey, es in sup])
        for (x, y, s) in missing:
            print(f"  missing ({x},{y}) s={s} cell=({x>>3},{y>>3})")
print("bad frames:", nbad)
