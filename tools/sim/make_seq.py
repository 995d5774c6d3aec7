"""Descriptors of the bench sequence (bench.py's frames: base frame seed 4321 rolled by (3i, i)) made with the CPU
oracle, cached under /tmp/pgx_sim -- input of the matcher-tail simulations in this directory (design studies, not product)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
W, H, NKP, RADIUS, T = 1920, 1080, 4096, 16, np.float32(0.1)
OUT = "/tmp/pgx_sim/seq_desc.npz"

def one(i):
    import bench
    from oracle import cref
    import photogrammetry_amd as pg  # host helpers only (no GPU)
    base = bench.base_frame(W, H, 4321)
    f = np.roll(base, shift=(i % H, (3 * i) % W), axis=(0, 1))
    pairs = pg.make_brief_pairs(0, 50, 256)
    dmap = pg.build_dewarp_map(W, H, [3e-4, 1e-7, 0, 0, 0])
    g = cref.gray(cref.apply_distortion(f, dmap))
    raw = cref.detect(g, T)
    kept = raw[cref.nms(raw, RADIUS)][:NKP]
    d = cref.brief(g, np.stack([kept["x"], kept["y"]], 1), pairs)
    return i, d, np.stack([kept["x"], kept["y"]], 1)

if __name__ == "__main__":
    import multiprocessing as mp
    frames = [int(a) for a in sys.argv[1:]] or list(range(64))
    t0 = time.time()
    with mp.get_context("spawn").Pool(8) as p:
        res = p.map(one, frames)
    np.savez(OUT, frames=np.array(frames), **{"d%d" % i: d for i, d, _ in res}, **{"xy%d" % i: xy for i, _, xy in res})
    print("done", time.time() - t0, [len(d) for _, d, _ in res][:8])
