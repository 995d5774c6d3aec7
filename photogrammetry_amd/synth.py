"""Synthetic inputs for tests and bench.py: corner-rich RGBA64 frames and
descriptor sets.  Pure data generation on the host, seeded; no reference code or data."""
import numpy as np


def make_frame(W, H, seed, n_shapes=None, background=0.5):
    """Seeded field of filled rectangles (axis-aligned and rotated) and 4-point stars of random
    grey on a mid-grey ground -> uint16 [H][W][4] (R=G=B, A=65535)."""
    rng = np.random.default_rng(seed)
    if n_shapes is None:
        n_shapes = max(8, (W * H) // 300)
    img = np.full((H, W), background, dtype=np.float32)
    for _ in range(n_shapes):
        cx, cy = rng.integers(0, W), rng.integers(0, H)
        hw, hh = rng.integers(3, 14), rng.integers(3, 14)
        grey = rng.choice([0.0, 0.15, 0.3, 0.7, 0.85, 1.0])
        kind = rng.integers(0, 3)
        r = int(max(hw, hh) * 1.5) + 1
        x0, x1 = max(0, cx - r), min(W, cx + r + 1)
        y0, y1 = max(0, cy - r), min(H, cy + r + 1)
        if x0 >= x1 or y0 >= y1:
            continue
        yy, xx = np.mgrid[y0:y1, x0:x1]
        dx, dy = (xx - cx).astype(np.float32), (yy - cy).astype(np.float32)
        if kind == 0:
            m = (np.abs(dx) <= hw) & (np.abs(dy) <= hh)
        elif kind == 1:
            a = rng.uniform(0, np.pi)
            u, v = dx * np.cos(a) + dy * np.sin(a), -dx * np.sin(a) + dy * np.cos(a)
            m = (np.abs(u) <= hw) & (np.abs(v) <= hh)
        else:
            m = (np.abs(dx) * hh + np.abs(dy) * hw <= hw * hh) | ((np.abs(dx) <= 1) & (np.abs(dy) <= hh * 1.4)) \
                | ((np.abs(dy) <= 1) & (np.abs(dx) <= hw * 1.4))
        img[y0:y1, x0:x1][m] = grey
    v = np.round(img * 65535.0).astype(np.uint16)
    out = np.empty((H, W, 4), dtype=np.uint16)
    out[..., 0] = v
    out[..., 1] = v
    out[..., 2] = v
    out[..., 3] = 65535
    return out


def shift_frame(frame, dx, dy, background=0.5):
    """Translate by (+dx, +dy) pixels, filling with the ground grey (how the reference made
    15pt_star_shifted_150.png: python_src/scripts/image_editing.py:4-15)."""
    H, W = frame.shape[:2]
    out = np.empty_like(frame)
    out[..., :3] = np.uint16(round(background * 65535.0))
    out[..., 3] = 65535
    sx0, sx1 = max(0, -dx), min(W, W - dx)
    sy0, sy1 = max(0, -dy), min(H, H - dy)
    if sx0 < sx1 and sy0 < sy1:
        out[sy0 + dy:sy1 + dy, sx0 + dx:sx1 + dx] = frame[sy0:sy1, sx0:sx1]
    return out


def random_descriptors(n, words, seed):
    """Uniform random descriptors: worst case for greedy rounds (distances ~ Binomial(P, 1/2))."""
    rng = np.random.default_rng(seed)
    return rng.integers(0, 2**32, size=(n, words), dtype=np.uint32)


def true_match_descriptors(n, words, seed, flip=0.15):
    """(set1, set2, perm): set2 = permuted set1 with `flip` of the bits flipped."""
    rng = np.random.default_rng(seed)
    d1 = rng.integers(0, 2**32, size=(n, words), dtype=np.uint32)
    bits = np.unpackbits(d1.view(np.uint8), axis=1)
    noise = (rng.random(bits.shape) < flip).astype(np.uint8)
    perm = rng.permutation(n)
    d2 = np.packbits(bits ^ noise, axis=1).view(np.uint32)[perm]
    return d1, np.ascontiguousarray(d2), perm
