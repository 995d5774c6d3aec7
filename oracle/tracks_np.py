"""Sequential restatement of the track graph -- TEST INFRASTRUCTURE ONLY.

Imported by tests/ and by bench.py's checker leg; nothing under photogrammetry_amd/ imports this module.

PARITY UNPINNED BY CONSTRUCTION: the reference has no multi-frame structure at all (TestService.cs:80-96
handles exactly one image pair), so there is no reference output to pin against.  What the reference does hold is the
distance gate: the Python prototype keeps a match only if `dist <= --match-threshold` (python_src/scripts/
match_keypoints.py:23,127) and an earlier C# matcher took one (`new KeypointMatching(100)` in commented code,
dotnet_src/Photogrammetry/Program.cs:165,224).  The semantics below are the build's own, chosen so that ANY order of
processing the edges gives the same result (a parallel union-find can then be compared bit for bit):

  nodes   (frame, keypoint), keypoint < counts[frame]
  edges   entry e < counts[a] of image pair (a, b)'s match list links (a, k1) with (b, k2) when dist <= max_dist,
          dist != INT_MAX (the (0, 0, int.MaxValue) tail of KeypointMatching.cs:40-42 never links), k1 < counts[a],
          k2 < counts[b]
  tracks  connected components; a component that holds two keypoints of ONE frame is inconsistent and dropped as a
          whole (its nodes get track id -2); a consistent component with fewer than min_len nodes is no track (-1)
  order   tracks by their first (frame, keypoint), nodes inside a track ascending

Formulations that must agree (tests/test_dist_gloo.py, tests/test_gpu_tracks.py): `tracks()` is a plain sequential
union-find in list order, `tracks_csgraph()` hands the same edge set to scipy.sparse.csgraph.connected_components,
`tracks_arrays()` is the vectorised form for full-size jobs in pgx_tracks_dev's output layout.
"""
import numpy as np

INT_MAX = 2**31 - 1


def edges(counts, pair_list, lists, max_dist):
    """The gated edge list as ((fa, k1), (fb, k2)) tuples, in list order."""
    out = []
    for (a, b), rows in zip(pair_list, lists):
        rows = np.asarray(rows).reshape(-1, 3)[:int(counts[a])]
        for k1, k2, d in rows.tolist():
            if d > max_dist or d == INT_MAX or k1 < 0 or k2 < 0 or k1 >= counts[a] or k2 >= counts[b]:
                continue
            out.append(((int(a), int(k1)), (int(b), int(k2))))
    return out


def _finish(counts, groups, min_len):
    """groups: iterable of node lists (every node exactly once) -> (tracks, track_of, summary)."""
    F = len(counts)
    stride = max([int(c) for c in counts] + [1])
    track_of = -np.ones((F, stride), dtype=np.int32)
    kept, dropped, dropped_nodes, largest_dropped = [], 0, 0, 0
    for g in groups:
        g = sorted(g)
        frames = [f for f, _ in g]
        if len(set(frames)) != len(frames):           # two keypoints of one frame: inconsistent, dropped as a whole
            dropped += 1
            dropped_nodes += len(g)
            largest_dropped = max(largest_dropped, len(g))
            for f, k in g:
                track_of[f, k] = -2
        elif len(g) >= min_len:
            kept.append(g)
    kept.sort()
    for t, g in enumerate(kept):
        for f, k in g:
            track_of[f, k] = t
    summary = {"n_tracks": len(kept), "n_nodes": sum(len(g) for g in kept), "dropped": dropped,
               "dropped_nodes": dropped_nodes, "longest": max([len(g) for g in kept] + [0]),
               "largest_dropped": largest_dropped}
    return kept, track_of, summary


def tracks(counts, pair_list, lists, max_dist, min_len=2):
    """Sequential union-find over the gated edges in list order.  Returns (tracks, track_of [F][max count], summary);
    tracks = sorted list of sorted [(frame, keypoint)] lists."""
    counts = [int(c) for c in counts]
    parent = {(f, k): (f, k) for f, c in enumerate(counts) for k in range(c)}

    def find(x):
        while parent[x] != x:
            parent[x] = parent[parent[x]]
            x = parent[x]
        return x

    es = edges(counts, pair_list, lists, max_dist)
    for u, v in es:
        ru, rv = find(u), find(v)
        if ru != rv:
            parent[rv] = ru
    groups = {}
    for x in parent:
        groups.setdefault(find(x), []).append(x)
    kept, track_of, summary = _finish(counts, groups.values(), min_len)
    summary["edges"] = len(es)
    return kept, track_of, summary


def tracks_csgraph(counts, pair_list, lists, max_dist, min_len=2):
    """The same through scipy's connected_components (an independent formulation of the component step)."""
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import connected_components
    counts = [int(c) for c in counts]
    base = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    n = int(base[-1])
    es = edges(counts, pair_list, lists, max_dist)
    if n == 0:
        return _finish(counts, [], min_len)
    u = np.array([base[f] + k for (f, k), _ in es], dtype=np.int64)
    v = np.array([base[f] + k for _, (f, k) in es], dtype=np.int64)
    g = coo_matrix((np.ones(len(es), dtype=np.int8), (u, v)), shape=(n, n))
    _, lab = connected_components(g, directed=False)
    groups = {}
    for f, c in enumerate(counts):
        for k in range(c):
            groups.setdefault(int(lab[base[f] + k]), []).append((f, k))
    kept, track_of, summary = _finish(counts, groups.values(), min_len)
    summary["edges"] = len(es)
    return kept, track_of, summary


def tracks_arrays(counts, pair_list, matches, stride, max_dist, min_len=2):
    """Vectorised form for full-size jobs (numpy + scipy; the component step is scipy's): the same semantics, results in the
    layout pgx_tracks_dev writes.  counts [F]; pair_list [M][2]; matches [M][stride][3] int32.
    -> (offsets [n_tracks + 1], nodes [n_nodes][2], track_of [F][stride], summary dict)."""
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import connected_components
    counts = np.asarray(counts, dtype=np.int64)
    pl = np.asarray(pair_list, dtype=np.int64).reshape(-1, 2)
    m = np.asarray(matches).reshape(len(pl), stride, 3)
    F = len(counts)
    N = F * stride
    ca, cb = counts[pl[:, 0]][:, None], counts[pl[:, 1]][:, None]
    e = np.arange(stride)[None, :]
    k1, k2, d = m[..., 0].astype(np.int64), m[..., 1].astype(np.int64), m[..., 2].astype(np.int64)
    ok = (e < ca) & (d <= max_dist) & (d != INT_MAX) & (k1 >= 0) & (k2 >= 0) & (k1 < ca) & (k2 < cb)
    u = (pl[:, 0][:, None] * stride + k1)[ok]
    v = (pl[:, 1][:, None] * stride + k2)[ok]
    g = coo_matrix((np.ones(len(u), dtype=np.int8), (u, v)), shape=(N, N))
    _, lab = connected_components(g, directed=False)
    valid = (np.arange(stride)[None, :] < counts[:, None]).reshape(-1)
    ids = np.nonzero(valid)[0]
    lab_v = lab[ids]
    # canonical label = smallest node id of the component
    first = np.full(lab.max() + 1, N, dtype=np.int64)
    np.minimum.at(first, lab_v, ids)
    root = first[lab_v]
    size = np.bincount(root, minlength=N)
    frame = ids // stride
    key = root * F + frame
    uniq, cnt = np.unique(key, return_counts=True)
    bad_roots = np.unique(uniq[cnt > 1] // F)
    bad = np.zeros(N, dtype=bool)
    bad[bad_roots] = True
    roots = np.nonzero(size > 0)[0]   # size is indexed by root id: non-zero exactly at the roots, ascending = by first node
    kept_roots = roots[(~bad[roots]) & (size[roots] >= max(1, min_len))]
    tidx = np.full(N, -1, dtype=np.int64)
    tidx[kept_roots] = np.arange(len(kept_roots))
    offsets = np.concatenate([[0], np.cumsum(size[kept_roots])]).astype(np.int32)
    t_of_node = tidx[root]
    keep = t_of_node >= 0
    o = np.lexsort((ids[keep], t_of_node[keep]))
    kn = ids[keep][o]
    nodes = np.stack([kn // stride, kn % stride], axis=1).astype(np.int32)
    track_of = np.full(N, -1, dtype=np.int32)
    track_of[ids] = np.where(bad[root], -2, t_of_node).astype(np.int32)
    summary = {"n_tracks": int(len(kept_roots)), "n_nodes": int(offsets[-1]), "dropped": int(len(bad_roots)),
               "dropped_nodes": int(size[bad_roots].sum()), "edges": int(ok.sum()),
               "longest": int(size[kept_roots].max()) if len(kept_roots) else 0,
               "largest_dropped": int(size[bad_roots].max()) if len(bad_roots) else 0}
    return offsets, nodes, track_of.reshape(F, stride), summary
