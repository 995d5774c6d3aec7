// pgx_tracks.cpp -- the track graph over match lists, HOST form (no GPU work): the small-input twin of k_tracks.hip.
//
// north_star names "a single RCCL all-gather ... to collect per-pair match lists into the global track graph"; the
// reference has no multi-frame structure at all, so the semantics are the build's own and are stated in
// include/pgx.h: union-find over (frame, keypoint) nodes; a match (k1, k2, dist) of image pair (a, b) links (a, k1) with
// (b, k2) when dist <= max_dist (the distance gate of python_src/scripts/match_keypoints.py:23,127 and of the commented
// `new KeypointMatching(100)`, Photogrammetry/Program.cs:165,224); tracks are the connected components; a component
// holding two keypoints of one frame is dropped as a whole.  Nothing depends on the order the pairs are added in.
#include <algorithm>
#include <cstdint>
#include <new>
#include <vector>

#include "../../include/pgx.h"

struct pgx_tracks {
    std::vector<int32_t> counts;
    std::vector<int64_t> base;     // node id of (frame, 0); ids ascend with (frame, keypoint)
    std::vector<int64_t> parent;
    // finished form
    bool finished = false;
    std::vector<int32_t> offsets, nodes;
    int dropped = 0, dropped_nodes = 0;

    int64_t find(int64_t x)
    {
        while (parent[x] != x) { parent[x] = parent[parent[x]]; x = parent[x]; }
        return x;
    }
};

extern "C" {

int pgx_tracks_create(const int32_t *counts, int n_frames, pgx_tracks **out)
{
    if (!counts || !out || n_frames < 0) return PGX_E_BADARG;
    pgx_tracks *t = new (std::nothrow) pgx_tracks();
    if (!t) return PGX_E_HIP;
    t->counts.assign(counts, counts + n_frames);
    t->base.resize(n_frames + 1);
    int64_t n = 0;
    for (int f = 0; f < n_frames; f++) {
        if (counts[f] < 0) { delete t; return PGX_E_BADARG; }
        t->base[f] = n;
        n += counts[f];
    }
    t->base[n_frames] = n;
    t->parent.resize(n);
    for (int64_t x = 0; x < n; x++) t->parent[x] = x;
    *out = t;
    return PGX_OK;
}

void pgx_tracks_destroy(pgx_tracks *t) { delete t; }

int pgx_tracks_add_pair(pgx_tracks *t, int frame_a, int frame_b, const pgx_pair *matches, int n, int max_dist)
{
    if (!t || (n > 0 && !matches) || n < 0) return PGX_E_BADARG;
    const int nf = (int)t->counts.size();
    if (frame_a < 0 || frame_a >= nf || frame_b < 0 || frame_b >= nf) return PGX_E_BADARG;
    if (n > t->counts[frame_a]) n = t->counts[frame_a];   // a list has counts[frame_a] entries (KeypointMatching.cs:38)
    t->finished = false;
    for (int e = 0; e < n; e++) {
        const pgx_pair &m = matches[e];
        if (m.dist > max_dist || m.dist == PGX_DIST_NONE || m.k1 < 0 || m.k2 < 0 || m.k1 >= t->counts[frame_a] ||
            m.k2 >= t->counts[frame_b])
            continue;
        int64_t ra = t->find(t->base[frame_a] + m.k1), rb = t->find(t->base[frame_b] + m.k2);
        if (ra == rb) continue;
        if (ra < rb) std::swap(ra, rb);
        t->parent[ra] = rb;   // the smaller id stays root: a component's root is its first (frame, keypoint)
    }
    return PGX_OK;
}

int pgx_tracks_finish(pgx_tracks *t, int min_len, int *n_tracks, int *n_nodes)
{
    if (!t || !n_tracks || !n_nodes) return PGX_E_BADARG;
    if (min_len < 1) min_len = 1;
    const int64_t n = (int64_t)t->parent.size();
    const int nf = (int)t->counts.size();
    std::vector<int64_t> root(n);
    std::vector<int32_t> size(n, 0), last_frame(n, -1);
    std::vector<char> bad(n, 0);
    // nodes in id order = (frame, keypoint) order: a second node of the same frame in a component shows as last_frame == f
    for (int f = 0; f < nf; f++)
        for (int32_t k = 0; k < t->counts[f]; k++) {
            const int64_t x = t->base[f] + k, r = t->find(x);
            root[x] = r;
            size[r]++;
            if (last_frame[r] == f) bad[r] = 1;
            last_frame[r] = f;
        }
    t->dropped = t->dropped_nodes = 0;
    std::vector<int32_t> slot(n, -1);
    t->offsets.assign(1, 0);
    int ntr = 0;
    for (int64_t x = 0; x < n; x++) {   // roots in id order = tracks by their first node
        if (root[x] != x) continue;
        if (bad[x]) { t->dropped++; t->dropped_nodes += size[x]; continue; }
        if (size[x] < min_len) continue;
        slot[x] = ntr++;
        t->offsets.push_back(t->offsets.back() + size[x]);
    }
    t->nodes.assign((size_t)t->offsets.back() * 2, 0);
    std::vector<int32_t> fill(t->offsets.begin(), t->offsets.end() - 1);
    for (int f = 0; f < nf; f++)
        for (int32_t k = 0; k < t->counts[f]; k++) {
            const int64_t r = root[t->base[f] + k];
            if (slot[r] < 0) continue;
            const int32_t p = fill[slot[r]]++;
            t->nodes[(size_t)p * 2] = f;
            t->nodes[(size_t)p * 2 + 1] = k;
        }
    t->finished = true;
    *n_tracks = ntr;
    *n_nodes = t->offsets.back();
    return PGX_OK;
}

int pgx_tracks_get(pgx_tracks *t, int32_t *track_offsets, int32_t *nodes)
{
    if (!t || !track_offsets || (!nodes && !t->nodes.empty())) return PGX_E_BADARG;
    if (!t->finished) return PGX_E_NOT_CONFIGURED; // pgx_tracks_finish first
    std::copy(t->offsets.begin(), t->offsets.end(), track_offsets);
    if (!t->nodes.empty()) std::copy(t->nodes.begin(), t->nodes.end(), nodes);
    return PGX_OK;
}

int pgx_tracks_dropped(pgx_tracks *t, int *n_components, int *n_nodes)
{
    if (!t) return PGX_E_BADARG;
    if (!t->finished) return PGX_E_NOT_CONFIGURED;
    if (n_components) *n_components = t->dropped;
    if (n_nodes) *n_nodes = t->dropped_nodes;
    return PGX_OK;
}

} // extern "C"
