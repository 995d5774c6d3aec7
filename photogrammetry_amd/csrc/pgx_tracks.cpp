// pgx_tracks.cpp -- the global track graph over the gathered match lists (host side of libpgx.so, no GPU work).
//
// north_star names "a single RCCL all-gather ... to collect per-pair match lists into the global track graph"; the
// reference has no multi-frame structure at all, so this is a build-side addition with the
// simplest defensible semantics: union-find over (frame, keypoint) nodes; a match (k1, k2, dist) of image pair
// (a, b) links (a, k1) with (b, k2) when dist <= max_dist (the distance gate the live C# matcher lacks; an earlier
// C# version took one: `new KeypointMatching(100)` in commented code, Photogrammetry/Program.cs:165,224); a union
// that would put two keypoints of one frame into a track is refused (first come, in list order).
#include <algorithm>
#include <cstdint>
#include <new>
#include <vector>

#include "../../include/pgx.h"

struct pgx_tracks {
    std::vector<int32_t> counts;
    std::vector<int64_t> base;                 // node id of (frame, 0)
    std::vector<int64_t> parent;
    std::vector<std::vector<int32_t>> frames;  // per root: sorted frames present in its set (moved on union)
    // finished form
    std::vector<int32_t> offsets, nodes;

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
    t->frames.resize(n);
    for (int f = 0; f < n_frames; f++)
        for (int32_t k = 0; k < counts[f]; k++) {
            t->parent[t->base[f] + k] = t->base[f] + k;
            t->frames[t->base[f] + k].assign(1, f);
        }
    *out = t;
    return PGX_OK;
}

void pgx_tracks_destroy(pgx_tracks *t) { delete t; }

int pgx_tracks_add_pair(pgx_tracks *t, int frame_a, int frame_b, const pgx_pair *matches, int n, int max_dist)
{
    if (!t || (n > 0 && !matches) || n < 0) return PGX_E_BADARG;
    const int nf = (int)t->counts.size();
    if (frame_a < 0 || frame_a >= nf || frame_b < 0 || frame_b >= nf) return PGX_E_BADARG;
    for (int e = 0; e < n; e++) {
        const pgx_pair &m = matches[e];
        if (m.dist > max_dist || m.k1 < 0 || m.k2 < 0 || m.k1 >= t->counts[frame_a] || m.k2 >= t->counts[frame_b]) continue;
        const int64_t ra = t->find(t->base[frame_a] + m.k1), rb = t->find(t->base[frame_b] + m.k2);
        if (ra == rb) continue;
        std::vector<int32_t> &fa = t->frames[ra], &fb = t->frames[rb];
        // refuse when the sets share a frame (both lists are sorted)
        bool clash = false;
        for (size_t i = 0, j = 0; i < fa.size() && j < fb.size();) {
            if (fa[i] == fb[j]) { clash = true; break; }
            if (fa[i] < fb[j]) i++; else j++;
        }
        if (clash) continue;
        t->parent[rb] = ra;
        std::vector<int32_t> merged(fa.size() + fb.size());
        std::merge(fa.begin(), fa.end(), fb.begin(), fb.end(), merged.begin());
        fa.swap(merged);
        std::vector<int32_t>().swap(fb);
    }
    return PGX_OK;
}

int pgx_tracks_finish(pgx_tracks *t, int min_len, int *n_tracks, int *n_nodes)
{
    if (!t || !n_tracks || !n_nodes) return PGX_E_BADARG;
    const int64_t n = (int64_t)t->parent.size();
    // nodes in (frame, keypoint) order are in node-id order: a track's nodes come out sorted, and tracks are
    // ordered by their first node
    std::vector<int64_t> root(n), first(n, -1);
    std::vector<int32_t> size(n, 0);
    for (int64_t x = 0; x < n; x++) { root[x] = t->find(x); size[root[x]]++; }
    std::vector<int32_t> slot(n, -1);
    t->offsets.assign(1, 0);
    int ntr = 0;
    for (int64_t x = 0; x < n; x++) {
        const int64_t r = root[x];
        if (size[r] < min_len) continue;
        if (slot[r] < 0) { slot[r] = ntr++; t->offsets.push_back(t->offsets.back() + size[r]); }
    }
    t->nodes.assign((size_t)t->offsets.back() * 2, 0);
    std::vector<int32_t> fill(t->offsets.begin(), t->offsets.end() - 1);
    for (size_t f = 0; f + 1 < t->base.size(); f++)
        for (int32_t k = 0; k < t->counts[f]; k++) {
            const int64_t r = root[t->base[f] + k];
            if (slot[r] < 0) continue;
            const int32_t p = fill[slot[r]]++;
            t->nodes[(size_t)p * 2] = (int32_t)f;
            t->nodes[(size_t)p * 2 + 1] = k;
        }
    *n_tracks = ntr;
    *n_nodes = t->offsets.back();
    return PGX_OK;
}

int pgx_tracks_get(pgx_tracks *t, int32_t *track_offsets, int32_t *nodes)
{
    if (!t || !track_offsets || (!nodes && !t->nodes.empty())) return PGX_E_BADARG;
    if (t->offsets.empty()) return PGX_E_NOT_CONFIGURED; // pgx_tracks_finish first
    std::copy(t->offsets.begin(), t->offsets.end(), track_offsets);
    if (!t->nodes.empty()) std::copy(t->nodes.begin(), t->nodes.end(), nodes);
    return PGX_OK;
}

} // extern "C"
