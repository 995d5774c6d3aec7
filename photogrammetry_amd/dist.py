"""Multi-GPU plumbing for the path: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU tests).

The reference handles exactly one image pair in one process (TestService.cs:80-96): nothing in
it couples image pairs, so the path shards with no data-path collective inside detect or match:

  phase 1  detect      frame f      -> rank f mod G        (independent per frame)
  phase 2  exchange    ONE all-gather of fixed-size per-frame records {count, descriptors}
  phase 3  match       image pair p -> rank p mod G        (independent per image pair)
  phase 4  exchange    ONE all-gather of the match lists; fixed size because the reference always
                       emits exactly N1 entries per image pair (KeypointMatching.cs:38)

The gathered lists feed the track graph (connected components over (frame, keypoint) nodes; the reference
does not have one), built on the device where the lists sit: pgx_tracks_dev, `ShardedSequence(tracks=...)`.
Only tensors cross this module; it never touches the oracle.
"""
import numpy as np
import torch
import torch.distributed as dist


def owner(index, world):
    return index % world


def local_items(n, rank, world):
    """Indices owned by `rank` under round-robin sharding."""
    return list(range(rank, n, world))


def slots(n, world):
    """Per-rank slot count (the last slots of some ranks are padding)."""
    return (n + world - 1) // world


def all_pairs(n_frames):
    """Ordered image pairs i < j of a sequence, the enumeration order of SURVEY 8d config 3."""
    return [(i, j) for i in range(n_frames) for j in range(i + 1, n_frames)]


def _world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def all_gather_slots(local, n_total):
    """local: [slots(n_total, G), ...] holding this rank's items in ownership order (item k of rank
    r is global index r + k*G).  Returns [n_total, ...] in global order on every rank.
    One fixed-size all_gather (no all-gatherv: records are padded to a common size)."""
    rank, world = _world()
    if world == 1:
        return local[:n_total].clone()
    gathered = [torch.empty_like(local) for _ in range(world)]
    dist.all_gather(gathered, local.contiguous())
    out = torch.empty((n_total,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    for r in range(world):
        idx = local_items(n_total, r, world)
        if idx:
            out[torch.as_tensor(idx, device=local.device)] = gathered[r][:len(idx)]
    return out


def exchange_descriptors(desc_local, counts_local, n_frames):
    """Phase 2.  desc_local [slots][cap][words] int32, counts_local [slots] int32 ->
    (desc_all [F][cap][words], counts_all [F]) identical on every rank."""
    return all_gather_slots(desc_local, n_frames), all_gather_slots(counts_local, n_frames)


def exchange_matches(out_local, n_pairs):
    """Phase 4.  out_local [slots][cap][3] int32 -> [M][cap][3] on every rank."""
    return all_gather_slots(out_local, n_pairs)


def tracks_host(counts_all, pair_list, matches_all, max_dist=64, min_len=2):
    """The track graph of gathered lists through the C ABI's HOST form (pgx_tracks_*: sequential, small inputs; what a host
    without device-resident lists calls) -> (tracks, dropped components, nodes in them).  The device form is
    ShardedSequence(tracks=...) / Engine.tracks_dev."""
    from . import api
    m = matches_all.cpu().numpy() if isinstance(matches_all, torch.Tensor) else np.asarray(matches_all)
    c = counts_all.cpu().numpy() if isinstance(counts_all, torch.Tensor) else np.asarray(counts_all)
    return api.tracks_host(c, pair_list, [m[p] for p in range(len(pair_list))], max_dist, min_len)


# ---- the four phases on this rank's GPU ------------------------------------------------------------

def slot_of(index, world, nslots):
    """Position of global item `index` in a rank-major gathered buffer [world][nslots]: its owner's
    block, then its place in the owner's list.  all_gather_into_tensor leaves the blocks in rank
    order, so nothing is permuted in memory: consumers address through this map."""
    return (index % world) * nslots + index // world


class ShardedSequence:
    """One multi-frame job on G ranks, device-resident end to end:

      phase 1  pgx_detect_batch_dev on the frames this rank owns (frame f -> rank f mod G), written
               straight into this rank's block of the gathered descriptor buffer
      phase 2  all-gather (in place) of the fixed-size per-frame records: descriptors + counts
      phase 3  pgx_match_batch_dev on the image pairs this rank owns (pair p -> rank p mod G)
      phase 4  all-gather (in place) of the fixed-size match lists (KeypointMatching.cs:38: always
               exactly N1 entries per image pair, padded to `nkp`)

    With G == 1 the collectives vanish and the same code runs.  All buffers are allocated once;
    `step()` only enqueues (pgx kernels and the collectives share `stream`).  Frames are addressed in
    the gathered buffers through slot_of(); the pair list handed to the matcher is pre-mapped."""

    def __init__(self, engine, W, H, n_frames, pair_list, nkp, words, device, stream=None, group=None, comm="torch",
                 overlap_exchange=False, tracks=None):
        """comm = "torch": the two exchanges are torch.distributed all_gather_into_tensor calls (RCCL under the "nccl"
        backend, gloo in the CPU tests); comm = "pgx": the whole step is ONE C-ABI call, pgx_sequence_step_dev, on the
        context's own RCCL communicator (engine.comm_init must have run; what a non-Python host would use).
        overlap_exchange (torch path, G > 1): the all-gather of the match lists -- by far the larger exchange, 12 B x nkp
        per image pair -- is issued asynchronously into one of TWO output buffers and awaited one step later, so it runs
        beside the next step's detect and match instead of after this step's; `finish()` awaits the last one and must be
        called before the lists are read.  `out_all` / `matches()` then refer to the step issued last.
        Both exchanges stay on ONE communicator on purpose: collectives of one communicator run in issue order, so the
        next step's (small) descriptor gather queues behind the previous step's list gather -- at 8 GPUs that is an
        estimated 0.4 ms of a 8.8 ms step (792 MB of lists at ~310 GB/s bus bandwidth = 2.2 ms against 1.75 ms of
        detect), nothing at <= 4 GPUs.  A second communicator would remove it, but two RCCL communicators with
        collectives in flight at once cannot be rehearsed on the one-GPU development box, and a hang costs the whole
        run.
        tracks = None | dict(max_dist=..., min_len=2, frames=None): build the track graph (pgx_tracks_dev) of every step's gathered
        lists on this rank's GPU, on the job's stream, as soon as the lists are complete (right behind the matcher at G == 1,
        behind the list gather otherwise -- with overlap_exchange that is one step later, or in finish()).  `frames` = the
        global frame numbers this rank's graph covers, numbered 0.. in that order (default: all frames, i.e. every rank
        builds the whole graph; a caller whose job is several independent sequences gives each rank its own sequences'
        frames: image pairs touching other frames are skipped).  Results: track_of / trk_offsets / trk_nodes / trk_summary
        (device), tracks() / track_summary() (host)."""
        assert comm in ("torch", "pgx")
        self.comm = comm
        self.overlap = bool(overlap_exchange) and comm == "torch"
        self._pending = None
        self.e, self.W, self.H, self.nkp, self.words = engine, W, H, nkp, words
        self.rank, self.world = _world()
        self.group = group
        self.n_frames, self.pair_list = n_frames, list(pair_list)
        G = self.world
        self.fs = slots(n_frames, G)
        self.ps = slots(len(self.pair_list), G)
        self.my_frames = local_items(n_frames, self.rank, G)
        self.my_pairs = local_items(len(self.pair_list), self.rank, G)
        device = torch.device(device)
        self.on_gpu = device.type == "cuda"   # CPU tensors + gloo: the world_size-2 tests, with a stand-in engine
        self.stream = (stream if stream is not None else torch.cuda.current_stream(device)) if self.on_gpu else None
        i32 = dict(dtype=torch.int32, device=device)
        self.desc_all = torch.zeros((G * self.fs, nkp, words), **i32)
        self.counts_all = torch.zeros(G * self.fs, **i32)
        self.out_bufs = [torch.zeros((G * self.ps, nkp, 3), **i32) for _ in range(2 if (self.overlap and G > 1) else 1)]
        self._cur = 0
        # the ranks have agreed that the local part of a step with THIS key went through (see front).  The key holds only what
        # is equal on every rank by construction -- slot counts, image size, capacity -- never this rank's share of the work
        # or its own error state: whether the status exchange runs must be the same decision on every rank
        # (pgx_sequence_step_dev takes it the same way, with the context's configuration epoch as part of the key)
        self._agreed_key = None
        self.out_all = self.out_bufs[0]
        self.kp_l = torch.zeros((self.fs, nkp, 4), **i32)
        self.nraw_l = torch.zeros(self.fs, **i32)
        lo_f, lo_p = self.rank * self.fs, self.rank * self.ps
        self.desc_l = self.desc_all[lo_f:lo_f + self.fs]
        self.counts_l = self.counts_all[lo_f:lo_f + self.fs]
        self.out_l = self.out_all[lo_p:lo_p + self.ps]
        mapped = [[slot_of(a, G, self.fs), slot_of(b, G, self.fs)] for a, b in (self.pair_list[p] for p in self.my_pairs)]
        self.pairlist_l = torch.tensor(mapped if mapped else [[0, 0]], **i32)
        if self.on_gpu:
            engine.set_stream(self.stream.cuda_stream)
        self.trk = None
        if tracks is not None:
            self.trk = {"max_dist": int(tracks.get("max_dist", 64)), "min_len": int(tracks.get("min_len", 2))}
            frames = list(tracks["frames"]) if tracks.get("frames") is not None else list(range(n_frames))
            ids = np.full(G * self.fs, -1, dtype=np.int32)
            for i, f in enumerate(frames):
                ids[slot_of(f, G, self.fs)] = i
            self.trk_frames = frames
            nfg = max(1, len(frames))
            self.trk_frame_ids = torch.from_numpy(ids).to(device)
            # the pair list of ALL image pairs in the row order of the gathered list buffer (padding rows name no frame)
            rows = np.full((G * self.ps, 2), -1, dtype=np.int32)
            for p, (a, b) in enumerate(self.pair_list):
                rows[slot_of(p, G, self.ps)] = (slot_of(a, G, self.fs), slot_of(b, G, self.fs))
            self.trk_pairlist = torch.from_numpy(rows).to(device)
            self.track_of = torch.zeros((nfg, nkp), **i32)
            self.trk_offsets = torch.zeros(nfg * nkp + 1, **i32)
            self.trk_nodes = torch.zeros((nfg * nkp, 2), **i32)
            self.trk_summary = torch.zeros(8, **i32)
            # the counts a step's lists were made with, kept per output buffer: with the overlapped exchange the next step's
            # detect has overwritten counts_all by the time the lists are complete
            self.trk_counts = [torch.zeros(G * self.fs, **i32) for _ in self.out_bufs]

    def _build_tracks(self, buf):
        """Enqueue the track graph over the complete lists in out_bufs[buf] (on the job's stream)."""
        self.e.tracks_dev(self.out_bufs[buf], self.trk_counts[buf], self.trk_pairlist, self.world * self.ps, self.world * self.fs,
                          self.nkp, max(1, len(self.trk_frames)), self.trk["max_dist"], self.trk["min_len"], self.track_of,
                          self.trk_offsets, self.trk_nodes, self.trk_summary, d_frame_ids=self.trk_frame_ids)

    def step(self, d_frames_local, after=None, n_local=None):
        """d_frames_local: uint16 [len(my_frames)][H][W][4] resident on this rank's GPU.
        after = (engine, detect_stage, match_stage): another job's engine on the same GPU (two jobs kept in flight, each on its
        own stream) and the stages of ITS most recent step that this step's detect chain / matcher wait for (PGX_STAGE_* or
        None; pgx_wait_stage / pgx_gate_match).  Ordering only.
        step = front + back.  A caller that keeps two jobs in flight on G > 1 ranks issues the halves interleaved --
        front(s + 1) BEFORE back(s) -- so that on the communicator (whose collectives run in issue order) the small descriptor
        gather of step s + 1 stands in front of the large match-list gather of step s instead of behind it: the matcher of step
        s + 1 then never waits for step s's lists to cross the links (792 MB per step at 8 GPUs)."""
        if self.comm == "pgx":   # ONE C call does all four phases
            nf, npr = len(self.my_frames), len(self.my_pairs)
            other, gate_detect, gate_match = after if after is not None else (None, None, None)
            if other is not None and self.on_gpu:
                if gate_detect is not None:
                    self.e.wait_stage(other, gate_detect)
                if gate_match is not None:
                    self.e.gate_match(other, gate_match)
            self.e.sequence_step_dev(d_frames_local, nf, self.fs, self.W, self.H, self.kp_l, self.desc_all, self.counts_all,
                                     self.nraw_l, self.nkp, self.pairlist_l, npr, self.ps, self.out_all)
            if self.trk is not None:
                self.trk_counts[0].copy_(self.counts_all)
                self._build_tracks(0)
            return
        self.front(d_frames_local, after, n_local)
        self.back(after)

    def front(self, d_frames_local, after=None, n_local=None):
        """Phases 1 and 2: detect this rank's frames, gather the descriptor records (comm = "torch").
        n_local: detect only the first n_local of this rank's frames (a short last batch on one rank; the other slots keep
        what they held) -- a rank-local quantity that must not change which collectives are issued."""
        import contextlib
        assert self.comm == "torch"
        nf = len(self.my_frames) if n_local is None else max(0, min(int(n_local), len(self.my_frames)))
        other, gate_detect, _ = after if after is not None else (None, None, None)
        if other is not None and gate_detect is not None and self.on_gpu:
            self.e.wait_stage(other, gate_detect)
        with (torch.cuda.stream(self.stream) if self.on_gpu else contextlib.nullcontext()):
            # A rank whose local part fails before the first collective (not configured, a size mismatch, no memory for a
            # workspace: all of them properties of the configuration, raised by the first call) must not leave its peers
            # waiting in the all-gather: the first step ends its local part with a one-word status exchange, and every rank
            # raises if any rank failed.  Later steps with the same buffers cannot fail that way and skip the exchange (it
            # would cost a host synchronisation per step).
            key = (self.fs, self.ps, self.W, self.H, self.nkp, self.words)
            agreed = self._agreed_key == key
            local_error = None
            try:
                if nf:
                    self.e.detect_batch_dev(d_frames_local, nf, self.W, self.H, self.kp_l, self.desc_l, self.counts_l,
                                            self.nraw_l, self.nkp)
            except Exception as ex:  # noqa: BLE001 -- whatever it is, the peers have to hear about it
                if self.world == 1 or agreed:
                    raise   # under an agreed key the peers issue the step's gathers, not a status exchange: the host must stop the job
                local_error = ex
            if self.world > 1 and not agreed:
                st = torch.full((1,), 0 if local_error is None else 1, dtype=torch.int32, device=self.desc_all.device)
                st_all = torch.zeros(self.world, dtype=torch.int32, device=self.desc_all.device)
                dist.all_gather_into_tensor(st_all, st, group=self.group)
                bad = [r for r, v in enumerate(st_all.cpu().tolist()) if v]
                if local_error is not None:
                    raise local_error
                if bad:
                    raise RuntimeError("rank(s) %s failed before the first exchange; no collective was started" % bad)
                self._agreed_key = key
            if self.world > 1:
                dist.all_gather_into_tensor(self.desc_all, self.desc_l, group=self.group)
                dist.all_gather_into_tensor(self.counts_all, self.counts_l, group=self.group)

    def back(self, after=None):
        """Phases 3 and 4: match this rank's image pairs, gather the match lists (comm = "torch")."""
        import contextlib
        assert self.comm == "torch"
        npr = len(self.my_pairs)
        other, _, gate_match = after if after is not None else (None, None, None)
        with (torch.cuda.stream(self.stream) if self.on_gpu else contextlib.nullcontext()):
            out_all = self.out_bufs[self._cur]
            lo_p = self.rank * self.ps
            out_l = out_all[lo_p:lo_p + self.ps]
            if other is not None and gate_match is not None and self.on_gpu and npr:
                self.e.gate_match(other, gate_match)   # inside the matcher call, behind its init kernel (one shot: armed only when that call follows)
            if npr:
                self.e.match_batch_dev(self.desc_all, self.counts_all, self.nkp, self.words, self.pairlist_l, npr,
                                       out_l, max_count=self.nkp)
            self.out_all, self.out_l = out_all, out_l
            if self.trk is not None:
                self.trk_counts[self._cur].copy_(self.counts_all)
            if self.world > 1:
                if self.overlap:
                    # the gather issued one step ago filled the OTHER buffer; it has had this whole step to complete, and
                    # the next step's matcher writes into that buffer, so the stream waits for it here
                    if self._pending is not None:
                        self._pending.wait()
                        if self.trk is not None:
                            self._build_tracks(self._cur ^ 1)   # the previous step's lists are complete now
                    self._pending = dist.all_gather_into_tensor(out_all, out_l, group=self.group, async_op=True)
                    self._cur ^= 1
                else:
                    dist.all_gather_into_tensor(out_all, out_l, group=self.group)
                    if self.trk is not None:
                        self._build_tracks(self._cur)
            elif self.trk is not None:
                self._build_tracks(self._cur)

    def finish(self):
        """Await the match-list exchange of the last step (overlap_exchange); a no-op otherwise."""
        import contextlib
        if self._pending is not None:
            with (torch.cuda.stream(self.stream) if self.on_gpu else contextlib.nullcontext()):
                self._pending.wait()
                if self.trk is not None:
                    self._build_tracks(self._cur ^ 1)   # the last step's lists
            self._pending = None

    # -- views for consumers (host side) ---------------------------------------------------------
    def counts(self):
        """Per-frame counts in global frame order."""
        c = self.counts_all.cpu().numpy()
        return np.array([c[slot_of(f, self.world, self.fs)] for f in range(self.n_frames)], dtype=np.int32)

    def descriptors(self, frame):
        return self.desc_all[slot_of(frame, self.world, self.fs)]

    def matches(self, pair_index):
        return self.out_all[slot_of(pair_index, self.world, self.ps)]

    def track_summary(self):
        """pgx_tracks_dev's d_summary as a dict (synchronises)."""
        v = self.trk_summary.cpu().tolist()
        return {"n_tracks": v[0], "n_nodes": v[1], "dropped": v[2], "dropped_nodes": v[3], "edges": v[4], "longest": v[5],
                "largest_dropped": v[6]}

    def tracks(self):
        """The most recent graph as a list of tracks, each a list of (frame, keypoint) in pgx_tracks_get's order; frames are
        numbered as in the `frames` list given to the constructor (synchronises)."""
        nt, nn = (int(x) for x in self.trk_summary[:2].cpu().tolist())
        off = self.trk_offsets[:nt + 1].cpu().numpy()
        nodes = self.trk_nodes[:nn].cpu().numpy()
        return [[(int(f), int(k)) for f, k in nodes[off[t]:off[t + 1]]] for t in range(nt)]
