#!/usr/bin/env python3
"""bench.py -- descriptor pairs matched per second on MI355X (BASELINE.json's metric).

Workload (BASELINE.json configs[1]): 1920x1080 RGBA64 frame pairs, dewarp -> gray -> FAST-like
detect -> NMS (r=16) -> BRIEF-256 -> all-pairs Hamming match with the reference's greedy
assignment, 4096 keypoints per frame (lists truncated to their first 4096 in NMS order: a harness
choice, the reference has no cap).  One "step" = one batch of B independent image pairs per GPU
(--pairs-per-step, default 64), frames already resident in HBM; value = sum over pairs of N1*N2
divided by the WHOLE step time (detect + match), max over ranks.  Weak scaling: every rank
processes its own B pairs; the per-pair match lists are all-gathered (RCCL) once after the timed
region's last step -- they are the input of the (host-side) track graph.

Extra objects on the JSON line:
  roofline     the dominant kernel of the step by measured time (HIP events around its launches,
               recorded on the launch stream during the timed steps)
  kernels      per-kernel-group launches / avg ms over the timed region
  match_only   pairs/s of the match stage alone (SURVEY 8d's definition of the metric)
  overlap      the same K steps timed again with --overlap-streams contexts in flight (informative:
               `value` is always the single-stream figure unless --streams says otherwise)
  cpu_baseline the CPU oracle (literal single-thread port of the C#) timed on a bounded sample; beside it the
               optimised matcher on one core and on --cpu-procs processes (N = 1 only)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W, H = 1920, 1080
P = 256
WORDS = 8
NKP = 4096
RADIUS = 16
THRESH = 0.1
I8_MFMA_PEAK_OPS = 5.0e15   # dense int8 MFMA, /opt/skills/guides/MI355X_MICROARCH.md (2x bf16 2.5 PF)
VALU_PEAK_LANEOPS = 256 * 128 * 2.4e9
HBM_PEAK = 8.0e12


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def make_inputs(pairs_per_step, rank, cache_dir="/tmp/pgx_bench_cache"):
    """2B frames: pair p = (base frame p mod NB, the same frame translated by a pair-specific offset), NB = min(B, 8)
    seeded base frames per rank.  Every frame of the batch is a different image."""
    from photogrammetry_amd import synth
    os.makedirs(cache_dir, exist_ok=True)
    nb = min(pairs_per_step, 8)
    bases = []
    for b in range(nb):
        seed = 1234 + 1000 * rank + b
        path = os.path.join(cache_dir, "frame_%dx%d_%d.npy" % (W, H, seed))
        f0 = None
        if os.path.exists(path):
            try:
                f0 = np.load(path)
                if f0.shape != (H, W, 4) or f0.dtype != np.uint16:
                    f0 = None
            except (OSError, ValueError):
                f0 = None   # a torn file from an interrupted run: regenerate
        if f0 is None:
            f0 = synth.make_frame(W, H, seed=seed, n_shapes=20000)
            try:
                tmp = "%s.%d.tmp.npy" % (path, os.getpid())
                np.save(tmp, f0)
                os.replace(tmp, path)   # atomic: another rank or run never sees a partial file
            except OSError:
                pass
        bases.append(f0)
    frames = []
    for p in range(pairs_per_step):
        k = p // nb
        base = bases[p % nb] if k == 0 else synth.shift_frame(bases[p % nb], -13 * k, 7 * k)
        frames.append(base)
        frames.append(synth.shift_frame(base, 37 + 5 * k, 11 + 3 * k))
    return np.stack(frames)  # [2B][H][W][4]


def _cpu_pair_worker(path):
    """One process of the multi-core CPU bar: the oracle's optimised single-thread pipeline on the sample pair."""
    from oracle import cref
    z = np.load(path)
    t0 = time.time()
    descs = []
    for f in (z["f0"], z["f1"]):
        g = cref.gray(cref.apply_distortion(f, z["dmap"]))
        raw = cref.detect(g, np.float32(THRESH))
        kept = raw[cref.nms(raw, RADIUS)][:NKP]
        descs.append(cref.brief(g, np.stack([kept["x"], kept["y"]], 1), z["pairs"]))
    cref.match_sorted(descs[0], descs[1])
    return len(descs[0]) * len(descs[1]), time.time() - t0


def cpu_multicore(frames, dmap, pairs, nproc):
    """nproc processes, each running the optimised single-thread pipeline on the sample pair at the same time."""
    import concurrent.futures
    import multiprocessing
    import tempfile
    d = tempfile.mkdtemp(prefix="pgx_cpu_")
    path = os.path.join(d, "pair0.npz")
    np.savez(path, f0=frames[0], f1=frames[1], dmap=dmap, pairs=pairs)
    # The workers are plain CPU processes: started with "spawn" (never fork a process that holds a HIP context) and
    # with an environment that keeps them off the GPU (no visible devices, no profiler/tool preloads).
    saved = dict(os.environ)
    try:
        for k in list(os.environ):
            if k in ("LD_PRELOAD", "HSA_TOOLS_LIB", "ROCP_TOOL_LIB", "ROCP_TOOL_LIBRARIES", "ROCPROFILER_REGISTER_FORCE_LOAD") or k.startswith("ROCPROF"):
                os.environ.pop(k)
        os.environ["HIP_VISIBLE_DEVICES"] = ""
        os.environ["ROCR_VISIBLE_DEVICES"] = ""
        ctx = multiprocessing.get_context("spawn")
        with concurrent.futures.ProcessPoolExecutor(max_workers=nproc, mp_context=ctx) as ex:
            list(ex.map(_cpu_pair_worker, [path] * nproc))           # start-up and first touch
            os.environ.clear()
            os.environ.update(saved)
            t0 = time.time()
            res = list(ex.map(_cpu_pair_worker, [path] * nproc))
            dt = time.time() - t0
        return {"value": sum(r[0] for r in res) / dt, "cores": nproc, "wall_s": dt,
                "what": "%d processes at once, each: detect chain of both sample frames + sorted-edge-scan greedy match "
                        "(hardware popcount); the multi-core CPU bar of SURVEY 8d" % nproc}
    finally:
        os.environ.clear()
        os.environ.update(saved)
        try:
            os.remove(path)
            os.rmdir(d)
        except OSError:
            pass


def cpu_baseline(frames, dmap, pairs, sample_n):
    """The oracle (literal C port, 1 thread) on a bounded sample of the same workload:
    detect chain on the two frames of pair 0, literal Theta(N^3) match on the first sample_n keypoints."""
    from oracle import cref
    t0 = time.time()
    descs = []
    for f in frames[:2]:
        g = cref.gray(cref.apply_distortion(f, dmap))
        raw = cref.detect(g, np.float32(THRESH))
        kept = raw[cref.nms(raw, RADIUS)][:NKP]
        descs.append(cref.brief(g, np.stack([kept["x"], kept["y"]], 1), pairs))
    t_detect = time.time() - t0
    n1, n2 = min(sample_n, len(descs[0])), min(sample_n, len(descs[1]))
    t0 = time.time()
    cref.match(descs[0][:n1], descs[1][:n2])
    t_match = time.time() - t0
    t0 = time.time()
    cref.match_sorted(descs[0][:n1], descs[1][:n2])   # same result from the sorted-edge-scan formulation
    t_sorted = time.time() - t0
    return {"value": n1 * n2 / (t_detect + t_match), "unit": "descriptor pairs/s", "cores": 1, "kind": "port",
            "optimised": {"value": n1 * n2 / (t_detect + t_sorted), "match_s": t_sorted,
                          "what": "same sample, matcher replaced by the oracle's sort-all-edges-then-scan greedy "
                                  "(hardware popcount, 1 thread): the fair single-core CPU bar of SURVEY 8d"},
            "sample": "pair 0 of the workload: dewarp+gray+detect+NMS+BRIEF of both 1920x1080 frames (%.2f s) + literal "
                      "Theta(N^3) greedy match of the first %dx%d keypoints (%.2f s); C restatement of the C# "
                      "(hardware popcount, so faster than the real BigInteger loop)" % (t_detect, n1, n2, t_match),
            "detect_s_per_frame": t_detect / 2, "match_s": t_match, "match_n": [n1, n2]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--pairs-per-step", type=int, default=64)
    ap.add_argument("--cpu-sample", type=int, default=4096)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-procs", type=int, default=16, help="processes of the multi-core CPU bar (0/1 = skip)")
    ap.add_argument("--no-dewarp", action="store_true")
    ap.add_argument("--streams", type=int, default=1, help="steps kept in flight (one pgx context + HIP stream each)")
    ap.add_argument("--overlap-streams", type=int, default=2,
                    help="after the timed region, time the same K steps again with this many contexts/streams in flight "
                         "and report it as `overlap` (0 = skip); `value` always comes from --streams")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import photogrammetry_amd as pg

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    B = args.pairs_per_step
    F = 2 * B

    t_setup = time.time()
    frames_h = make_inputs(B, rank)
    pairs = pg.make_brief_pairs(0, 50, P)
    dmap = None if args.no_dewarp else pg.build_dewarp_map(W, H, [3e-4, 1e-7, 0, 0, 0])
    NS = max(1, args.streams)
    NO = max(0, args.overlap_streams)
    CAP = 8192
    d_frames = torch.from_numpy(frames_h).to(dev)
    pairlist = torch.tensor([[2 * p, 2 * p + 1] for p in range(B)], dtype=torch.int32, device=dev)
    engs, bufs = [], []
    for _ in range(max(NS, NO)):
        e = pg.Engine(local_rank)          # own non-blocking HIP stream per context
        e.set_brief_pairs(pairs)
        e.set_detect_params(THRESH, RADIUS)
        e.set_capacity(1 << 18, 8192)
        e.set_dewarp_map(dmap)
        engs.append(e)
        bufs.append(dict(kp=torch.zeros((F, CAP, 4), dtype=torch.int32, device=dev),
                         desc=torch.zeros((F, CAP, WORDS), dtype=torch.int32, device=dev),
                         counts=torch.zeros(F, dtype=torch.int32, device=dev),
                         nraw=torch.zeros(F, dtype=torch.int32, device=dev),
                         out=torch.zeros((B, CAP, 3), dtype=torch.int32, device=dev)))
    eng = engs[0]
    d_counts, d_nraw, d_out = bufs[0]["counts"], bufs[0]["nraw"], bufs[0]["out"]
    torch.cuda.synchronize()
    log("[rank %d] setup %.1fs, %d frames resident (%.0f MB), %d stream(s)" % (rank, time.time() - t_setup, F, d_frames.numel() * 2 / 1e6, NS))
    step_no = [0]

    def step(ns=NS):
        k = step_no[0] % ns
        step_no[0] += 1
        e, b = engs[k], bufs[k]
        e.detect_batch_dev(d_frames, F, W, H, b["kp"], b["desc"], b["counts"], b["nraw"], CAP)
        e.match_batch_dev(b["desc"], b["counts"], CAP, WORDS, pairlist, B, b["out"], max_count=NKP)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(args.warmup, NS, NO)):
        step(max(NS, NO))
    torch.cuda.synchronize()
    step_no[0] = 0
    # NMS survivors above the output capacity only flag a truncation here; anything else is fatal
    try:
        eng.check_status()
    except pg.CapacityError as e:
        log("[rank %d] note: %s" % (rank, e))
    counts = d_counts.cpu().numpy()
    nraw = d_nraw.cpu().numpy()
    n_used = np.minimum(counts, NKP)
    pairs_per_step = int(sum(int(n_used[2 * p]) * int(n_used[2 * p + 1]) for p in range(B)))
    log("[rank %d] survivors per frame %s, raw %s" % (rank, counts.tolist(), nraw.tolist()))

    for e in engs[:NS]:
        e.profile_reset()
        e.profile_enable(True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    for e in engs[:NS]:
        e.profile_enable(False)

    # the same K steps again with NO contexts in flight (events off): what stream-level overlap of the narrow
    # kernels (scans, tails, the per-pair finish) is worth; reported beside `value`, never as `value`
    dt_overlap = None
    if NO > 1 and NO != NS:
        step_no[0] = 0
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step(NO)
        barrier()
        dt_overlap = time.perf_counter() - t0

    # one exchange step: every rank's match lists -> all ranks (input of the track graph)
    if world > 1:
        gathered = [torch.empty_like(d_out) for _ in range(world)]
        dist.all_gather(gathered, d_out)
        torch.cuda.synchronize()

    t = torch.tensor([dt, dt_overlap if dt_overlap is not None else 0.0], dtype=torch.float64, device=dev)
    tot_pairs = torch.tensor([pairs_per_step], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot_pairs, op=dist.ReduceOp.SUM)
    dt_max = float(t[0].item())
    dt_overlap_max = float(t[1].item())
    job_pairs_per_step = float(tot_pairs.item())

    if rank == 0:
        kern = {}
        for name in ("dewarp_gray", "fast", "nms", "brief", "match_init", "ham_argmin", "match_select", "tail_fill", "match_finish"):
            n, ms = 0, 0.0
            for e in engs[:NS]:
                n_e, ms_e = e.profile_get(name)
                n, ms = n + n_e, ms + ms_e
            if n:
                kern[name] = {"launches": n, "avg_ms": ms / n, "ms_per_step": ms / args.steps}
        rounds_wide, evals, evals0 = eng.match_stats()
        log("tail debug counters (pairs, sumR, sumC, rounds, row rescans, col rescans):", eng.debug_counters())
        step_ms = dt_max / args.steps * 1e3
        match_ms = sum(kern[k]["ms_per_step"] for k in ("match_init", "ham_argmin", "match_select", "tail_fill", "match_finish") if k in kern)
        detect_ms = sum(kern[k]["ms_per_step"] for k in ("dewarp_gray", "fast", "nms", "brief") if k in kern)
        npix = W * H
        n_raw_tot = float(nraw.sum())
        n_kept_tot = float(n_used.sum())

        def hbm(name, byts, what):
            t = kern[name]["ms_per_step"] * 1e-3
            return {"kernel": name, "bound": "hbm", "achieved": byts / t / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                    "frac": byts / t / HBM_PEAK, "traffic": traffic.get(name), "algorithmic": what}

        traffic = {}
        tpath = os.path.join(ROOT, "profiles", "r01_traffic.json")   # HBM bytes per frame from separate rocprofv3 --pmc passes
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
            for k, v in tj.get("bytes_per_frame", {}).items():
                traffic[k] = v * F
            for k, v in tj.get("bytes_per_pair", {}).items():
                traffic[k] = v * B
        rooflines = {}
        if "dewarp_gray" in kern:
            # 8 B gathered source + 4 B grey per pixel and frame; the 8 B/px map is read once per group of 4 frames
            # (k_image.hip FB = 4), i.e. 2 B/px/frame -- the minimum this kernel's blocking allows
            per_px = 12.0 if dmap is None else 14.0
            rooflines["dewarp_gray"] = hbm("dewarp_gray", per_px * npix * F,
                                           "%.0f B/pixel/frame (8 gathered source + 4 grey + 8 map per 4 frames) x %d px x %d frames per launch" % (per_px, npix, F))
        if "fast" in kern:
            nseg = H * ((W + 63) // 64)
            rooflines["fast"] = hbm("fast", (4.5 * npix + 8.0 * nseg) * F,
                                    "4 B/px grey read + 0.5 B/px ballot planes + 8 B per 64-px row segment (count, then raster "
                                    "offset), %d frames (2 launches; the raw lists are not materialised on this path)" % F)
        if "nms" in kern:
            rooflines["nms"] = hbm("nms", 8.0 * n_raw_tot + 4.0 * n_kept_tot,
                                   "integer/latency-bound stage on L2-resident lists: 8 B per raw hit in + 4 B per survivor out "
                                   "(%d raw hits, %d survivors per step)" % (n_raw_tot, n_kept_tot))
        if "brief" in kern:
            rooflines["brief"] = hbm("brief", (2.0 * P * 4 + 48.0) * n_kept_tot,
                                     "2*P*4 B gathered + 48 B written per survivor; the kernel is bound by the L2->L1 line rate of fully divergent "
                                     "4-byte gathers (one 128-B line per sample), not by HBM")
        if "ham_argmin" in kern:
            ops = evals * 2.0 * P
            t = kern["ham_argmin"]["ms_per_step"] * 1e-3
            rooflines["ham_argmin"] = {"kernel": "ham_argmin", "bound": "mfma", "achieved": ops / t / 1e12,
                                       "peak": I8_MFMA_PEAK_OPS / 1e12, "unit": "TOP/s", "frac": ops / t / I8_MFMA_PEAK_OPS,
                                       "traffic": traffic.get("ham_argmin"),
                                       "algorithmic": "2*P = 512 int8 ops per descriptor-pair evaluation x %d evaluations "
                                                      "per step over %d launches" % (evals, rounds_wide)}
        if detect_ms:
            byts = 24.0 * npix * F
            rooflines["detect_chain"] = {"kernel": "dewarp_gray+fast+nms+brief", "bound": "hbm",
                                         "achieved": byts / (detect_ms * 1e-3) / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                                         "frac": byts / (detect_ms * 1e-3) / HBM_PEAK, "traffic": None,
                                         "algorithmic": "SURVEY 8d: 24 B/pixel for the fused-minimum detect stream x %d frames" % F}
        dominant = max(kern, key=lambda k: kern[k]["ms_per_step"]) if kern else None
        roof = rooflines.get(dominant)
        result = {
            "metric": "descriptor pairs matched/sec", "value": job_pairs_per_step * args.steps / dt_max,
            "unit": "descriptor pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": step_ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32 xor/popcount (int8 MFMA when enabled); f32 grey",
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[1] x %d: %d independent 1920x1080 RGBA64 pairs per GPU per step, "
                                   "dewarp(%s)+gray+FAST(T=0.1)+NMS(r=%d)+BRIEF-256+greedy Hamming match, %d keypoints "
                                   "per frame (truncated to the first %d in NMS order)"
                                   % (B, B, "off" if dmap is None else "shipped coeffs", RADIUS, NKP, NKP),
                       "pairs_per_step_per_gpu": B, "streams": NS, "keypoints": [int(x) for x in n_used.tolist()],
                       "raw_hits": [int(x) for x in nraw.tolist()], "parallelism": "pair-sharded x%d" % world},
            "roofline": roof,
            "rooflines": rooflines,
            "kernels": kern,
            "overlap": ({"streams": NO, "value": job_pairs_per_step * args.steps / dt_overlap_max,
                         "ms_per_step": dt_overlap_max / args.steps * 1e3,
                         "note": "same K steps with %d contexts/HIP streams in flight, events off" % NO}
                        if dt_overlap_max > 0 else None),
            "detect": {"ms_per_step": detect_ms, "frames_per_s": F / (detect_ms * 1e-3) if detect_ms else None},
            "match_only": {"ms_per_step": match_ms,
                           "pairs_per_s": pairs_per_step / (match_ms * 1e-3) if match_ms else None,
                           "wide_rounds": rounds_wide, "evaluations_per_step": evals,
                           "mfma_frac_of_peak_on_match_stage": (evals * 2.0 * P / (match_ms * 1e-3)) / I8_MFMA_PEAK_OPS
                           if match_ms else None},
        }
        if not args.no_cpu_baseline and world == 1:   # rank 0 at N = 1 only: at N > 1 the other ranks would sit in the barrier
            t1 = time.time()
            result["cpu_baseline"] = cpu_baseline(frames_h, dmap if dmap is not None else
                                                  np.stack(np.meshgrid(np.arange(W), np.arange(H)), axis=2).astype(np.int32),
                                                  pairs, args.cpu_sample)
            result["cpu_baseline"]["host_cpus"] = os.cpu_count()
            if args.cpu_procs > 1:
                try:
                    result["cpu_baseline"]["optimised_multicore"] = cpu_multicore(
                        frames_h, dmap if dmap is not None else np.stack(np.meshgrid(np.arange(W), np.arange(H)), axis=2).astype(np.int32),
                        pairs, min(args.cpu_procs, os.cpu_count() or 1))
                except Exception as e:   # a report nicety, never a reason to lose the bench line
                    log("multi-core CPU bar skipped: %r" % (e,))
            log("cpu baseline took %.1fs" % (time.time() - t1))
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    for e in engs:
        e.close()


if __name__ == "__main__":
    main()
