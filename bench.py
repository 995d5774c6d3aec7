#!/usr/bin/env python3
"""bench.py -- descriptor pairs matched per second on MI355X (BASELINE.json's metric).

    python bench.py --gpus N --steps K --warmup W

Headline workload (BASELINE configs[2] = SURVEY 8d config 3, the batched form of configs[1]): a 64-frame
1920x1080 RGBA64 sequence (frame_i = frame_0 translated by (3i, i) px), every frame through
dewarp -> gray -> FAST-like detect -> NMS (r = 16) -> BRIEF-256, then ALL ordered image pairs i < j (2016) through
the all-pairs Hamming distance + the reference's greedy assignment, <= 4096 keypoints per frame (lists cut to
their first 4096 in NMS order: a harness choice, the reference has no cap).  One "step" = one such job, frames
resident in HBM when the timed region starts; value = sum over image pairs of N1*N2 / whole step time
(detect + exchange + match + exchange), max over ranks.

N GPUs: one process per GPU.  `python bench.py --gpus N` is itself the launcher: the parent spawns N children
(RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment) BEFORE anything touches a GPU -- it imports
neither torch nor the pgx library -- forwards rank 0's JSON line and exits non-zero if any child fails.  Under
`python -m torch.distributed.run ... bench.py --gpus N` the ranks already exist and --gpus must equal WORLD_SIZE.
Every step runs photogrammetry_amd.dist.ShardedSequence's four phases on the GPU: frames f mod N ->
pgx_detect_batch_dev -> RCCL all-gather of {count, descriptors} -> image pairs p mod N -> pgx_match_batch_dev ->
RCCL all-gather of the match lists.  N = 1 runs the same code with the collectives elided.
--scaling weak (default): the job is N independent 64-frame sequences (64 frames and 2016 image pairs per GPU,
frames and pairs of all sequences dealt round-robin over the ranks, so both exchanges are real);
--scaling strong: one sequence whatever N.

Extra objects on the JSON line:
  roofline      the dominant kernel of the step by measured time (HIP events on the launch stream, timed steps)
  mfma          the metric's kernel (k_ham_fp4): achieved multiply-add op/s against the dense FP4 MFMA peak (10 PF)
  rooflines / kernels   every kernel group; traffic = HBM bytes from separate rocprofv3 --pmc passes (profiles/)
  match_only    pairs/s of the match stage alone (SURVEY 8d's definition of the metric)
  tracks        the track graph of every step's match lists (pgx_tracks_dev): built INSIDE the timed step, on the
                device, behind the matcher (N = 1) / behind the list gather (N > 1); counts, kernel time, and the check of
                the timed job's graph against the sequential oracle
  without_tracks the headline region once more with the track graph switched off (outside `value`): comparable with rounds 1-4
  sustained     the same steps again for >= 3 s right after the headline region (outside `value`): a rate the driver's
                clock and gpu_busy samples can corroborate
  configs       (N = 1) the other BASELINE configurations on this GPU: configs[1] x 64 independent pairs,
                configs[3]'s one-GPU window variant (1024 x 3840x2160, 0 < j - i <= 16, 8192 keypoints)
  host_api      (N = 1) pgx_detect / pgx_match from HOST buffers (what a P/Invoke caller sees), PCIe included
  cpu_baseline  (N = 1) the CPU oracle (literal single-thread port of the C#) on a bounded sample; beside it the
                optimised matcher on one core and on --cpu-procs processes, CPU model and core counts stated
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W, H = 1920, 1080
P = 256
WORDS = 8
NKP = 4096
RADIUS = 16
THRESH = 0.1
SEQ_FRAMES = 64
# The distance kernel runs on the block-scaled FP4 matrix instruction (v_mfma_scale_f32_32x32x64_f8f6f4, e2m1 operands
# +-1, exact): its dense peak is the FP6/FP4 figure of /opt/skills/guides/MI355X_MICROARCH.md (4x the 2.5 PF bf16 rate).
MFMA_FP4 = True
I8_MFMA_PEAK_OPS = 10.0e15   # name kept from round 1: operations of the 256-bit +-1 contraction per second (dense FP4 peak)
HBM_PEAK = 8.0e12
TRAFFIC_FILE = os.path.join("profiles", "r05_traffic.json")
TRACK_MAX_DIST = 64   # the gate of the track graph: the reference's matchers took one (match_keypoints.py:23 default 75 on its own
                      # descriptors; `new KeypointMatching(100)`, Program.cs:165); forced assignments of the greedy matcher on
                      # 256-bit BRIEF sit near 100-128, true correspondences of the translated frames near 0


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--frames", type=int, default=SEQ_FRAMES, help="frames per sequence (all ordered pairs are matched)")
    ap.add_argument("--cpu-sample", type=int, default=4096)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-procs", type=int, default=0,
                    help="processes of the multi-core CPU bar: 0 = one per CPU this process may run on (os.sched_getaffinity), 1 = skip")
    ap.add_argument("--no-dewarp", action="store_true")
    ap.add_argument("--no-extra-configs", action="store_true", help="skip the configs / host_api objects (N = 1)")
    ap.add_argument("--no-profile", action="store_true", help="no per-kernel HIP events in the timed region")
    ap.add_argument("--no-standalone-pass", action="store_true",
                    help="skip the untimed pass that times every kernel with the matcher stages in order (under rocprofv3 --stats: "
                         "every launch the profiler sees then ran the way the timed region runs it)")
    ap.add_argument("--no-overlap-exchange", action="store_true",
                    help="N > 1: run the all-gather of the match lists synchronously at the end of every step (default: it is issued "
                         "asynchronously and awaited one step later, double-buffered; the last one is awaited inside the timed region)")
    ap.add_argument("--in-flight", type=int, default=2,
                    help="jobs kept in flight: consecutive steps alternate between this many contexts (each with its own stream, buffers and "
                         "workspaces); with 2 (default) the detect chain of step k + 1 runs beside the distance kernel of step k "
                         "(round 4, same box: 7.43 ms per step against 8.04 with one job at a time); 1 = strictly one job at a time")
    ap.add_argument("--gate", default="none,rows",
                    help="--in-flight >= 2: DETECT,MATCH = the stages of the previous job that a job's detect chain / matcher wait for "
                         "(pgx_wait_stage; each one of none, detect, wide, rows, done).  Default none,rows (round 5): the distance kernel of "
                         "step k + 1 starts when step k's residual rows are written, i.e. beside step k's per-pair finish and track graph, "
                         "and its detect chain as soon as its stream gets to it (beside step k's distance kernel): 7.13 ms per step against "
                         "7.21 with none,done (round 4's default: the matcher of step k + 1 waits for all of step k's; three interleaved "
                         "runs each, same box); none,wide (the residual rows beside the next distance kernel too) 7.49.  none,none: the GPU "
                         "interleaves the jobs as it likes")
    ap.add_argument("--c-abi-comm", action="store_true",
                    help="N > 1: after the timed region, run the same job once more through the C ABI's own RCCL communicator "
                         "(pgx_comm_init + pgx_sequence_step_dev) and compare.  Off by default: it loads a second RCCL instance next to "
                         "torch's, has never run at N > 1 (the development box has one GPU), and a crash or hang there would lose the headline line")
    ap.add_argument("--no-tracks", action="store_true", help="do not build the track graph inside the step")
    ap.add_argument("--track-max-dist", type=int, default=TRACK_MAX_DIST)
    ap.add_argument("--sustain-s", type=float, default=3.2,
                    help="after the headline region: run the same steps for about this long and report `sustained` (0 = skip)")
    ap.add_argument("--master-port", type=int, default=0)
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------------
# launcher: runs in a process that never touches the GPU
# ---------------------------------------------------------------------------------------------------

def spawn_ranks(args, argv):
    """Start args.gpus copies of this script, one per GPU, and relay rank 0's JSON line.  Nothing in this
    process imports torch or loads libpgx: the children are created before any GPU call exists anywhere."""
    import socket
    assert "torch" not in sys.modules and "photogrammetry_amd" not in sys.modules
    port = args.master_port
    if not port:
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
    import tempfile
    procs = []
    out0 = tempfile.TemporaryFile()
    for r in range(args.gpus):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(args.gpus), "LOCAL_WORLD_SIZE": str(args.gpus),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "PGX_BENCH_CHILD": "1"})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=out0 if r == 0 else subprocess.DEVNULL))
    # wait for all ranks; one failed rank means the others may sit in a rendezvous for minutes: stop exactly the processes
    # started here (by handle, never by pattern) and report failure
    bad = []
    while True:
        rcs = [p.poll() for p in procs]
        bad = [(r, rc) for r, rc in enumerate(rcs) if rc not in (None, 0)]
        if bad or all(rc is not None for rc in rcs):
            break
        time.sleep(0.2)
    if bad:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=20)
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    out0.seek(0)
    lines = out0.read().decode(errors="replace").splitlines()
    js = [ln for ln in lines if ln.startswith("{")]
    for ln in lines:   # anything else a library wrote to rank 0's stdout goes to stderr: stdout carries the ONE JSON line
        if not ln.startswith("{"):
            log(ln)
    if bad:
        log("bench.py: ranks failed (rank, exit code): %s" % bad)
        return 1
    if not js:
        log("bench.py: rank 0 printed no JSON line")
        return 1
    print(js[-1], flush=True)
    return 0


# ---------------------------------------------------------------------------------------------------
# inputs (made on the device from seeded host-made base frames)
# ---------------------------------------------------------------------------------------------------

def base_frame(w, h, seed, cache_dir="/tmp/pgx_bench_cache"):
    import numpy as np
    from photogrammetry_amd import synth
    os.makedirs(cache_dir, exist_ok=True)
    path = os.path.join(cache_dir, "frame_%dx%d_%d.npy" % (w, h, seed))
    if os.path.exists(path):
        try:
            f0 = np.load(path)
            if f0.shape == (h, w, 4) and f0.dtype == np.uint16:
                return f0
        except (OSError, ValueError):
            pass   # a torn file from an interrupted run: regenerate
    f0 = synth.make_frame(w, h, seed=seed, n_shapes=int(20000 * (w * h) / (1920 * 1080)))
    try:
        tmp = "%s.%d.tmp.npy" % (path, os.getpid())
        np.save(tmp, f0)
        os.replace(tmp, path)   # atomic: another rank or run never sees a partial file
    except OSError:
        pass
    return f0


def roll_frames(torch, d_base, shifts, out=None):
    """Wrap-around translations of one device-resident RGBA64 base frame (one pixel = one int64: roll has no uint16)."""
    h, w = d_base.shape[:2]
    if out is None:
        out = torch.empty((len(shifts), h, w, 4), dtype=torch.uint16, device=d_base.device)
    b64, o64 = d_base.view(torch.int64), out.view(torch.int64)
    for k, (dx, dy) in enumerate(shifts):
        o64[k] = torch.roll(b64, shifts=(dy % h, dx % w), dims=(0, 1))
    return out


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


# ---------------------------------------------------------------------------------------------------
# CPU baseline (rank 0, N = 1): the oracle as the checker's clock, never in the product path
# ---------------------------------------------------------------------------------------------------

def _cpu_pair_worker(path):
    """One process of the multi-core CPU bar: the oracle's optimised single-thread pipeline on the sample pair."""
    import numpy as np
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


def _cpu_warm_worker(_):
    """Start-up of one worker process: the imports and the oracle library, nothing timed."""
    import numpy as np  # noqa: F401
    from oracle import cref
    cref.lib()
    return os.getpid()


def cpu_multicore(frames, dmap, pairs, nproc):
    """nproc processes, each running the optimised single-thread pipeline on the sample pair at the same time."""
    import concurrent.futures
    import multiprocessing
    import tempfile
    import numpy as np
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
            list(ex.map(_cpu_warm_worker, range(4 * nproc), chunksize=1))   # start every worker (imports, dlopen); not timed
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
    detect chain on the first two frames of the sequence, literal Theta(N^3) match on their first sample_n keypoints."""
    import numpy as np
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
            "sample": "frames 0 and 1 of the sequence: dewarp+gray+detect+NMS+BRIEF of both 1920x1080 frames (%.2f s) + literal "
                      "Theta(N^3) greedy match of the first %dx%d keypoints (%.2f s); C restatement of the C# "
                      "(hardware popcount, so faster than the real BigInteger loop)" % (t_detect, n1, n2, t_match),
            "detect_s_per_frame": t_detect / 2, "match_s": t_match, "match_n": [n1, n2]}


# ---------------------------------------------------------------------------------------------------
# the timed job checks itself (rank 0, after the timed region, outside it)
# ---------------------------------------------------------------------------------------------------

def verify_job(job, pair_list, host_frame0, dmap, pairs_tbl, FS):
    """Compare the job that was just timed with the CPU oracle: three image pairs (first, middle, last) -- their whole match
    lists against the oracle's sorted-edge-scan matcher on the GPU's own descriptors -- and one frame (global frame 0:
    keypoints, grey values by bit pattern, descriptors) against the oracle's detect chain on the host copy of that frame.
    The oracle is the checker here, never the thing measured."""
    import numpy as np
    from oracle import cref
    counts = job.counts()
    M = len(pair_list)
    res = {"pairs": [], "frames": [], "ok": True}
    for p in sorted({0, M // 2, M - 1}):
        a, b = pair_list[p]
        da = job.descriptors(a).cpu().numpy().view(np.uint32)[:counts[a]]
        db = job.descriptors(b).cpu().numpy().view(np.uint32)[:counts[b]]
        got = job.matches(p).cpu().numpy()[:counts[a]]
        exp = cref.match_sorted(da, db)
        ok = bool((got[:, 0] == exp["k1"]).all() and (got[:, 1] == exp["k2"]).all() and (got[:, 2] == exp["dist"]).all())
        res["pairs"].append({"pair": [int(a), int(b)], "n1": int(counts[a]), "n2": int(counts[b]), "ok": ok})
        res["ok"] &= ok
    if job.my_frames and job.my_frames[0] == 0 and host_frame0 is not None:   # rank 0 owns global frame 0 = the base frame itself
        ident = np.stack(np.meshgrid(np.arange(W), np.arange(H)), axis=2).astype(np.int32)
        g = cref.gray(cref.apply_distortion(host_frame0, dmap if dmap is not None else ident))
        raw = cref.detect(g, np.float32(THRESH))
        kept = raw[cref.nms(raw, RADIUS)][:NKP]
        edesc = cref.brief(g, np.stack([kept["x"], kept["y"]], 1), pairs_tbl)
        kp = job.kp_l[0].cpu().numpy()[:counts[0]]
        desc = job.descriptors(0).cpu().numpy().view(np.uint32)[:counts[0]]
        ok = bool(len(kept) == counts[0] and int(job.nraw_l[0].item()) == len(raw)
                  and (kp[:, 0] == kept["x"]).all() and (kp[:, 1] == kept["y"]).all() and (kp[:, 2] == kept["fast_score"]).all()
                  and kp[:, 3].view(np.float32).tobytes() == kept["value"].tobytes() and (desc == edesc).all())
        res["frames"].append({"frame": 0, "raw_hits": int(len(raw)), "keypoints": int(len(kept)), "ok": ok})
        res["ok"] &= ok
    res["what"] = ("after the timed region: match lists of the first / middle / last image pair vs the oracle's sorted-edge-scan matcher, "
                   "and frame 0's keypoints, grey values and descriptors vs the oracle's detect chain; bit-exact or the run fails")
    return res


def verify_tracks(job, pair_list, track_frames, max_dist, summary):
    """The track graph the timed job built on the device (this rank's: `track_frames`) against the oracle's vectorised
    sequential restatement on the job's own gathered lists: offsets, nodes and per-node track ids array for array.
    Parity is unpinned by construction (the reference has no track graph): this checks the parallel build."""
    import numpy as np
    from oracle import tracks_np
    counts = job.counts()
    fset = {f: i for i, f in enumerate(track_frames)}
    rows = [(p, fset[a], fset[b]) for p, (a, b) in enumerate(pair_list) if a in fset and b in fset]
    out = job.out_all.cpu().numpy()
    m = np.stack([out[job_slot(job, p)] for p, _, _ in rows]) if rows else np.zeros((0, job.nkp, 3), np.int32)
    e_off, e_nodes, e_tof, e_s = tracks_np.tracks_arrays(counts[track_frames], [(a, b) for _, a, b in rows], m, job.nkp, max_dist, 2)
    nt, nn = summary["n_tracks"], summary["n_nodes"]
    off = job.trk_offsets[:nt + 1].cpu().numpy()
    nodes = job.trk_nodes[:nn].cpu().numpy()
    tof = job.track_of.cpu().numpy()
    ok = bool(summary == e_s and off.shape == e_off.shape and (off == e_off).all() and nodes.shape == e_nodes.shape
              and (nodes == e_nodes).all() and (tof == e_tof).all())
    lens = np.diff(e_off)
    return {"n_tracks": int(nt), "n_nodes": int(nn), "mean_len": float(nn / nt) if nt else 0.0, "longest": int(summary["longest"]),
            "dropped": int(summary["dropped"]), "dropped_nodes": int(summary["dropped_nodes"]),
            "largest_dropped": int(summary["largest_dropped"]), "edges": int(summary["edges"]),
            "match_entries": int(len(rows) * job.nkp), "frames": len(track_frames), "image_pairs": len(rows),
            "max_dist": int(max_dist), "min_len": 2,
            "len_histogram": {str(k): int(v) for k, v in zip(*np.unique(np.minimum(lens, 64), return_counts=True))} if len(lens) else {},
            "semantics": "nodes (frame, keypoint); edges = match entries with dist <= max_dist; tracks = connected components with >= 2 "
                         "nodes; a component with two keypoints of one frame is dropped whole (include/pgx.h)",
            "verified": {"ok": ok, "what": "offsets, nodes and per-node track ids of the timed job's graph == oracle/tracks_np.tracks_arrays "
                                           "(numpy + scipy connected components) on the job's own lists; parity unpinned by construction "
                                           "(no track graph in the reference)"}}


def job_slot(job, p):
    from photogrammetry_amd import dist as pdist
    return pdist.slot_of(p, job.world, job.ps)


# ---------------------------------------------------------------------------------------------------
# one rank
# ---------------------------------------------------------------------------------------------------

KERNEL_GROUPS = ("dewarp_gray", "fast", "nms", "brief", "match_init", "ham_argmin", "match_select", "tail_fill", "tail_rows", "match_finish", "tracks")


def kernel_table(engs, steps):
    """Per kernel group: launches and milliseconds, summed over the given context(s)."""
    kern = {}
    for name in KERNEL_GROUPS:
        n, ms = 0, 0.0
        for e in (engs if isinstance(engs, (list, tuple)) else [engs]):
            n_e, ms_e = e.profile_get(name)
            n, ms = n + n_e, ms + ms_e
        if n:
            kern[name] = {"launches": n, "avg_ms": ms / n, "ms_per_step": ms / steps}
    return kern


def worker(args):
    import numpy as np
    import torch
    import torch.distributed as dist
    import photogrammetry_amd as pg
    from photogrammetry_amd import dist as pdist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        log("bench.py: --gpus %d but WORLD_SIZE is %d: refusing to report a run on a different number of GPUs" % (args.gpus, world))
        return 2
    # PGX_BENCH_REHEARSE=1: every rank on GPU 0 with the gloo backend -- a functional rehearsal of the N > 1 code path on a
    # one-GPU box (sharding, slot addressing, in-place all-gathers between real pgx contexts); its timing means nothing
    rehearse = os.environ.get("PGX_BENCH_REHEARSE") == "1"
    ndev = torch.cuda.device_count()
    if rehearse:
        local_rank = 0
    if local_rank >= ndev:
        log("bench.py: rank %d needs GPU %d but only %d visible: --gpus %d cannot run here" % (rank, local_rank, ndev, world))
        return 3
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    t_setup = time.time()
    nseq = world if args.scaling == "weak" else 1
    FS = args.frames
    n_frames = nseq * FS
    pair_list = [(s * FS + i, s * FS + j) for s in range(nseq) for i in range(FS) for j in range(i + 1, FS)]
    pairs = pg.make_brief_pairs(0, 50, P)
    dmap = None if args.no_dewarp else pg.build_dewarp_map(W, H, [3e-4, 1e-7, 0, 0, 0])
    # The track graph of every step, built on the device inside the step.  Weak scaling: the job is `world`
    # independent sequences and every gathered list is on every rank, so rank r builds the graph of sequence r (no rank
    # idles, no extra exchange); strong scaling: one sequence, every rank builds the same whole graph (a replica each).
    track_frames = list(range(rank * FS, (rank + 1) * FS)) if args.scaling == "weak" else list(range(n_frames))
    track_cfg = None if args.no_tracks else {"max_dist": args.track_max_dist, "min_len": 2, "frames": track_frames}
    NI = max(1, args.in_flight)
    stage_of = {"none": None, "detect": 0, "wide": 1, "rows": 2, "done": 3}   # PGX_STAGE_*
    GATES = tuple(stage_of[x] for x in args.gate.split(","))
    assert len(GATES) == 2
    engs, jobs = [], []
    for k in range(NI):
        e = pg.Engine(local_rank)
        e.set_brief_pairs(pairs)
        e.set_detect_params(THRESH, RADIUS)
        e.set_capacity(1 << 18, NKP)       # survivor limit: lists cut to the first NKP in NMS order (harness choice)
        if os.environ.get("PGX_BENCH_CHUNK"):   # developer A/B switch: image pairs per matcher workspace chunk
            e.set_match_chunk(int(os.environ["PGX_BENCH_CHUNK"]))
        e.set_dewarp_map(dmap)
        engs.append(e)
        jobs.append(pdist.ShardedSequence(e, W, H, n_frames, pair_list, NKP, WORDS, dev, stream=torch.cuda.Stream(device=dev),
                                          overlap_exchange=not args.no_overlap_exchange, tracks=track_cfg))
    eng, job = engs[0], jobs[0]
    stream = job.stream

    def run_steps(n):
        """n whole steps.  With two or more jobs in flight the halves of consecutive steps are issued interleaved -- front(s + 1)
        (detect + descriptor gather) before back(s) (match + list gather): the GPU work of a job stays in order on its own
        stream either way, but on the communicator, whose collectives run in issue order, the next step's small descriptor
        gather then stands in FRONT of this step's large list gather (N > 1: the next matcher never waits for 792 MB of lists
        to cross the links).  Exactly n fronts and n backs are issued; at N = 1 there is no collective and the order of issue
        changes nothing on the device."""
        def gates(s):
            # step s runs on job s % NI; its gates refer to the previous step's job.  Default none,rows: the distance kernel of step s
            # starts when step s - 1's residual rows are done (beside its per-pair finish), its detect chain as soon as its stream
            # gets to it (beside step s - 1's distance kernel)
            return (engs[(s - 1) % NI],) + GATES if (NI > 1 and s > 0) else None
        if NI == 1 or GATES[0] is not None:   # a detect gate names a stage of the PREVIOUS step's matcher call: keep whole steps in order
            for s in range(n):
                jobs[s % NI].step(d_frames, after=gates(s))
            return
        jobs[0].front(d_frames, gates(0))
        for s in range(n):
            if s + 1 < n:
                jobs[(s + 1) % NI].front(d_frames, gates(s + 1))
            jobs[s % NI].back(gates(s))

    # this rank's frames, made on the device: frame i of sequence s = base_s translated by (3i, i), wrap-around
    bases = {}
    with torch.cuda.stream(stream):
        d_frames = torch.empty((max(1, len(job.my_frames)), H, W, 4), dtype=torch.uint16, device=dev)
        for k, f in enumerate(job.my_frames):
            s, i = divmod(f, FS)
            if s not in bases:
                bases[s] = torch.from_numpy(base_frame(W, H, 4321 + s)).to(dev)
            roll_frames(torch, bases[s], [(3 * i, i)], out=d_frames[k:k + 1])
    host_base0 = base_frame(W, H, 4321)
    bases.clear()
    torch.cuda.synchronize()
    log("[rank %d/%d] setup %.1fs: %d of %d frames resident (%.0f MB), %d of %d image pairs, scaling %s"
        % (rank, world, time.time() - t_setup, len(job.my_frames), n_frames, d_frames.numel() * 2 / 1e6, len(job.my_pairs),
           len(pair_list), args.scaling))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    run_steps(max(NI, args.warmup))
    for j in jobs:
        j.finish()
    torch.cuda.synchronize()
    for e in engs:
        e.check_status()
    for j in jobs[1:]:   # every context in flight computes the same job: identical results
        assert torch.equal(j.out_all, job.out_all) and torch.equal(j.counts_all, job.counts_all)
    counts = job.counts()
    pairs_per_step = float(sum(int(counts[a]) * int(counts[b]) for a, b in pair_list))
    my_pairs_per_step = float(sum(int(counts[a]) * int(counts[b]) for a, b in (pair_list[p] for p in job.my_pairs)))
    nraw_l = job.nraw_l.cpu().numpy()[:len(job.my_frames)]
    log("[rank %d] survivors per frame min/mean/max %d/%.0f/%d, raw hits mean %.0f"
        % (rank, counts.min(), counts.mean(), counts.max(), nraw_l.mean() if len(nraw_l) else 0))

    for e in engs:
        e.profile_reset()
        # events on every launch cost 0.6 ms of a 10.7 ms step: the timed region brackets the metric's kernel only (the
        # roofline's live measurement); every other kernel is bracketed in the untimed stand-alone pass below
        e.profile_filter("ham_argmin")
        e.profile_enable(not args.no_profile)
        e.debug_counters()   # clears them
    barrier()
    t0 = time.perf_counter()
    run_steps(args.steps)
    for j in jobs:
        j.finish()   # the last steps' match-list exchanges belong to the timed region
    barrier()
    dt = time.perf_counter() - t0
    for j in jobs[1:]:   # every context in flight computed the same job in the timed region: identical results (verify_job checks jobs[0])
        assert torch.equal(j.out_all, job.out_all) and torch.equal(j.counts_all, job.counts_all) and torch.equal(j.desc_all, job.desc_all)
    dbg = [0] * 8
    for e in engs:
        e.profile_enable(False)
        e.profile_filter(None)
        e.check_status()
        dbg = [a + b for a, b in zip(dbg, e.debug_counters())]
    per_pair = float(max(1, args.steps * max(1, len(job.my_pairs))))
    tail_scans_per_pair, tail_props_per_pair = dbg[4] / per_pair, dbg[6] / per_pair
    kern_timed = kernel_table(engs, args.steps) if rank == 0 else {}
    track_summary = job.track_summary() if track_cfg else None

    # sustained leg: the same steps, same schedule, for about --sustain-s seconds (every rank the same count: it is derived
    # from the max-over-ranks time of the headline region); outside `value`
    sustained = None
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt_max = float(t[0].item())
    if args.sustain_s > 0:
        for e in engs:
            e.profile_reset()
            e.profile_filter("ham_argmin")
            e.profile_enable(not args.no_profile)   # the same instrumentation as the headline region
        barrier()
        t0 = time.perf_counter()
        n_sus, n_next = 0, max(NI, int(args.sustain_s / max(dt_max / args.steps, 1e-6)) + 1)
        while True:   # normally one stretch; a second one if the headline rate (a few steps, pipeline fill included) overestimated the step
            run_steps(n_next)
            for j in jobs:
                j.finish()
            barrier()
            n_sus += n_next
            ts = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
            if world > 1:
                dist.all_reduce(ts, op=dist.ReduceOp.MAX)   # every rank takes the same decision
            el = float(ts[0].item())
            if el >= args.sustain_s or n_sus > 200000:
                break
            n_next = max(NI, int((args.sustain_s - el) * 1.1 / (el / n_sus)) + 1)
        for e in engs:
            e.profile_enable(False)
            e.profile_filter(None)
            e.check_status()
            e.debug_counters()
        for j in jobs[1:]:
            assert torch.equal(j.out_all, job.out_all) and torch.equal(j.counts_all, job.counts_all)
        sustained = {"steps": n_sus, "seconds": float(ts[0].item()), "ms_per_step": float(ts[0].item()) / n_sus * 1e3,
                     "value": pairs_per_step * n_sus / float(ts[0].item()),
                     "what": "the headline job again, same schedule and instrumentation, right after the headline region; not part of `value`"}
        log("[rank %d] sustained: %d steps in %.2f s = %.3f ms per step" % (rank, n_sus, sustained["seconds"], sustained["ms_per_step"]))
    # the headline region once more WITHOUT the track graph (the same K steps, same schedule, same instrumentation): the figure that
    # compares with the rounds before the graph joined the step (round 4: 7.17 ms at the driver's 20 steps); outside `value`
    without_tracks = None
    if track_cfg and args.sustain_s > 0:
        saved = [j.trk for j in jobs]
        for j in jobs:
            j.trk = None
        for e in engs:
            e.profile_reset()
            e.profile_filter("ham_argmin")
            e.profile_enable(not args.no_profile)
        run_steps(NI)   # refill nothing, just leave the graph's buffers behind
        for j in jobs:
            j.finish()
        barrier()
        t0 = time.perf_counter()
        run_steps(args.steps)
        for j in jobs:
            j.finish()
        barrier()
        tw = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(tw, op=dist.ReduceOp.MAX)
        for e in engs:
            e.profile_enable(False)
            e.profile_filter(None)
            e.check_status()
            e.debug_counters()
        for j, t_ in zip(jobs, saved):
            j.trk = t_
        without_tracks = {"steps": args.steps, "ms_per_step": float(tw[0].item()) / args.steps * 1e3,
                          "value": pairs_per_step * args.steps / float(tw[0].item()),
                          "what": "the headline region repeated with the track graph switched off (detect + match only, as in the rounds "
                                  "before it joined the step); not part of `value`"}
        log("[rank %d] without the track graph: %.3f ms per step" % (rank, without_tracks["ms_per_step"]))
    # stand-alone kernel times: in the timed region three matcher stages of consecutive chunks share the chip, so their
    # event brackets overlap and stretch each other; a second, untimed pass runs the same steps with the stages in order
    kern_alone, k_alone = {}, max(3, min(20, args.steps))
    if not args.no_profile and not args.no_standalone_pass:
        eng.profile_reset()
        eng.profile_serialize(True)
        eng.profile_enable(True)
        for _ in range(k_alone):
            job.step(d_frames)
        job.finish()
        torch.cuda.synchronize()
        eng.profile_enable(False)
        eng.profile_serialize(False)
        eng.check_status()
        if rank == 0:
            kern_alone = kernel_table(eng, k_alone)
    log('finish (k_match_gs) per image pair: queue entries %.1f, matrix-row scans %.1f, proposals %.1f' % (dbg[3] / per_pair, tail_scans_per_pair, tail_props_per_pair))

    # N > 1: the same job once more through the C ABI's own communicator (pgx_comm_init / pgx_sequence_step_dev: what a
    # non-Python host would call) -- results must equal the torch.distributed run; informative, never `value`
    c_abi = None
    if world > 1 and not rehearse and args.c_abi_comm:
        try:
            box = [pg.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            eng.comm_init(rank, world, box[0])
            job2 = pdist.ShardedSequence(eng, W, H, n_frames, pair_list, NKP, WORDS, dev, stream=stream, comm="pgx")
            job2.step(d_frames)
            eng.check_status()
            same = bool(torch.equal(job2.out_all, job.out_all) and torch.equal(job2.counts_all, job.counts_all))
            k2 = max(3, min(20, args.steps))
            barrier()
            t0 = time.perf_counter()
            for _ in range(k2):
                job2.step(d_frames)
            barrier()
            t2 = torch.tensor([(time.perf_counter() - t0) / k2], dtype=torch.float64, device=dev)
            dist.all_reduce(t2, op=dist.ReduceOp.MAX)
            eng.check_status()
            c_abi = {"ms_per_step": float(t2[0].item()) * 1e3, "steps": k2, "equal_to_torch_distributed_run": same,
                     "value": pairs_per_step / float(t2[0].item()),
                     "what": "the same job as ONE C call per step (pgx_sequence_step_dev) on the context's own RCCL communicator"}
            del job2
            eng.comm_destroy()
        except Exception as e:   # informative leg: never lose the headline line to it
            log("[rank %d] C-ABI communicator leg failed: %r" % (rank, e))
            c_abi = {"error": repr(e)}

    rc = 0
    if rank == 0:
        t_v = time.time()
        verified = verify_job(job, pair_list, host_base0, dmap, pairs, FS)
        log("verification of the timed job took %.1fs: %s" % (time.time() - t_v, "ok" if verified["ok"] else "MISMATCH"))
        tracks_obj = None
        if track_cfg:
            t_v = time.time()
            tracks_obj = verify_tracks(job, pair_list, track_frames, args.track_max_dist, track_summary)
            log("track graph check took %.1fs: %s" % (time.time() - t_v, "ok" if tracks_obj["verified"]["ok"] else "MISMATCH"))
            verified["tracks_ok"] = tracks_obj["verified"]["ok"]
            verified["ok"] = bool(verified["ok"] and tracks_obj["verified"]["ok"])
        if not verified["ok"]:
            rc = 4
        F_l, M_l = len(job.my_frames), len(job.my_pairs)
        # per-kernel times: the timed region brackets the metric's kernel only; the other groups come from the stand-alone pass
        kern = dict(kern_alone)
        kern.update(kern_timed)
        rounds_wide, evals, evals0 = eng.match_stats()
        step_ms = dt_max / args.steps * 1e3
        match_ms = sum(kern[k]["ms_per_step"] for k in ("match_init", "ham_argmin", "match_select", "tail_fill", "tail_rows", "match_finish") if k in kern)
        detect_ms = sum(kern[k]["ms_per_step"] for k in ("dewarp_gray", "fast", "nms", "brief") if k in kern)
        npix = W * H
        n_raw_tot = float(nraw_l.sum())
        n_kept_tot = float(sum(int(counts[f]) for f in job.my_frames))

        traffic, traffic_src = {}, None
        tpath = os.path.join(ROOT, TRAFFIC_FILE)
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
            for k, v in tj.get("bytes_per_frame", {}).items():
                traffic[k] = v * F_l
            for k, v in tj.get("bytes_per_pair", {}).items():
                traffic[k] = v * M_l
            traffic_src = ("not measured in this run: scaled from %s (separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` "
                           "passes over the same workload, bytes per frame / per image pair x this step's counts)" % TRAFFIC_FILE)

        def hbm(name, byts, what):
            tt = kern[name]["ms_per_step"] * 1e-3
            return {"kernel": name, "bound": "hbm", "achieved": byts / tt / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                    "frac": byts / tt / HBM_PEAK, "traffic": traffic.get(name), "algorithmic": what}

        rooflines = {}
        if "dewarp_gray" in kern:
            per_px = 12.0 if dmap is None else 14.0
            rooflines["dewarp_gray"] = hbm("dewarp_gray", per_px * npix * F_l,
                                           "%.0f B/pixel/frame (8 gathered source + 4 grey + 8 map per 4 frames) x %d px x %d frames per launch" % (per_px, npix, F_l))
        if "fast" in kern:
            nseg = H * ((W + 63) // 64)
            rooflines["fast"] = hbm("fast", (4.5 * npix + 8.0 * nseg) * F_l,
                                    "4 B/px grey read + 0.5 B/px ballot planes + 8 B per 64-px row segment, %d frames" % F_l)
        if "nms" in kern:
            rooflines["nms"] = hbm("nms", 8.0 * n_raw_tot + 4.0 * n_kept_tot,
                                   "integer/latency-bound stage on L2-resident lists: 8 B per raw hit in + 4 B per survivor out "
                                   "(%d raw hits, %d survivors per step)" % (n_raw_tot, n_kept_tot))
        if "brief" in kern:
            rooflines["brief"] = hbm("brief", (2.0 * P * 4 + 48.0) * n_kept_tot,
                                     "2*P*4 B gathered + 48 B written per survivor (gathers are L2-resident: see profiles/ for the "
                                     "L2 request counters)")
        mfma = None
        if "ham_argmin" in kern:
            ops = evals * 2.0 * P
            tt = kern["ham_argmin"]["ms_per_step"] * 1e-3
            mfma = {"kernel": "ham_argmin", "bound": "mfma", "achieved": ops / tt / 1e12,
                    "peak": I8_MFMA_PEAK_OPS / 1e12, "unit": "TOP/s", "frac": ops / tt / I8_MFMA_PEAK_OPS,
                    "traffic": traffic.get("ham_argmin"),
                    "matrix_dtype": "fp4 e2m1 (+-1, exact), f32 accumulate, block scale 2^13" if MFMA_FP4 else "int8 (+-64), i32 accumulate",
                    "frac_of_int8_peak": ops / tt / 5.0e15,
                    "algorithmic": "2*P = 512 ops per descriptor-pair evaluation x %d evaluations per step "
                                   "(device-counted: sum over launches and image pairs of n1*n2 of the whole-chip rounds; %d planned per "
                                   "chunk of image pairs, of which the spare one exits early on this workload; the residual R x C "
                                   "distances that k_tail_rows_fp4 evaluates once more are NOT counted)"
                                   % (evals, rounds_wide)}
            if NI > 1:
                mfma["note"] = ("two jobs in flight: the next step's detect chain runs beside this kernel on the same CUs, so its launch "
                                "duration in the timed region includes that sharing (the step is shorter for it); `standalone` is the kernel "
                                "with the chip to itself")
            if "ham_argmin" in kern_alone:
                ta = kern_alone["ham_argmin"]["ms_per_step"] * 1e-3
                mfma["standalone"] = {"achieved": ops / ta / 1e12, "frac": ops / ta / I8_MFMA_PEAK_OPS,
                                      "ms_per_step": ta * 1e3,
                                      "note": "the same steps again, untimed, ONE job at a time, every kernel group bracketed by events "
                                              "(pgx_profile_serialize)"}
            rooflines["ham_argmin"] = mfma
        if detect_ms:
            byts = 24.0 * npix * F_l
            rooflines["detect_chain"] = {"kernel": "dewarp_gray+fast+nms+brief", "bound": "hbm",
                                         "achieved": byts / (detect_ms * 1e-3) / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                                         "frac": byts / (detect_ms * 1e-3) / HBM_PEAK, "traffic": None,
                                         "algorithmic": "SURVEY 8d: 24 B/pixel for the fused-minimum detect stream x %d frames" % F_l}
        dominant = max((k for k in kern if k in rooflines), key=lambda k: kern[k]["ms_per_step"], default=None)
        result = {
            "metric": "descriptor pairs matched/sec", "value": pairs_per_step * args.steps / dt_max,
            "unit": "descriptor pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": step_ms, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "u8 / bits (descriptor bits expanded to fp4 +-1 for the MFMA distance: exact integers in f32 accumulators; "
                     "integer keys); f32 grey",
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[2] (SURVEY 8d config 3) x %d: %d-frame 1920x1080 RGBA64 sequence(s), "
                                   "dewarp(%s)+gray+FAST(T=0.1)+NMS(r=%d; SURVEY 8d writes r = 20 for this config: both leave more than 4096 "
                                   "survivors per frame, the lists are cut to the same length either way)+BRIEF-256 per frame, all %d ordered "
                                   "image pairs per sequence through the greedy Hamming match, <=%d keypoints per frame (first %d in NMS order)"
                                   % (nseq, FS, "off" if dmap is None else "shipped coeffs", RADIUS, FS * (FS - 1) // 2, NKP, NKP),
                       "frames": n_frames, "image_pairs": len(pair_list), "frames_per_gpu": F_l, "image_pairs_per_gpu": M_l,
                       "descriptor_pairs_per_step": pairs_per_step,
                       "keypoints_min_mean_max": [int(counts.min()), float(counts.mean()), int(counts.max())],
                       "raw_hits_mean": float(nraw_l.mean()) if len(nraw_l) else 0.0,
                       "parallelism": "frames f mod %d, image pairs p mod %d; 2 all-gathers per step%s"
                                      % (world, world, "" if world > 1 else " (elided at N = 1)"),
                       "jobs_in_flight": NI,
                       "stage_gates": None if NI < 2 else
                       "step k+1 on the other context: detect chain waits for %s, matcher for %s of step k (pgx_wait_stage)" % tuple(args.gate.split(",")),
                       "rehearsal_on_one_gpu_with_gloo": rehearse},
            "roofline": rooflines.get(dominant),
            "verified": verified,
            "tracks": tracks_obj,
            "without_tracks": without_tracks,
            "sustained": sustained,
            "mfma": mfma,
            "c_abi_comm": c_abi,
            "traffic_note": traffic_src,
            "rooflines": rooflines,
            "kernels": kern,
            "kernels_note": "kernels: ham_argmin = HIP events over the TIMED steps (live; a job's matcher stages run in order, one chunk of "
                            "<= 2048 image pairs at a time, and with two jobs in flight the other job's detect chain runs beside this "
                            "kernel); every other group is taken from kernels_standalone (bracketing every launch costs 0.6 ms per step, so "
                            "the timed region brackets the metric's kernel only).  kernels_standalone: the same steps again, untimed, one "
                            "job at a time, every group bracketed -- their sum exceeds ms_per_step by what the two jobs overlap",
            "kernels_standalone": kern_alone,
            "detect": {"ms_per_step": detect_ms, "frames_per_s": F_l / (detect_ms * 1e-3) if detect_ms else None},
            "match_only": {"ms_per_step_sum_of_kernels": match_ms,
                           "wall_ms_per_step": max(step_ms - detect_ms, 0.0) if detect_ms else None,
                           "pairs_per_s": my_pairs_per_step / (max(step_ms - detect_ms, 1e-9) * 1e-3) if detect_ms else None,
                           "wide_rounds": rounds_wide, "evaluations_per_step": evals,
                           "finish_row_scans_per_image_pair": tail_scans_per_pair, "finish_proposals_per_image_pair": tail_props_per_pair,
                           "mfma_frac_of_peak_on_match_stage": (evals * 2.0 * P / (max(step_ms - detect_ms, 1e-9) * 1e-3)) / I8_MFMA_PEAK_OPS
                           if detect_ms else None,
                           "note": "match stage wall = step minus the detect kernels (rank 0); SURVEY 8d's match-only definition"},
        }
        if tracks_obj is not None:
            tracks_obj["ms_per_step"] = kern["tracks"]["ms_per_step"] if "tracks" in kern else None
            tracks_obj["in_timed_region"] = True
            tracks_obj["built_by"] = ("rank r builds the graph of sequence r (all lists are on every rank after the gather)" if args.scaling == "weak"
                                      else "every rank builds the whole graph of the one sequence (replicas)") if world > 1 else "the one GPU"
            tracks_obj["ms_per_step_note"] = ("kernel time of pgx_tracks_dev's launches from the untimed stand-alone pass (HIP events); the "
                                              "graph IS built inside every timed step, so ms_per_step / value include it")
        if world == 1 and not args.no_extra_configs:
            try:
                result["configs"] = extra_configs(torch, pg, pdist, np, eng, dev, stream, pairs, dmap)
                result["host_api"] = host_api_timing(torch, pg, np, eng, host_base0, dev)
            except Exception as e:   # never lose the headline line to an add-on
                log("extra configs failed: %r" % (e,))
                result["configs_error"] = repr(e)
        if not args.no_cpu_baseline and world == 1:   # rank 0 at N = 1 only
            t1 = time.time()
            ident = np.stack(np.meshgrid(np.arange(W), np.arange(H)), axis=2).astype(np.int32)
            sample = roll_frames(torch, torch.from_numpy(host_base0).to(dev), [(0, 0), (3, 1)]).cpu().numpy()
            cb = cpu_baseline(sample, dmap if dmap is not None else ident, pairs, args.cpu_sample)
            cb["cpu_model"] = cpu_model()
            cb["host_logical_cpus"] = os.cpu_count()
            cb["cpus_available_to_this_process"] = len(os.sched_getaffinity(0))
            if args.cpu_procs != 1:
                try:
                    avail = len(os.sched_getaffinity(0))
                    nproc = avail if args.cpu_procs <= 0 else min(args.cpu_procs, avail)
                    cb["optimised_multicore"] = cpu_multicore(sample, dmap if dmap is not None else ident, pairs, nproc)
                    cb["optimised_multicore"]["note"] = ("measured, not extrapolated: %d worker processes at once = every CPU this process may "
                                                         "run on (%d of the host's %d logical CPUs, %s), each through the whole sample pair"
                                                         % (nproc, avail, os.cpu_count(), cb["cpu_model"]))
                except Exception as e:   # a report nicety, never a reason to lose the bench line
                    log("multi-core CPU bar skipped: %r" % (e,))
            result["cpu_baseline"] = cb
            log("cpu baseline took %.1fs" % (time.time() - t1))
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    for e in engs:
        e.close()
    return rc


# ---------------------------------------------------------------------------------------------------
# the other BASELINE configurations and the host-buffer entry points (rank 0, N = 1)
# ---------------------------------------------------------------------------------------------------

def _timed(torch, fn, reps):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def extra_configs(torch, pg, pdist, np, eng, dev, stream, pairs_tbl, dmap):
    out = {}
    # -- configs[1] x 64: 64 independent 1920x1080 image pairs per step (detect 128 frames + match 64 pairs) --------
    B = 64
    with torch.cuda.stream(stream):
        nb = 8
        d_bases = [torch.from_numpy(base_frame(W, H, 1234 + b)).to(dev) for b in range(nb)]
        d_frames = torch.empty((2 * B, H, W, 4), dtype=torch.uint16, device=dev)
        for p in range(B):
            k = p // nb
            roll_frames(torch, d_bases[p % nb], [(-13 * k, 7 * k), (-13 * k + 37 + 5 * k, 7 * k + 11 + 3 * k)], out=d_frames[2 * p:2 * p + 2])
        del d_bases
        i32 = dict(dtype=torch.int32, device=dev)
        kp, desc = torch.zeros((2 * B, NKP, 4), **i32), torch.zeros((2 * B, NKP, WORDS), **i32)
        cnt, nraw, outp = torch.zeros(2 * B, **i32), torch.zeros(2 * B, **i32), torch.zeros((B, NKP, 3), **i32)
        pl = torch.tensor([[2 * p, 2 * p + 1] for p in range(B)], **i32)

        def step2():
            eng.detect_batch_dev(d_frames, 2 * B, W, H, kp, desc, cnt, nraw, NKP)
            eng.match_batch_dev(desc, cnt, NKP, WORDS, pl, B, outp, max_count=NKP)
        step2()
        torch.cuda.synchronize()
        eng.profile_reset()
        eng.profile_enable(True)
        reps = 20
        dt = _timed(torch, step2, reps)
        eng.profile_enable(False)
        kern = kernel_table(eng, reps + 1)
        c = cnt.cpu().numpy()
        npairs = float(sum(int(c[2 * p]) * int(c[2 * p + 1]) for p in range(B)))
        _, evals, _ = eng.match_stats()
        det_ms = sum(kern[k]["ms_per_step"] for k in ("dewarp_gray", "fast", "nms", "brief") if k in kern)
        out["config2_batch64"] = {
            "workload": "BASELINE configs[1] x 64: 64 independent 1920x1080 pairs per step = detect 128 frames + match 64 image pairs "
                        "(round 1's headline workload)",
            "pairs_per_s": npairs / dt, "ms_per_step": dt * 1e3, "detect_frames_per_s": 2 * B / (det_ms * 1e-3) if det_ms else None,
            "kernels_ms_per_step": {k: v["ms_per_step"] for k, v in kern.items()},
            "mfma_frac": (evals * 2.0 * P / (kern["ham_argmin"]["ms_per_step"] * 1e-3) / I8_MFMA_PEAK_OPS) if "ham_argmin" in kern else None}
        del d_frames, kp, desc, cnt, nraw, outp
    torch.cuda.empty_cache()
    # -- configs[3], one-GPU window variant: 1024 x 3840x2160, image pairs 0 < j - i <= 16, 8192 keypoints ----------
    W4, H4, F4, NKP4, R4, WIN = 3840, 2160, 1024, 8192, 22, 16
    e4 = pg.Engine(eng.device)
    try:
        e4.set_brief_pairs(pairs_tbl)
        e4.set_detect_params(THRESH, R4)
        e4.set_capacity(1 << 20, NKP4)
        e4.set_dewarp_coeffs(W4, H4, [3e-4, 1e-7, 0, 0, 0])   # the table is built on the device (no 66 MB upload)
        e4.set_stream(stream.cuda_stream)
        with torch.cuda.stream(stream):
            d_base = torch.from_numpy(base_frame(W4, H4, 4321)).to(dev)
            d_frames = roll_frames(torch, d_base, [(3 * i, i) for i in range(F4)])
            i32 = dict(dtype=torch.int32, device=dev)
            kp, desc = torch.zeros((F4, NKP4, 4), **i32), torch.zeros((F4, NKP4, WORDS), **i32)
            cnt, nraw = torch.zeros(F4, **i32), torch.zeros(F4, **i32)
            pl_h = [(i, j) for i in range(F4) for j in range(i + 1, min(F4, i + WIN + 1))]
            pl = torch.tensor(pl_h, **i32)
            outp = torch.zeros((len(pl_h), NKP4, 3), **i32)
            DB = 128

            def det4():
                for f0 in range(0, F4, DB):
                    n = min(DB, F4 - f0)
                    e4.detect_batch_dev(d_frames[f0:f0 + n], n, W4, H4, kp[f0:f0 + n], desc[f0:f0 + n], cnt[f0:f0 + n], nraw[f0:f0 + n], NKP4)

            def mat4():
                e4.match_batch_dev(desc, cnt, NKP4, WORDS, pl, len(pl_h), outp, max_count=NKP4)
            td = _timed(torch, det4, 1)
            e4.profile_reset()
            e4.profile_enable(True)
            tm = _timed(torch, mat4, 1)
            e4.profile_enable(False)
            kern = kernel_table(e4, 2)
            e4.check_status()
            c = cnt.cpu().numpy()
            npairs = float(sum(int(c[a]) * int(c[b]) for a, b in pl_h))
            _, evals, _ = e4.match_stats()
            # the track graph of those lists at this size (8.4 M nodes, 133 M match entries): pgx_tracks_dev on the resident buffers
            trk_of = torch.zeros((F4, NKP4), **i32)
            trk_off, trk_nodes, trk_sum = torch.zeros(F4 * NKP4 + 1, **i32), torch.zeros((F4 * NKP4, 2), **i32), torch.zeros(8, **i32)

            def trk4():
                e4.tracks_dev(outp, cnt, pl, len(pl_h), F4, NKP4, F4, TRACK_MAX_DIST, 2, trk_of, trk_off, trk_nodes, trk_sum)
            tt = _timed(torch, trk4, 2)
            e4.check_status()
            ts = trk_sum.cpu().tolist()
            tracks4 = {"s": tt, "match_entries_per_s": len(pl_h) * NKP4 / tt, "n_tracks": ts[0], "n_nodes": ts[1], "dropped": ts[2],
                       "dropped_nodes": ts[3], "edges": ts[4], "longest": ts[5], "largest_dropped": ts[6],
                       "consistent": bool(int(trk_off[ts[0]].item()) == ts[1] and int((trk_of >= 0).sum().item()) == ts[1]
                                          and int((trk_of == -2).sum().item()) == ts[3]),
                       "max_dist": TRACK_MAX_DIST}
            del trk_of, trk_off, trk_nodes
            out["config4_window16"] = {
                "workload": "BASELINE configs[3], one-GPU window variant (SURVEY 8d config 4): 1024 x 3840x2160 frames made on the device, "
                            "image pairs 0 < j - i <= 16 (%d), r = %d, <=%d keypoints per frame" % (len(pl_h), R4, NKP4),
                "detect_frames_per_s": F4 / td, "detect_s": td, "match_s": tm, "descriptor_pairs": npairs,
                "match_pairs_per_s": npairs / tm, "end_to_end_pairs_per_s": npairs / (td + tm),
                "keypoints_min_max": [int(c.min()), int(c.max())],
                "kernels_ms": {k: v["ms_per_step"] for k, v in kern.items()},
                "mfma_frac": (evals * 2.0 * P / (kern["ham_argmin"]["ms_per_step"] * 1e-3) / I8_MFMA_PEAK_OPS) if "ham_argmin" in kern else None,
                "tracks": tracks4}
            del d_frames, d_base, kp, desc, cnt, nraw, outp
    finally:
        e4.close()
    torch.cuda.empty_cache()
    # -- configs[4], one-GPU window variant: 256 x 1920x1080, dewarp -> detect -> match -> RANSAC fundamental -> pose ---------
    F5, WIN5, NS5, PPS5, THR5 = 256, 16, 2000, 32, 0.001   # samples / pairs per sample / threshold: Program.cs:229 (commented call)
    with torch.cuda.stream(stream):
        d_base = torch.from_numpy(base_frame(W, H, 4321)).to(dev)
        d_frames = roll_frames(torch, d_base, [(3 * i, i) for i in range(F5)])
        i32 = dict(dtype=torch.int32, device=dev)
        kp, desc = torch.zeros((F5, NKP, 4), **i32), torch.zeros((F5, NKP, WORDS), **i32)
        cnt, nraw = torch.zeros(F5, **i32), torch.zeros(F5, **i32)
        pl_h = [(i, j) for i in range(F5) for j in range(i + 1, min(F5, i + WIN5 + 1))]
        M5 = len(pl_h)
        pl = torch.tensor(pl_h, **i32)
        outp = torch.zeros((M5, NKP, 3), **i32)
        d_F = torch.zeros((M5, 9), dtype=torch.float32, device=dev)
        d_in, d_bs = torch.zeros(M5, **i32), torch.zeros(M5, **i32)
        d_Rt = torch.zeros((M5, 12), dtype=torch.float32, device=dev)
        d_votes, d_best = torch.zeros((M5, 4), **i32), torch.zeros(M5, **i32)

        def det5():
            for f0 in range(0, F5, 64):
                eng.detect_batch_dev(d_frames[f0:f0 + 64], 64, W, H, kp[f0:f0 + 64], desc[f0:f0 + 64], cnt[f0:f0 + 64], nraw[f0:f0 + 64], NKP)

        def mat5():
            eng.match_batch_dev(desc, cnt, NKP, WORDS, pl, M5, outp, max_count=NKP)

        def ran5():
            eng.fundamental_ransac_dev(kp, outp, cnt, pl, M5, NKP, NS5, PPS5, THR5, d_F, d_in, d_bs, seed=7)

        def pose5():
            eng.pose_dev(kp, outp, cnt, pl, M5, NKP, d_F, d_Rt, d_votes, d_best)
        td, tm, tr, tp = _timed(torch, det5, 1), _timed(torch, mat5, 1), _timed(torch, ran5, 1), _timed(torch, pose5, 1)
        eng.check_status()
        c = cnt.cpu().numpy()
        npairs = float(sum(int(c[a]) * int(c[b]) for a, b in pl_h))
        inl = d_in.cpu().numpy()
        votes = d_votes.cpu().numpy()
        out["config5_pose_window16"] = {
            "workload": "BASELINE configs[4], one-GPU window variant: 256 x 1920x1080 frames, dewarp -> detect -> match (image pairs "
                        "0 < j - i <= 16: %d) -> RANSAC fundamental matrix (%d samples x %d pairs, threshold %g: the reference's "
                        "commented call, Program.cs:229) -> essential matrix / pose with triangulation vote" % (M5, NS5, PPS5, THR5),
            "detect_s": td, "match_s": tm, "ransac_s": tr, "pose_s": tp, "frames_per_s_end_to_end": F5 / (td + tm + tr + tp),
            "descriptor_pairs": npairs, "match_pairs_per_s": npairs / tm, "ransac_samples_per_s": M5 * NS5 / tr,
            "image_pairs_with_a_model": int((inl >= 0).sum()), "inliers_median": float(np.median(inl[inl >= 0])) if (inl >= 0).any() else None,
            "winning_vote_share_median": float(np.median(votes.max(1) / np.maximum(votes.sum(1), 1))),
            "note": "pose arithmetic is parity-unpinned (unseeded RNG and MathNet SVD in the reference); timings only"}
        del d_frames, d_base, kp, desc, cnt, nraw, outp
    torch.cuda.empty_cache()
    return out


def host_api_timing(torch, pg, np, eng, base, dev):
    """pgx_detect + pgx_match on ONE 1920x1080 pair from HOST buffers: the entry points the C# call sites would bind
    (DeWarpTransformStepFactory.cs:58-60, KeyPointDetectionTransformStepFactory.cs:33-35, TestService.cs:96)."""
    from photogrammetry_amd import synth
    res = {}
    f0 = base
    f1 = synth.shift_frame(base, 37, 11)
    pinned = [torch.empty(f0.shape, dtype=torch.uint16).pin_memory() for _ in range(2)]
    pinned[0].numpy()[...] = f0
    pinned[1].numpy()[...] = f1
    for label, frames in (("pageable", (f0, f1)), ("pinned", (pinned[0].numpy(), pinned[1].numpy()))):
        def det():
            return [eng.detect(f, capacity=NKP) for f in frames]
        det()
        reps = 5
        t0 = time.perf_counter()
        for _ in range(reps):
            r = det()
        t_det = (time.perf_counter() - t0) / reps / 2
        d0, d1 = r[0][1], r[1][1]
        eng.match(d0, d1)
        t0 = time.perf_counter()
        for _ in range(reps):
            eng.match(d0, d1)
        t_match = (time.perf_counter() - t0) / reps
        res[label] = {"pgx_detect_ms_per_frame": t_det * 1e3, "pgx_match_ms_per_pair": t_match * 1e3,
                      "pairs_per_s_one_pair_at_a_time": len(d0) * len(d1) / (2 * t_det + t_match),
                      "keypoints": [len(d0), len(d1)]}
        if label == "pinned":
            # the batched host entry point: 64 image pairs (the same two sets, both directions) per pgx_match_batch call
            plb = [(0, 1), (1, 0)] * 32
            eng.match_batch([d0, d1], plb)
            t0 = time.perf_counter()
            for _ in range(reps):
                eng.match_batch([d0, d1], plb)
            t_b = (time.perf_counter() - t0) / reps
            res["pgx_match_batch"] = {"image_pairs_per_call": len(plb), "ms_per_call": t_b * 1e3, "ms_per_pair": t_b * 1e3 / len(plb),
                                      "descriptor_pairs_per_s": len(plb) * len(d0) * len(d1) / t_b,
                                      "what": "pgx_match_batch from host arrays: one upload (2 descriptor sets), one enqueue of the batched "
                                              "matcher, one download of 64 match lists"}
    frame_mb = f0.nbytes / 1e6
    res["note"] = ("synchronous host-buffer calls, one frame / one image pair at a time: each pgx_detect uploads %.1f MB (PCIe Gen5 x16 "
                   "spec 63 GB/s = %.2f ms at best) and waits; the batched _dev entry points above are the throughput path"
                   % (frame_mb, frame_mb / 63e3 * 1e3))
    return res


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    if args.gpus < 1:
        log("bench.py: --gpus must be >= 1")
        return 2
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args, argv)
    return worker(args)


if __name__ == "__main__":
    sys.exit(main())
