"""The driver's contract on bench.py's output (one JSON line on stdout, rank 0): a short run on this GPU must carry every
field the driver and the judge read -- metric/value/unit, the step bookkeeping, `roofline` (bound, achieved, peak, unit, frac,
traffic), `cpu_baseline` (value, unit, cores, kind, sample) and the run's own verification against the oracle."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


@pytest.mark.gpu
def test_bench_json_line_contract():
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "1", "--steps", "3", "--warmup", "1", "--frames", "12", "--cpu-sample", "256",
                        "--cpu-procs", "2", "--no-extra-configs"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, "exactly one line on stdout"
    d = json.loads(lines[0])
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert base["metric"].startswith(d["metric"]) and d["unit"]   # BASELINE: "descriptor pairs matched/sec at 1/2/4/8 MI355X; % MFMA peak"
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert d["value"] > 0 and d["ms_per_step"] > 0 and isinstance(d["dtype"], str)
    assert "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and rf["unit"] in ("GB/s", "TFLOP/s", "TOP/s")
    assert rf["achieved"] > 0 and rf["peak"] > 0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    assert "traffic" in rf
    cb = d["cpu_baseline"]
    assert cb["value"] > 0 and cb["unit"] and cb["cores"] >= 1 and cb["kind"] in ("reference", "port") and cb["sample"]
    assert d["verified"]["ok"] is True and len(d["verified"]["pairs"]) == 3 and len(d["verified"]["frames"]) == 1
    # whole-job throughput: value x time per step = the descriptor pairs of one step
    per_step = d["config"]["descriptor_pairs_per_step"]
    assert per_step > 0 and abs(d["value"] * d["ms_per_step"] * 1e-3 - per_step) < 1e-6 * per_step
    # the track graph is built inside the step and checked against the sequential oracle
    tr = d["tracks"]
    assert tr["in_timed_region"] is True and tr["verified"]["ok"] is True and d["verified"]["tracks_ok"] is True
    assert tr["n_tracks"] > 100 and tr["mean_len"] >= 2.0 and tr["frames"] == 12 and tr["image_pairs"] == 66
    assert tr["ms_per_step"] > 0 and tr["edges"] > 0
    # the sustained leg: the same job for about three seconds, outside `value`
    su = d["sustained"]
    assert su["steps"] >= 2 and su["seconds"] >= 3.0 and su["ms_per_step"] > 0
    # and the headline region once more without the track graph (the figure that compares with rounds 1-4), outside `value`
    wt = d["without_tracks"]
    assert wt["steps"] == 3 and wt["ms_per_step"] > 0 and wt["value"] > 0


def _rehearse(extra):
    """bench.py --gpus 2 exactly as the driver's SCALE run launches it -- the parent spawns two ranks, each with two jobs in
    flight, interleaved front / back halves, overlapped list exchange -- except that both ranks sit on GPU 0 and the
    collectives run on gloo (PGX_BENCH_REHEARSE=1: RCCL refuses two ranks on one device)."""
    env = dict(os.environ, PGX_BENCH_REHEARSE="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "2", "--frames", "12", "--no-extra-configs",
                        "--no-cpu-baseline", "--sustain-s", "0.5"] + extra, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, "exactly one line on stdout"
    return json.loads(lines[0])


@pytest.mark.gpu
@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_default_multi_rank_path_rehearsed_on_one_gpu(scaling):
    """The code path the first real SCALE run takes (bench.py:run_steps, dist.ShardedSequence front / back / finish with
    overlap_exchange, the track graph behind the awaited list gather) at world size 2: rc 0, one JSON line, verified.
    Reference anchors: nothing couples image pairs (TestService.cs:80-96), lists are fixed size (KeypointMatching.cs:38)."""
    d = _rehearse(["--scaling", scaling])
    assert d["n_gpus"] == 2 and d["scaling"] == scaling and d["steps"] == 3
    assert d["config"]["jobs_in_flight"] == 2 and d["config"]["rehearsal_on_one_gpu_with_gloo"] is True
    assert d["verified"]["ok"] is True and d["verified"]["tracks_ok"] is True
    nseq = 2 if scaling == "weak" else 1
    assert d["config"]["frames"] == 12 * nseq and d["config"]["image_pairs"] == 66 * nseq
    assert d["config"]["frames_per_gpu"] == 6 * nseq and d["config"]["image_pairs_per_gpu"] == 33 * nseq
    tr = d["tracks"]
    assert tr["frames"] == 12 and tr["image_pairs"] == 66 and tr["n_tracks"] > 100   # rank 0's graph: its sequence / the one sequence
    assert d["sustained"]["steps"] >= 2 and d["without_tracks"]["steps"] == 3
