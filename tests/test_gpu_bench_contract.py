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
