"""bench.py --gpus N must start N ranks itself, before anything touches a GPU, and must never report a run on a
different number of GPUs than it was asked for.  No GPU here: the children fail loudly, which
is exactly what is checked."""
import os
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _run(code, env=None, timeout=300):
    e = dict(os.environ)
    e.pop("WORLD_SIZE", None)
    e.pop("RANK", None)
    if env:
        e.update(env)
    return subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=e, capture_output=True, text=True, timeout=timeout)


def test_launcher_spawns_ranks_before_any_gpu_import_and_fails_loudly():
    code = ("import sys, bench\n"
            "a = bench.parse_args(['--gpus', '2', '--steps', '1'])\n"
            "rc = bench.spawn_ranks(a, ['--gpus', '2', '--steps', '1'])\n"
            "assert 'torch' not in sys.modules and 'photogrammetry_amd' not in sys.modules and 'ctypes' not in sys.modules\n"
            "print('RC', rc)\n")
    r = _run(code, env={"HIP_VISIBLE_DEVICES": "", "ROCR_VISIBLE_DEVICES": ""})
    assert r.returncode == 0, r.stderr
    assert "RC 1" in r.stdout                       # both children failed -> the launcher reports failure
    # each child named its rank and the missing GPU; no JSON line was printed
    assert "rank 0 needs GPU 0" in r.stderr and "rank 1 needs GPU 1" in r.stderr
    assert "{" not in r.stdout.replace("RC 1", "")


def test_gpus_must_equal_world_size():
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "1"], cwd=ROOT, capture_output=True, text=True,
                       timeout=300, env=dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0",
                                             HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES=""))
    assert r.returncode == 2
    assert "refusing to report" in r.stderr and "{" not in r.stdout


def test_main_routes_gpus_gt_1_to_the_launcher():
    code = ("import sys, bench\n"
            "called = []\n"
            "bench.spawn_ranks = lambda a, argv: called.append((a.gpus, argv)) or 0\n"
            "bench.worker = lambda a: called.append('worker') or 0\n"
            "sys.argv = ['bench.py', '--gpus', '4']\n"
            "assert bench.main() == 0 and called == [(4, ['--gpus', '4'])], called\n"
            "sys.argv = ['bench.py']\n"
            "assert bench.main() == 0 and called[-1] == 'worker'\n"
            "assert 'torch' not in sys.modules\n")
    r = _run(code)
    assert r.returncode == 0, r.stderr
