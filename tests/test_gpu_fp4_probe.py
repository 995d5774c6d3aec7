"""The distance kernel rests on one hardware fact: v_mfma_scale_f32_32x32x64_f8f6f4 with e2m1 operands +-1 and an A-side
block scale of 2^k returns 2^k * dot + C EXACTLY (every partial sum is an integer below 2^24).  tools/probe/fp4_probe.hip
checks that against popcounts on random descriptors and on pairs covering every Hamming distance 0..248, with the scales and
the biased C the kernel uses.  Built with hipcc on the GPU box (same image); skipped where hipcc is absent."""
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fp4_mfma_is_an_exact_plus_minus_one_dot_product(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this box")
    exe = str(tmp_path / "fp4_probe")
    subprocess.run([hipcc, "-O2", "--offload-arch=gfx950", os.path.join(ROOT, "tools", "probe", "fp4_probe.hip"), "-o", exe],
                   check=True, capture_output=True, timeout=300)
    out = subprocess.run([exe], check=True, capture_output=True, text=True, timeout=60).stdout
    lines = [l for l in out.splitlines() if "mismatches" in l]
    assert len(lines) == 3, out
    for l in lines:
        assert ": 0 mismatches of 1024" in l, out
