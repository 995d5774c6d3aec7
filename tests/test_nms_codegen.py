"""CPU test (no GPU): the generated code of the shipped NMS round kernel keeps the shape DESIGN.md section 4 relies on.

Background ("the predicated-load hazard"): a form of the mask round that read its neighbours' MUTABLE state (alive words,
champion entries: agent-scope loads, `sc1` in the ISA) from global memory under per-lane predicates, with divergent selects
between the loads and their use, produces a wrong survivor in 10^4 on gfx950 when other workgroups' acceptance atomics are in
flight; the cause is not found (tests/nmsexp/: reproducer, the failing and the two exact builds' listings, their diffs).  The
shipped k_nmsm_round avoids the shape by construction: mutable state is read ONCE, in the staging phase, straight into LDS;
after the workgroup barrier the evaluation works on the LDS snapshot with branch-free level arithmetic, and the only global
loads left are of the read-only disk table.

What this test pins, on the code object inside libpgx.so itself (llvm-objdump -d):
  * k_nmsm_round<2> and <3> have a workgroup barrier, at least one `sc1` load before it, and NO `sc1` load after it;
  * the committed listing of the failing form is flagged by the same check (the checker sees what it is meant to see).
A compiler upgrade or a refactor that brings mutable-state loads back into the evaluation step turns this red on the CPU,
before any GPU run.  photogrammetry_amd/csrc/Makefile additionally refuses a hipcc other than the one this was verified with."""
import os
import re
import shutil
import subprocess

import pytest

import photogrammetry_amd._lib as L

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
SC1_LOAD = re.compile(r"\b(global|flat)_load_\w+\b.*\bsc1\b")


def mutable_loads_confined_to_staging(lines):
    """(ok, reason): every sc1 load precedes the first s_barrier, and there is at least one of each."""
    ins = [ln for ln in lines if ln.strip() and not ln.strip().endswith(":")]
    bar = [k for k, ln in enumerate(ins) if re.search(r"\bs_barrier\b", ln)]
    sc1 = [k for k, ln in enumerate(ins) if SC1_LOAD.search(ln)]
    if not sc1:
        return False, "no sc1 load at all: the marker of mutable-state loads is gone, re-derive the check"
    if not bar:
        return False, "no workgroup barrier: mutable state is not staged through LDS"
    late = [k for k in sc1 if k > bar[0]]
    if late:
        return False, "%d sc1 load(s) after the first barrier, first at instruction %d: %s" % (len(late), late[0], ins[late[0]].strip())
    return True, "%d sc1 loads, all before the first of %d barriers" % (len(sc1), len(bar))


def _device_disassembly(tmp_path):
    if not os.path.exists(OBJDUMP):
        pytest.skip("llvm-objdump not found")
    L.build()
    so = os.path.join(str(tmp_path), "libpgx.so")
    shutil.copy(L.LIB_PATH, so)
    subprocess.run([OBJDUMP, "--offloading", so], cwd=str(tmp_path), check=True, capture_output=True)   # unbundles next to the copy
    text = []
    for f in sorted(os.listdir(str(tmp_path))):
        if "amdgcn" in f:
            text.append(subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", os.path.join(str(tmp_path), f)], check=True,
                                       capture_output=True, text=True).stdout)
    return "\n".join(text)


def _function(dis, needle):
    """Instruction lines of the one function whose symbol contains `needle`."""
    out, on = [], False
    for ln in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", ln)
        if m:
            if on:
                break
            on = needle in m.group(1)
            continue
        if on:
            out.append(re.sub(r"//.*$", "", ln))
    return out


@pytest.mark.parametrize("rr", [2, 3])
def test_shipped_round_reads_mutable_state_only_while_staging(tmp_path, rr):
    dis = _device_disassembly(tmp_path)
    fn = _function(dis, "k_nmsm_roundILi%dE" % rr)
    assert len(fn) > 500, "k_nmsm_round<%d> not found in libpgx.so" % rr
    ok, why = mutable_loads_confined_to_staging(fn)
    assert ok, why
    # the acceptance atomics sit behind the barriers as well
    ins = [ln for ln in fn if ln.strip()]
    first_bar = next(k for k, ln in enumerate(ins) if "s_barrier" in ln)
    atom = [k for k, ln in enumerate(ins) if re.search(r"\bglobal_atomic_", ln)]
    assert atom and min(atom) > first_bar


def test_checker_flags_the_failing_form():
    path = os.path.join(ROOT, "tests", "nmsexp", "disasm", "round_s3_fail.s")
    ok, why = mutable_loads_confined_to_staging(open(path).read().splitlines())
    assert not ok, why
    # and the two builds of the same source that happen to be exact have the same shape: the check is about the source
    # form, not about the 40 instructions the peephole switch moves (those are in the committed .shape.diff)
    for v in ("nopeephole", "nohoist"):
        ok, _ = mutable_loads_confined_to_staging(open(os.path.join(ROOT, "tests", "nmsexp", "disasm", "round_s3_%s.s" % v)).read().splitlines())
        assert not ok
    d = open(os.path.join(ROOT, "tests", "nmsexp", "disasm", "round_s3_fail_vs_nopeephole.shape.diff")).read().splitlines()
    changed = [ln for ln in d if ln[:1] in "+-" and not ln.startswith(("+++", "---"))]
    assert len(changed) == 40
