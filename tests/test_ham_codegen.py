"""CPU test (no GPU): the distance kernel inside libpgx.so keeps the two properties its place in the step depends on.

DESIGN.md section 11 ("vector work moved onto the matrix pipe, and held to 240 registers"):
  * k_ham_fp4 is held to 240 vector registers with NO scratch.  Two of its waves then leave 32 registers of a SIMD free, in which
    the small kernels of the neighbouring jobs run without displacing one; a build that grew to 252 registers (the allocation
    granule above 240 is 256) was 1.6 % faster alone and made the whole bench step 1.8 % SLOWER.  A compiler upgrade or a
    refactor that lets the kernel grow again shows here, on the CPU, before any GPU run.
  * the column-tile loop carries 13 MFMAs per tile (12 distance + the one that steps the C vector) and no packed float add:
    the step of the argmin key's tile field runs on the matrix pipe, not as eight v_pk_add_f32 on the vector ALU the loop is
    bound by.
Read from the code object itself (llvm-readelf --notes for the kernel descriptor's metadata, llvm-objdump -d for the loop)."""
import os
import re
import shutil
import subprocess

import pytest

import photogrammetry_amd._lib as L

LLVM = "/opt/rocm/lib/llvm/bin"


def _code_objects(tmp_path):
    if not os.path.exists(os.path.join(LLVM, "llvm-readelf")):
        pytest.skip("llvm-readelf not found")
    L.build()
    so = os.path.join(str(tmp_path), "libpgx.so")
    shutil.copy(L.LIB_PATH, so)
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", so], cwd=str(tmp_path), check=True, capture_output=True)
    return [os.path.join(str(tmp_path), f) for f in sorted(os.listdir(str(tmp_path))) if "amdgcn" in f]


def _kernel_metadata(objs, needle):
    """{field: value} of the one kernel whose name contains `needle` (AMDGPU metadata note, YAML-ish text)."""
    for o in objs:
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", o], check=True, capture_output=True, text=True).stdout
        # kernels are list items that start with "  - .agpr_count:" or "  - .args:"; split on the item marker
        for item in re.split(r"\n  - (?=\.)", notes):
            m = re.search(r"\.name:\s+(\S+)", item)
            if m and needle in m.group(1):
                out = {"name": m.group(1), "object": o}
                for key in ("vgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size", "agpr_count",
                            "group_segment_fixed_size", "max_flat_workgroup_size"):
                    mm = re.search(r"\.%s:\s+(\d+)" % key, item)
                    if mm:
                        out[key] = int(mm.group(1))
                return out
    return None


def test_distance_kernel_register_budget(tmp_path):
    objs = _code_objects(tmp_path)
    md = _kernel_metadata(objs, "k_ham_fp4ILi3E")
    assert md is not None, "k_ham_fp4<3> not found in libpgx.so"
    assert md["vgpr_count"] <= 240, md              # two waves + 32 free registers per SIMD (512 / SIMD, granule 8)
    assert md.get("agpr_count", 0) == 0, md
    assert md["vgpr_spill_count"] == 0 and md["sgpr_spill_count"] == 0 and md["private_segment_fixed_size"] == 0, md
    assert md["max_flat_workgroup_size"] == 256
    assert 2 * md["group_segment_fixed_size"] <= 160 * 1024 - 32 * 1024, md   # two workgroups per CU and room for a neighbour's LDS


def test_distance_kernel_steps_its_key_vector_on_the_matrix_pipe(tmp_path):
    """On the compiler's own listing of k_match.hip (the Makefile's flags; labels make the loop a basic block of its own)."""
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not found")
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    out = os.path.join(str(tmp_path), "k_match.s")
    subprocess.run([hipcc, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math",
                    "--cuda-device-only", "-S", os.path.join(root, "photogrammetry_amd", "csrc", "k_match.hip"), "-o", out],
                   check=True, capture_output=True)
    lines = open(out).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_ZN.*k_ham_fp4ILi3E.*:", l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    blocks, cur = [], []
    for l in lines[start:end]:
        if re.match(r"^\.LBB\d+_\d+:", l):
            blocks.append(cur)
            cur = []
        elif l.startswith("\t") and not l.startswith("\t.") and not l.strip().startswith(";"):
            cur.append(l.strip().split(";")[0].strip())
    blocks.append(cur)
    loops = [b for b in blocks if sum(1 for i in b if i.startswith("v_mfma_scale_f32_32x32x64_f8f6f4")) == 52]
    assert len(loops) == 1, "expected ONE basic block with 4 x 13 MFMAs (the three-row-tile loop, four tiles per trip): %d" % len(loops)
    loop = loops[0]
    assert not [i for i in loop if i.startswith("v_pk_add_f32") or i.startswith("v_add_f32")], "the C vector is stepped on the vector ALU again"
    valu = [i for i in loop if i.startswith("v_") and not i.startswith("v_mfma")]
    assert len(valu) <= 4 * 84, "vector instructions per four tiles: %d (327 when this was written)" % len(valu)
    assert not [i for i in loop if i.startswith("scratch_")]
