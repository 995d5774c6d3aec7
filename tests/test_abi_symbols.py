"""CPU test: libpgx.so builds for gfx950, loads, and exports every symbol include/pgx.h declares.
No compute call is made (there is no GPU here); creating a context must fail loudly, not fall back."""
import os
import re

import pytest

import photogrammetry_amd._lib as L

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _declared():
    src = open(os.path.join(ROOT, "include", "pgx.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pgx_[a-z_0-9]+)\s*\(", src)))


def test_header_and_export_list_agree():
    assert _declared() == sorted(L.EXPORTS)


def test_library_builds_loads_and_exports_everything():
    L.build()
    lib = L.lib()
    for name in _declared():
        assert hasattr(lib, name), name
    assert lib.pgx_version().decode().startswith("pgx")


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import photogrammetry_amd as pg
    with pytest.raises(pg.PgxError):
        pg.Engine(0)


def test_host_helpers_need_no_gpu():
    import numpy as np
    import photogrammetry_amd as pg
    from oracle import cref
    assert (pg.make_brief_pairs(3, 50, 256) == cref.gaussian_pairs(3, 50, 256)).all()
    m = pg.build_dewarp_map(97, 61, [3e-4, 1e-7, 0, 0, 0])
    assert (m == cref.build_distortion_matrix(97, 61, [3e-4, 1e-7, 0, 0, 0])).all()
    with pytest.raises(pg.ArgumentException):
        pg.build_dewarp_map(8, 8, [0.0] * 4)


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under photogrammetry_amd/ may reference it."""
    pkg = os.path.join(ROOT, "photogrammetry_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h", ".hpp", ".inc")) or f == "Makefile":
                text = open(os.path.join(dp, f), errors="ignore").read()
                assert "liboracle" not in text and "pgx_oracle" not in text and "from oracle" not in text \
                    and "import oracle" not in text, os.path.join(dp, f)


def test_missing_rccl_is_an_error_code_not_a_crash():
    """pgx_comm.hip loads librccl at run time; when the file is not there the comm entry points must answer PGX_E_RCCL
    (round 2's loader called dlerror() twice and dereferenced the NULL of the second call).  PGX_RCCL_LIB points the
    loader at one file; a child process, because the loader caches its result for the life of the process."""
    import subprocess
    import sys
    code = ("import ctypes, sys\n"
            "L = ctypes.CDLL(%r)\n"
            "buf = ctypes.create_string_buffer(128)\n"
            "rc1 = L.pgx_comm_unique_id(buf)\n"
            "rc2 = L.pgx_comm_unique_id(buf)\n"
            "print(rc1, rc2)\n" % L.LIB_PATH)
    env = dict(os.environ, PGX_RCCL_LIB="/nonexistent/librccl.so.1")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.split() == [str(L.PGX_E_RCCL)] * 2
