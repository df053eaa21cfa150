"""C-ABI surface: the in-tree library loads and exports every symbol include/hfagp.h declares, and its
argument validation returns error codes (never aborts) — no compute call is made, so no GPU is needed."""
import ctypes as C
import os
import re

import pytest

from tests.util import ROOT


@pytest.fixture(scope="module")
def lib():
    from hfa_gp_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return _lib


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "hfagp.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hfagp_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree(lib):
    assert declared_symbols() == sorted(lib.SYMBOLS)


def test_every_declared_symbol_is_exported(lib):
    handle = lib.lib()
    for name in declared_symbols():
        assert hasattr(handle, name), name
    assert handle.hfagp_abi_version() == lib.ABI_VERSION


def test_argument_validation_returns_codes(lib):
    h = lib.lib()
    assert h.hfagp_raymarch_fwd(None, None) == -1
    assert b"null pointer" in h.hfagp_last_error()
    a = lib.ModconvArgs()
    assert h.hfagp_modconv_fwd(C.byref(a), None) == -1
    a.x = a.wt = a.y = 1          # non-null, never dereferenced: dims are rejected first
    a.B, a.H, a.W, a.Cin, a.Cout = 1, 4, 4, 6, 8
    assert h.hfagp_modconv_fwd(C.byref(a), None) == -2          # Cin not a multiple of 8
    assert b"multiple of 8" in h.hfagp_last_error()
    a.Cin, a.mode = 8, 7
    assert h.hfagp_modconv_fwd(C.byref(a), None) == -1          # unknown mode
    assert h.hfagp_weight_prep(1, 1, None, 8, 6, 9, None) == -2
    assert h.hfagp_upfirdn2d_fwd(None, None, None, 1, 1, 4, 4, 4, 4, 1, 1, 0, 0, 0, 0, 1.0, None) == -1
    r = lib.RaymarchArgs()
    for f in ("planes", "cam2world", "intrinsics", "u_strat", "u_imp", "dec_w0", "dec_b0", "dec_w1", "dec_b1",
              "feat", "depth", "wsum", "tminmax"):
        setattr(r, f, 1)
    r.B, r.H, r.W, r.res, r.Sc, r.Sf = 1, 64, 64, 16, 24, 24
    r.ray_start, r.ray_end, r.box_warp = 2.25, 3.3, 1.0
    assert h.hfagp_raymarch_fwd(C.byref(r), None) == -2         # unsupported sample count
    assert b"unsupported sample counts" in h.hfagp_last_error()


def test_missing_library_fails_loudly(lib, monkeypatch, tmp_path):
    monkeypatch.setattr(lib, "_lib", None)
    monkeypatch.setattr(lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        lib.lib()


def test_product_path_refuses_cpu_tensors():
    import torch
    from hfa_gp_amd.config import tiny64
    from hfa_gp_amd.generator import TriPlaneGenerator
    g = TriPlaneGenerator(tiny64())
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        g.synthesis(torch.zeros(1, g.cfg.num_ws, 512), torch.zeros(1, 25))


def test_header_is_valid_c_and_a_plain_c_caller_links(lib, tmp_path):
    """include/hfagp.h compiles as C99 (pedantic) and a plain-C program links against the in-tree library and drives
    its host-side entry points (examples/c_abi_smoke.c) — the boundary is a C ABI, not a Python extension."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("gcc not available")
    lib.lib()
    src = os.path.join(ROOT, "examples", "c_abi_smoke.c")
    exe = str(tmp_path / "c_abi_smoke")
    libdir = os.path.dirname(lib.LIB_PATH)
    cmd = [gcc, "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"), src,
           "-L", libdir, "-lhfagp_hip", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
    out = subprocess.run(cmd, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    run = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0, run.stdout + run.stderr
    assert f"abi {lib.ABI_VERSION}" in run.stdout and "null pointer" in run.stdout and "multiple of" in run.stdout
