"""Host-side sanitizer run (the reference has none; "compile ... with -fsanitize=address host side in
CI-on-CPU").  tools/sanitize/run.sh rebuilds the library's HOST code with AddressSanitizer + UndefinedBehaviorSanitizer
(device code untouched) and runs tools/sanitize/host_driver.c: every planning / sizing / validation path that needs no GPU,
over a sweep of shapes, modes and precisions, and the null-argument path of every entry point.  ~40 s on 8 cores."""
import os
import shutil
import subprocess

import pytest

from tests.util import ROOT

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
CLANG = os.environ.get("CLANG", "/opt/rocm/lib/llvm/bin/clang")


@pytest.mark.skipif(not (os.path.exists(HIPCC) and os.path.exists(CLANG)), reason="hipcc / clang not available")
def test_host_code_is_clean_under_asan_and_ubsan(tmp_path):
    out = subprocess.run(["bash", os.path.join(ROOT, "tools", "sanitize", "run.sh"), str(tmp_path)], capture_output=True,
                         text=True, timeout=1200)
    tail = (out.stdout + out.stderr)[-3000:]
    assert out.returncode == 0, tail
    assert "host sanitizer driver OK" in out.stdout, tail
    assert "ERROR: AddressSanitizer" not in tail and "runtime error:" not in tail, tail
    # the sanitizer runtime really was in the process: the instrumented library carries its symbols
    nm = shutil.which("nm")
    if nm:
        syms = subprocess.run([nm, "-D", str(tmp_path / "libhfagp_asan.so")], capture_output=True, text=True).stdout
        assert "__asan_init" in syms or "__asan_" in syms
