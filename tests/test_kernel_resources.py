"""Build-time guard: the MFMA conv kernels must compile without register spills for gfx950 (hipcc cross-compiles
without a GPU).  A refactoring that looked neutral once moved the 9-tap split-bf16 kernel from 190 to 256 VGPRs +
spills (+27 % run time); this catches that on the CPU suite instead of on the GPU box."""
import os
import re
import shutil
import subprocess

import pytest

from tests.util import ROOT

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.mark.skipif(not (os.path.exists(HIPCC) or shutil.which("hipcc")), reason="hipcc not available")
@pytest.mark.parametrize("src,patterns", [
    ("modconv_bf16.hip", [r"modconv_bf16_kernelILi[124]E", r"upconv_bf16_kernelILi[124]E"]),
    ("modconv.hip", [r"modconv_kernelI"]),
    ("wgrad_bf16.hip", [r"wgrad_bf16_kernelI"]),
    ("wgrad.hip", [r"wgrad_kernelI"]),
])
def test_conv_kernels_do_not_spill(tmp_path, src, patterns):
    out = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c",
                          os.path.join(ROOT, "hfa-gp_amd", "csrc", src), "-o", str(tmp_path / "x.o"),
                          "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    name, seen = None, 0
    for line in out.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
        m = re.search(r"VGPRs Spill: (\d+)", line)
        if m and name and any(re.search(p, name) for p in patterns):
            seen += 1
            assert int(m.group(1)) == 0, f"{name} spills {m.group(1)} VGPRs"
    assert seen >= len(patterns), "resource remarks not found"
