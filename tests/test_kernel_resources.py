"""The 128-token grouped GEMM (sgl-cpu-tests_amd/csrc/moe_gemm_fp8w_s128.hip) sits at the edge of the register file: 229-240 of 256
VGPRs at two waves per SIMD, and ROCm 7.2's allocator answers small changes of its tile function with 170-400 spilled values
 -- silently, and at twice the run time.  This test cross-compiles the file for gfx950 (no GPU needed) and fails if
any of its kernels spills, uses scratch or loses its two workgroups per CU."""
import os
import re
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_the_128_token_kernels_do_not_spill():
    src = os.path.join(ROOT, "sgl-cpu-tests_amd", "csrc", "moe_gemm_fp8w_s128.hip")
    with tempfile.TemporaryDirectory() as tmp:
        cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-I", os.path.join(ROOT, "include"),
               "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", os.path.join(tmp, "s128.o")]
        r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    blocks = re.split(r"remark: [^\n]*Function Name: ", r.stderr)[1:]
    assert len(blocks) >= 14, "expected the two-term, one-term and int8 kernels of both GEMMs"
    bad = []
    for b in blocks:
        name = b.split()[0]
        get = lambda k: int(re.search(k + r": (\d+)", b).group(1))
        vgpr, spill, scratch, occ, lds = (get("VGPRs"), get("VGPRs Spill"), get(r"ScratchSize \[bytes/lane\]"),
                                          get(r"Occupancy \[waves/SIMD\]"), get(r"LDS Size \[bytes/block\]"))
        if spill or scratch or occ < 2 or vgpr > 256 or 2 * lds > 160 * 1024:
            bad.append((name, vgpr, spill, scratch, occ, lds))
    assert not bad, bad
