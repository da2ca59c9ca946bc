#!/usr/bin/env python3
"""Developer tool: registers / spills / occupancy of every kernel of one csrc/*.hip file (hipcc -Rpass-analysis=kernel-resource-usage).

    python tools/kernel_resources.py sgl-cpu-tests_amd/csrc/moe_gemm_fp8w_s128.hip [extra hipcc flags]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-I", os.path.join(ROOT, "include"),
       "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", os.path.join(tempfile.gettempdir(), "kres.o")] + sys.argv[2:]
txt = subprocess.run(cmd, capture_output=True, text=True).stderr
for b in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
    name = subprocess.run(["c++filt", b.split()[0]], capture_output=True, text=True).stdout.strip()
    g = lambda k: (re.search(k + r": (\d+)", b) or [None, "?"])[1]
    keys = ["VGPRs", "AGPRs", "VGPRs Spill", r"ScratchSize \[bytes/lane\]", r"Occupancy \[waves/SIMD\]", r"LDS Size \[bytes/block\]"]
    v = [g(k) for k in keys]
    print("V %3s A %3s spill %3s scratch %4s occ %s lds %6s  %s" % (*v, name[:110]))
