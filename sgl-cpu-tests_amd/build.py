#!/usr/bin/env python3
"""Builds sgl_kernel/libsglk.so (the C-ABI HIP library, include/sglk.h) with hipcc for gfx950.

    python sgl-cpu-tests_amd/build.py [--force] [--verbose]

The .so is built IN-TREE next to the Python package so it travels with the repo snapshot to the GPU box.
hipcc cross-compiles without a GPU.  One translation unit per csrc/*.hip, compiled in parallel, then linked.
"""
import concurrent.futures
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "sgl_kernel", "libsglk.so")
OBJ = os.path.join(HERE, "build", "obj")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-fno-gpu-rdc", "-I", os.path.join(HERE, "..", "include")]
if os.environ.get("SGLK_EXTRA_FLAGS"):     # developer experiments, e.g. -DSGLK_DMA_SLOT_A=5
    FLAGS += os.environ["SGLK_EXTRA_FLAGS"].split()
_suffix = ""
if os.environ.get("SGLK_DEV_ABLATE"):      # developer-only build (in-kernel time stamps): its own library, loaded with
    FLAGS.append("-DSGLK_DEV_ABLATE")      # SGLK_LIB_PATH=.../libsglk_dev.so; the product library is untouched
    _suffix += "_dev"
if os.environ.get("SGLK_BUILD_TAG"):        # developer A/B builds (with SGLK_EXTRA_FLAGS): libsglk[_dev]_<tag>.so
    _suffix += "_" + os.environ["SGLK_BUILD_TAG"]
if _suffix:
    OUT = os.path.join(HERE, "sgl_kernel", "libsglk%s.so" % _suffix)
    OBJ = os.path.join(HERE, "build", "obj" + _suffix)


def _newer(a, b):
    return not os.path.exists(b) or os.path.getmtime(a) > os.path.getmtime(b)


def build_all(force=False, verbose=False):
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    hdrs = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(HERE, "..", "include", "*.h"))
    os.makedirs(OBJ, exist_ok=True)
    stamp = os.path.join(OBJ, "flags.txt")       # a change of compile flags (e.g. SGLK_DEV_ABLATE) rebuilds everything
    flags_now = " ".join(FLAGS)
    if not os.path.exists(stamp) or open(stamp).read() != flags_now:
        force = True
        with open(stamp, "w") as f:
            f.write(flags_now)
    newest_hdr = max(os.path.getmtime(h) for h in hdrs)
    jobs = []
    for s in srcs:
        o = os.path.join(OBJ, os.path.basename(s)[:-4] + ".o")
        if force or _newer(s, o) or newest_hdr > os.path.getmtime(o):
            jobs.append((s, o))

    def cc(job):
        s, o = job
        cmd = [HIPCC] + FLAGS + ["-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {s}:\n{r.stdout}\n{r.stderr}")
        if r.stderr.strip() and verbose:
            print(r.stderr)
        return o

    with concurrent.futures.ThreadPoolExecutor(max_workers=min(6, max(1, len(jobs)))) as ex:
        list(ex.map(cc, jobs))
    objs = [os.path.join(OBJ, os.path.basename(s)[:-4] + ".o") for s in srcs]
    if force or jobs or not os.path.exists(OUT):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    out = build_all(force="--force" in sys.argv, verbose="--verbose" in sys.argv or "-v" in sys.argv)
    print(out)
