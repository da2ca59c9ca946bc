"""Same-process A/B of fp8 fused_experts under different developer knobs (csrc/knobs.h), bench.py's data and rotation:
interleaved rounds, per-stage HIP-event times (align / GEMM-1 / GEMM-2 / combine), median and minimum per variant.

    python tools/ab_knobs.py [--tokens 16384,4096] [--rounds 5] [--iters 12] "name:K1=V1,K2=V2" "base:" ...

A variant is `name:` followed by SGLK_* assignments (without the prefix); an empty list = the shipped defaults.  The library
re-reads its environment between variants (sglk_reload_env), so all arms share one process, one device and one clock history
(cdna_hip_programming.md rule 24)."""
import argparse
import ctypes
import json
import os
import statistics
import sys

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "sgl-cpu-tests_amd"))
sys.path.insert(0, ROOT)
import sgl_kernel  # noqa: E402,F401
from sgl_kernel import _lib, _ops  # noqa: E402
import bench  # noqa: E402

ops = torch.ops.sgl_kernel


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tokens", default="16384")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=12)
    ap.add_argument("--fp8-act", action="store_true", help="opt-in a8 mode for every variant")
    ap.add_argument("--int8", action="store_true", help="the int8 W8A8 operator (bench_moe.py:89-106) instead of fp8 W8A16")
    ap.add_argument("variants", nargs="+")
    args = ap.parse_args()
    variants = []
    for v in args.variants:
        name, _, rest = v.partition(":")
        kv = dict(x.split("=", 1) for x in rest.split(",") if x)
        variants.append((name, {"SGLK_" + k: val for k, val in kv.items()}))
    all_keys = sorted({k for _, kv in variants for k in kv})
    L = _lib.lib()
    dev = torch.device("cuda", 0)
    timer = L.sglk_stage_timer_create(args.iters + 8)
    if args.fp8_act:
        _ops.set_fp8_activations(True)
    for M in [int(x) for x in args.tokens.split(",")]:
        a, w1, w2, w1s, w2s, tw, ids = bench.make_inputs(M, bench.N_EXPERTS, dev, 1111)
        if args.int8:
            g = torch.Generator(device=dev).manual_seed(77)
            E, N2, K = w1.shape
            w1 = torch.randint(-127, 128, w1.shape, generator=g, dtype=torch.int8, device=dev)
            w2 = torch.randint(-127, 128, w2.shape, generator=g, dtype=torch.int8, device=dev)
            w1s = torch.rand(E, N2, generator=g, device=dev) * 1e-2
            w2s = torch.rand(E, K, generator=g, device=dev) * 1e-2
        w1p = [ops.convert_weight_packed(w1)]
        w2p = [ops.convert_weight_packed(w2)]
        del w1, w2
        w1p.append(w1p[0].clone())
        w2p.append(w2p[0].clone())
        inputs = [a.clone() for _ in range(args.iters)]

        def call(i):
            if args.int8:
                return ops.fused_experts_cpu(inputs[i % len(inputs)], w1p[i & 1], w2p[i & 1], tw, ids, False, True, False, w1s, w2s,
                                             None, None, None, True)
            return ops.fused_experts_cpu(inputs[i % len(inputs)], w1p[i & 1], w2p[i & 1], tw, ids, False, False, True, w1s, w2s,
                                         bench.BLOCK, None, None, True)

        def set_env(kv):
            for k in all_keys:
                os.environ.pop(k, None)
            os.environ.update(kv)
            L.sglk_reload_env()

        res = {name: [] for name, _ in variants}
        outs = {}
        for name, kv in variants:      # warm every arm (and the clock) first, keep one output per arm for a cross-check
            set_env(kv)
            for i in range(4):
                o = call(i)
            torch.cuda.synchronize()
            outs[name] = (o.float().clone(), _ops.last_path)
        for _ in range(args.rounds):
            for name, kv in variants:
                set_env(kv)
                call(0)
                torch.cuda.synchronize()
                _ops.set_stage_timer(timer)
                L.sglk_stage_timer_reset(timer)
                for i in range(args.iters):
                    call(i)
                torch.cuda.synchronize()
                _ops.set_stage_timer(None)
                ms = (ctypes.c_float * _lib.NUM_STAGES)()
                calls = ctypes.c_int32(0)
                _lib.check(L.sglk_stage_timer_read(timer, ms, ctypes.byref(calls)), "stage_timer_read")
                res[name].append([float(ms[i]) for i in range(_lib.NUM_STAGES)])
        base = outs[variants[0][0]][0]
        for name, kv in variants:
            rs = res[name]
            med = [statistics.median(r[i] for r in rs) for i in range(_lib.NUM_STAGES)]
            tot = [sum(r) for r in rs]
            rel = float((outs[name][0] - base).norm() / base.norm().clamp_min(1e-20))
            print(json.dumps({"M": M, "variant": name, "knobs": kv, "path": hex(outs[name][1]),
                              "stage_ms_median": {n: round(med[i], 4) for i, n in enumerate(_lib.STAGE_NAMES)},
                              "sum_ms_median": round(statistics.median(tot), 4), "sum_ms_min": round(min(tot), 4),
                              "tflops_median": round(M * bench.FLOP_PER_TOKEN / statistics.median(tot) / 1e9, 1),
                              "gemm1_pf_median": round(M * bench.GEMM1_FLOP_PER_TOKEN / med[1] / 1e12, 4) if med[1] > 0 else None,
                              "rel_diff_vs_first": round(rel, 6)}), flush=True)
        set_env({})
        del w1p, w2p, inputs
        torch.cuda.empty_cache()
    L.sglk_stage_timer_destroy(timer)


if __name__ == "__main__":
    main()
