"""GPU parity of the fp8 (W8A16) fused_experts hot path, through torch.ops.sgl_kernel -> C-ABI -> HIP.

Mirrors /root/reference/test_moe_fp8_ext.py:94-124 and /root/reference/test_moe_offloading_cpu.py:54-145:
same call signature, same pass predicate (utils.compare: allclose rtol=atol=1e-2 on bf16), plus a tighter
bound and bit-exact structural properties.  Expected outputs come from tests/golden (reference's own oracle)
and, for shapes too large to store, from oracle/ (pinned to the goldens by tests/test_oracle_golden.py).
"""
import numpy as np
import pytest
import torch

import recipes
from conftest import load_golden
from oracle import c_oracle, moe

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    import sgl_kernel  # noqa: F401
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.ops.sgl_kernel


def run_fp8(ops, inp, block, inplace=False, dev="cuda"):
    d = {k: v.to(dev) for k, v in inp.items()}
    w1p = ops.convert_weight_packed(d["w1"])
    w2p = ops.convert_weight_packed(d["w2"])
    out = ops.fused_experts_cpu(d["a"], w1p, w2p, d["topk_weight"], d["topk_ids"], inplace, False, True,
                                d["w1s"], d["w2s"], list(block), None, None, True)
    torch.cuda.synchronize()
    return out, d


def check_close(out_bf16, ref_f32, what):
    out = out_bf16.float().cpu()
    # 1. the reference's own predicate (/root/reference/utils.py:9-13), reference value second as in test_moe_fp8_ext.py:120
    assert torch.allclose(ref_f32.bfloat16(), out_bf16.cpu(), rtol=1e-2, atol=1e-2), f"{what}: reference predicate failed"
    # 2. stated tolerance of this implementation: 3 bf16 roundings (ic1, ic2, out) -> |err| <= 2e-3 + 1.6% * |ref| is
    #    loose; measured errors are far smaller, so also bound the relative RMS error
    err = (out - ref_f32).norm() / ref_f32.norm().clamp_min(1e-12)
    assert err < 6e-3, f"{what}: relative RMS error {err:.2e}"


@pytest.mark.parametrize("case", recipes.MOE_FP8_CASES, ids=lambda c: c[0])
def test_fused_experts_fp8_golden(ops, case):
    name, M, N, K, E, topk, bn, bk, masked, seed, full = case
    g, meta = load_golden("moe_fp8_" + name)
    inp = recipes.moe_fp8_inputs(M, N, K, E, topk, bn, bk, masked, seed)
    out, _ = run_fp8(ops, inp, (bn, bk))
    assert out.dtype == torch.bfloat16 and tuple(out.shape) == (int(meta["M"]), K)
    check_close(out, g["ref_out_f32"], name)


def test_fused_experts_fp8_inplace_and_aliasing(ops):
    name, M, N, K, E, topk, bn, bk, masked, seed, full = recipes.MOE_FP8_CASES[1]
    inp = recipes.moe_fp8_inputs(M, N, K, E, topk, bn, bk, masked, seed)
    out, d = run_fp8(ops, inp, (bn, bk), inplace=False)
    assert out.data_ptr() != d["a"].data_ptr()
    assert torch.equal(d["a"].cpu(), inp["a"]), "inplace=False must not touch hidden_states"
    out2, d2 = run_fp8(ops, inp, (bn, bk), inplace=True)
    assert out2.data_ptr() == d2["a"].data_ptr(), "inplace=True returns hidden_states' storage (bench_moe.py:65)"
    assert torch.equal(out2, out), "inplace and out-of-place results must be bit-identical"


def test_fused_experts_fp8_run_to_run_bit_identical(ops):
    name, M, N, K, E, topk, bn, bk, masked, seed, full = recipes.MOE_FP8_CASES[2]
    inp = recipes.moe_fp8_inputs(M, N, K, E, topk, bn, bk, masked, seed)
    a, _ = run_fp8(ops, inp, (bn, bk))
    b, _ = run_fp8(ops, inp, (bn, bk))
    assert torch.equal(a, b)


def test_fused_experts_fp8_properties(ops):
    """Size-independent properties, bit-exact: token permutation, x2 routing weights, -1 id == zero weight."""
    M, N, K, E, topk, bn, bk = 333, 256, 512, 16, 4, 128, 128
    inp = recipes.moe_fp8_inputs(M, N, K, E, topk, bn, bk, False, 4242)
    base, _ = run_fp8(ops, inp, (bn, bk))
    # permuting tokens permutes outputs exactly (each row is an independent dot-product chain)
    perm = torch.randperm(M, generator=torch.Generator().manual_seed(1))
    inp_p = dict(inp, a=inp["a"][perm], topk_weight=inp["topk_weight"][perm], topk_ids=inp["topk_ids"][perm])
    out_p, _ = run_fp8(ops, inp_p, (bn, bk))
    assert torch.equal(out_p.cpu(), base.cpu()[perm])
    # doubling every routing weight doubles the output exactly (power-of-two scaling commutes with rounding)
    out2, _ = run_fp8(ops, dict(inp, topk_weight=inp["topk_weight"] * 2), (bn, bk))
    assert torch.equal(out2.float().cpu(), base.float().cpu() * 2)
    # a slot masked with -1 == the same slot with routing weight 0 (up to the sign of zero)
    ids = inp["topk_ids"].clone()
    w = inp["topk_weight"].clone()
    ids[::3, 1] = -1
    w0 = w.clone()
    w0[::3, 1] = 0.0
    out_m, _ = run_fp8(ops, dict(inp, topk_ids=ids), (bn, bk))
    out_z, _ = run_fp8(ops, dict(inp, topk_weight=w0), (bn, bk))
    assert torch.equal(out_m.float().cpu(), out_z.float().cpu())


def test_fused_experts_fp8_all_masked_and_empty(ops):
    M, N, K, E, topk, bn, bk = 5, 128, 128, 8, 2, 128, 128
    inp = recipes.moe_fp8_inputs(M, N, K, E, topk, bn, bk, False, 7)
    ids = torch.full_like(inp["topk_ids"], -1)
    out, _ = run_fp8(ops, dict(inp, topk_ids=ids), (bn, bk))
    assert torch.count_nonzero(out.float()) == 0
    empty = {k: (v[:0] if k in ("a", "topk_weight", "topk_ids") else v) for k, v in inp.items()}
    out0, _ = run_fp8(ops, empty, (bn, bk))
    assert tuple(out0.shape) == (0, K)


@pytest.mark.parametrize("M", [1, 64, 300, 1000])
def test_fused_experts_fp8_qwen3_full_experts(ops, M):
    """Qwen3-30B-A3B expert dims with all 128 experts (604 MB of fp8 weights) — BASELINE.json config 2.

    Weights are generated on the GPU (too large for a fixture); the expected values come from the plain-C
    oracle run on a sample of tokens with the very same weights."""
    N, K, E, topk, bn, bk = 768, 2048, 128, 8, 128, 128
    g = torch.Generator(device="cuda").manual_seed(1234)
    w1 = (torch.randn(E, 2 * N, K, device="cuda", generator=g) * 400).clamp(-400, 400).to(torch.float8_e4m3fn)
    w2 = (torch.randn(E, K, N, device="cuda", generator=g) * 400).clamp(-400, 400).to(torch.float8_e4m3fn)
    w1s = torch.randn(E, 2 * N // bn, K // bk, device="cuda", generator=g) * 1e-3
    w2s = torch.randn(E, K // bn, N // bk, device="cuda", generator=g) * 1e-3
    a = (torch.randn(M, K, device="cuda", generator=g) / K ** 0.5).bfloat16()
    score = torch.softmax(torch.randn(M, E, device="cuda", generator=g).bfloat16(), dim=-1, dtype=torch.float32)
    tw, ids = torch.topk(score, topk)
    ids = ids.to(torch.int32)
    w1p, w2p = ops.convert_weight_packed(w1), ops.convert_weight_packed(w2)
    out = ops.fused_experts_cpu(a, w1p, w2p, tw, ids, False, False, True, w1s, w2s, [bn, bk], None, None, True)
    torch.cuda.synchronize()
    sample = torch.arange(0, M, max(1, M // 24))[:24]
    ref = c_oracle.fused_experts_fp8(a[sample].cpu(), w1.cpu(), w2.cpu(), w1s.cpu(), w2s.cpu(), (bn, bk),
                                     tw[sample].cpu(), ids[sample].cpu())
    check_close(out[sample], ref, f"qwen3 M={M}")


def test_cpu_tensors_are_staged_through_the_gpu(ops):
    """The reference scripts pass CPU tensors (/root/reference/test_moe_fp8_ext.py:96-118): same call, host in/out."""
    name, M, N, K, E, topk, bn, bk, masked, seed, full = recipes.MOE_FP8_CASES[0]
    g, _ = load_golden("moe_fp8_" + name)
    inp = recipes.moe_fp8_inputs(M, N, K, E, topk, bn, bk, masked, seed)
    w1p = ops.convert_weight_packed(inp["w1"])
    w2p = ops.convert_weight_packed(inp["w2"])
    assert w1p.device.type == "cpu" and w1p.shape == inp["w1"].shape and w1p.dtype == inp["w1"].dtype
    a = inp["a"].clone()
    out = ops.fused_experts_cpu(a, w1p, w2p, inp["topk_weight"], inp["topk_ids"], True, False, True,
                                inp["w1s"], inp["w2s"], [bn, bk], None, None, True)
    assert out.device.type == "cpu" and out.data_ptr() == a.data_ptr()
    check_close(out, g["ref_out_f32"], "cpu-staged")


@pytest.mark.parametrize("tile_m", ["32", "96", "128", "256"])
@pytest.mark.parametrize("block", [(128, 128), (64, 128)])
def test_fused_experts_fp8_tile_variants(ops, tile_m, block, knob):
    """Every grouped-GEMM tiling (32- and 96-row weight streaming, 128-row 2-stage, 256-row 3-deep ring) against the plain-C oracle; ragged expert
    loads (rows per expert not a multiple of either tile), wide dynamic range of block scales incl. zero/negative."""
    knob(SGLK_MOE_TILE_M=tile_m)
    M, N, K, E, topk = 1531, 256, 512, 8, 4
    bn, bk = block
    inp = recipes.moe_fp8_inputs(M, N, K, E, topk, bn, bk, False, 9001)
    g = torch.Generator().manual_seed(5)
    # scales spanning 2^-12 .. 2^4 with random sign, one exact zero block, one power of two
    inp["w1s"] = inp["w1s"].sign() * torch.exp2(torch.rand(inp["w1s"].shape, generator=g) * 16 - 12) * 1e-2
    inp["w1s"][0, 0, 0] = 0.0
    inp["w2s"][1, 0, 0] = 0.0
    inp["w2s"][2, 1, 1] = 2.0 ** -9
    ref = c_oracle.fused_experts_fp8(inp["a"], inp["w1"], inp["w2"], inp["w1s"], inp["w2s"], block,
                                     inp["topk_weight"], inp["topk_ids"])
    # bring the outputs into the O(1) range the reference's atol=1e-2 is meant for (its tests produce |out| <~ 2)
    k = float(2.0 / ref.abs().max())
    inp["topk_weight"] = inp["topk_weight"] * k
    out, _ = run_fp8(ops, inp, block)
    check_close(out, ref * k, f"tile_m={tile_m} block={block}")


@pytest.mark.parametrize("mode", ["0", "1", "2"])
def test_fused_experts_fp8_tail_tiles(ops, mode, knob):
    """Experts with one full 256-row tile plus a short tail (rows ~ 300): the tails run on the mid kernel -- off ("0"), on the
    caller's stream ("1"), or (default, "2") on the aux stream the Python layer owns and passes in (same bits as on the
    caller's); also under hipGraph capture."""
    knob(SGLK_TAIL_SPLIT=None if mode == "2" else mode)
    M, N, K, E, topk, block = 600, 256, 512, 8, 4, (128, 128)
    inp = recipes.moe_fp8_inputs(M, N, K, E, topk, block[0], block[1], False, 9055)
    ref = c_oracle.fused_experts_fp8(inp["a"], inp["w1"], inp["w2"], inp["w1s"], inp["w2s"], block,
                                     inp["topk_weight"], inp["topk_ids"])
    k = float(2.0 / ref.abs().max())
    inp["topk_weight"] = inp["topk_weight"] * k
    out, d = run_fp8(ops, inp, block)
    check_close(out, ref * k, f"tail split mode {mode}")
    # the stream the tails run on must not change the bits (the kernel they run on may: its rounding points differ)
    if mode == "1":
        test_fused_experts_fp8_tail_tiles._bits = out.cpu()
    elif mode == "2" and hasattr(test_fused_experts_fp8_tail_tiles, "_bits"):
        assert torch.equal(test_fused_experts_fp8_tail_tiles._bits, out.cpu())
    # capture + replay (the side stream is forked and joined inside the captured call)
    w1p, w2p = ops.convert_weight_packed(d["w1"]), ops.convert_weight_packed(d["w2"])
    call = lambda: ops.fused_experts_cpu(d["a"], w1p, w2p, d["topk_weight"], d["topk_ids"], False, False, True, d["w1s"],
                                         d["w2s"], list(block), None, None, True)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        call()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out_g = call()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out_g.cpu(), out.cpu())


@pytest.mark.parametrize("shape", [(200, 384, 640, 8, 2), (64, 256, 4352, 4, 2), (300, 384, 7168, 8, 4)],
                         ids=lambda s: "M%d_N%d_K%d_E%d_top%d" % s)
def test_fused_experts_fp8_mid_odd_and_long_reductions(ops, shape, knob):
    """The weight-streaming mid kernel beyond the Qwen3 shape: odd numbers of 128-wide K blocks (N = 384 -> 3, K = 640 -> 5)
    and more than 32 of them (K = 4352 -> 34, K = 7168 -> 56: the reference bench's literal expert shape), against the
    plain-C oracle."""
    knob(SGLK_MOE_TILE_M=96)
    M, N, K, E, topk = shape
    block = (128, 128)
    inp = recipes.moe_fp8_inputs(M, N, K, E, topk, block[0], block[1], False, 9077 + M)
    ref = c_oracle.fused_experts_fp8(inp["a"], inp["w1"], inp["w2"], inp["w1s"], inp["w2s"], block,
                                     inp["topk_weight"], inp["topk_ids"])
    k = float(2.0 / ref.abs().max())
    inp["topk_weight"] = inp["topk_weight"] * k
    out, _ = run_fp8(ops, inp, block)
    check_close(out, ref * k, f"mid kernel {shape}")


def test_fused_experts_fp8_rowmajor_weights_are_retiled_at_prefill_sizes(ops):
    """is_vnni=False (the reference's un-prepacked call form, /root/reference/test_moe_fp8_ext.py:118) from 64 rows per expert on:
    the call re-tiles the weights into its workspace (SGLK_MOE_PACK_WEIGHTS) and runs the packed kernels -- the same bits as a
    caller that packed them itself; below that size the row-major generic engine answers within the usual tolerance."""
    N, K, E, topk, bn, bk = 256, 512, 8, 2, 128, 128
    for M, exact in ((1500, True), (40, False)):
        inp = recipes.moe_fp8_inputs(M, N, K, E, topk, bn, bk, False, 6100 + M)
        packed, d = run_fp8(ops, inp, (bn, bk))
        rowmajor = ops.fused_experts_cpu(d["a"], d["w1"], d["w2"], d["topk_weight"], d["topk_ids"], False, False, True,
                                         d["w1s"], d["w2s"], [bn, bk], None, None, False)
        if exact:
            assert torch.equal(packed, rowmajor)
        ref = moe.fused_experts_fp8(inp["a"], inp["w1"], inp["w2"], inp["w1s"], inp["w2s"], (bn, bk), inp["topk_weight"], inp["topk_ids"])
        check_close(rowmajor, ref, f"row-major M={M}")


@pytest.mark.parametrize("shape", [(4, 384, 1024, 32, 8), (1, 384, 7168, 16, 8), (7, 128 * 3, 512, 8, 2), (20, 640, 768, 64, 4)],
                         ids=lambda s: "x".join(map(str, s)))
def test_fused_experts_fp8_decode_sizes_on_narrow_workgroups(ops, knob, shape):
    """Decode-size batches on the mid kernel (expert widths the 32-token stream kernel does not take, e.g. the reference's decode
    bench shape N = 384, K = 7168, /root/reference/bench_moe.py:144): GEMM-1 on four-wave workgroups of 64 ic1 columns and with the
    activations three K blocks ahead (moe_gemm_fp8w_mid.hip: NW, XD).  Every combination of the two, forced by knob, gives the SAME
    bits (the arithmetic of a column does not depend on which workgroup owns it), and the oracle's values."""
    from sgl_kernel import _lib, _ops
    M, N, K, E, topk = shape
    bn, bk = 128, 128
    inp = recipes.moe_fp8_inputs(M, N, K, E, topk, bn, bk, False, 5150 + M + N)
    ref = c_oracle.fused_experts_fp8(inp["a"], inp["w1"], inp["w2"], inp["w1s"], inp["w2s"], (bn, bk), inp["topk_weight"], inp["topk_ids"])
    k = float(2.0 / ref.abs().max())
    inp["topk_weight"] = inp["topk_weight"] * k
    outs = {}
    for nw, far in ((8, 0), (8, 1), (4, 0), (4, 1)):
        knob(SGLK_MOE_TILE_M=96, SGLK_MID_NW=nw, SGLK_MID_FAR=far)
        outs[(nw, far)], _ = run_fp8(ops, inp, (bn, bk))
        assert (_ops.last_path & _lib.PATH_TILE_MASK) == 96
    check_close(outs[(8, 0)], ref * k, f"mid kernel, decode size {shape}")
    for key, o in outs.items():
        assert torch.equal(o, outs[(8, 0)]), f"workgroup width / prefetch distance {key} changed the result"
    knob(SGLK_MOE_TILE_M=96, SGLK_MID_NW=None, SGLK_MID_FAR=None)
    default, _ = run_fp8(ops, inp, (bn, bk))
    assert torch.equal(default, outs[(8, 0)])


def test_pack_weights_flag_is_a_hint(ops):
    """C-ABI: SGLK_MOE_PACK_WEIGHTS on a call that cannot use it -- weights already packed, a workspace sized WITHOUT the flag
    (sglk_fused_experts_workspace_bytes), M == 0 -- runs on the weights as given instead of failing with "unknown flags"
    (ADVICE r2)."""
    import ctypes
    from sgl_kernel import _lib
    L = _lib.lib()
    M, N, K, E, topk, bn, bk = 96, 256, 512, 8, 2, 128, 128
    inp = recipes.moe_fp8_inputs(M, N, K, E, topk, bn, bk, False, 6200)
    ref_out, d = run_fp8(ops, inp, (bn, bk))
    w1p, w2p = ops.convert_weight_packed(d["w1"]), ops.convert_weight_packed(d["w2"])
    tw, ids = d["topk_weight"].float().contiguous(), d["topk_ids"].int().contiguous()
    w1s, w2s = d["w1s"].float().contiguous(), d["w2s"].float().contiguous()
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def call(m, w1, w2, packed, ws_bytes):
        out = torch.zeros(max(m, 1), K, dtype=torch.bfloat16, device="cuda")
        ws = torch.empty(max(ws_bytes, 256), dtype=torch.uint8, device="cuda")
        args = _lib.FusedExpertsArgs(
            hidden=d["a"].data_ptr(), hidden_stride=K, out=out.data_ptr(), out_stride=K, w1=w1.data_ptr(), w2=w2.data_ptr(),
            w1_scale=w1s.data_ptr(), w2_scale=w2s.data_ptr(), topk_weights=tw.data_ptr(), topk_ids=ids.data_ptr(),
            M=m, N=N, K=K, E=E, topk=topk, wtype=_lib.W_FP8_E4M3, packed=packed, block_n=bn, block_k=bk,
            workspace=ws.data_ptr(), workspace_bytes=ws_bytes, stage_timer=None, aux_stream=None,
            aux_events=(ctypes.c_void_p * 2)(), flags=_lib.MOE_PACK_WEIGHTS, path_taken=None)
        rc = L.sglk_fused_experts(ctypes.byref(args), stream)
        torch.cuda.synchronize()
        return rc, out

    small = L.sglk_fused_experts_workspace_bytes(M, N, K, E, topk, _lib.W_FP8_E4M3)
    rc, out = call(M, w1p, w2p, 3, small)                       # already packed
    assert rc == 0 and torch.equal(out, ref_out)
    rc, out = call(M, d["w1"], d["w2"], 0, small)               # row-major, but no room for the re-tiled copy: generic engine
    assert rc == 0
    check_close(out, moe.fused_experts_fp8(inp["a"], inp["w1"], inp["w2"], inp["w1s"], inp["w2s"], (bn, bk), inp["topk_weight"],
                                           inp["topk_ids"]), "row-major, workspace without room for the copy")
    rc, _ = call(0, d["w1"], d["w2"], 0, small)                 # nothing to do
    assert rc == 0


@pytest.mark.parametrize("M,topk,masked", [(1, 8, False), (3, 8, True), (4, 8, False), (4, 8, True), (16, 2, True), (32, 1, False)])
def test_fused_experts_fp8_decode_sizes_sort_ids_in_the_gemm_kernels(ops, M, topk, masked, knob):
    """At most 32 slots: no moe_align launch, the weight-streaming kernels derive their tile from topk_ids themselves
    (moe_align_inline.h).  Same bits as with the align launch, and the oracle's values; -1 ids skipped as in
    /root/reference/test_moe_offloading_cpu.py:62-68; several tokens on one expert."""
    from sgl_kernel import _lib, _ops
    N, K, E, bn, bk = 256, 512, 12, 128, 128      # twelve experts: tokens share experts
    inp = recipes.moe_fp8_inputs(M, N, K, E, topk, bn, bk, masked, 4242 + M)
    ref = c_oracle.fused_experts_fp8(inp["a"], inp["w1"], inp["w2"], inp["w1s"], inp["w2s"], (bn, bk),
                                     inp["topk_weight"], inp["topk_ids"])
    knob(SGLK_INLINE_ALIGN_MAX=32)     # the default stops at 16 slots (where it pays); the kernels take 32
    out, _ = run_fp8(ops, inp, (bn, bk))
    assert _ops.last_path & _lib.PATH_INLINE_ALIGN and (_ops.last_path & _lib.PATH_TILE_MASK) == 32
    check_close(out, ref, f"inline align M={M} topk={topk}")
    knob(SGLK_INLINE_ALIGN_MAX=0)
    out_t, _ = run_fp8(ops, inp, (bn, bk))
    assert not (_ops.last_path & _lib.PATH_INLINE_ALIGN)
    assert torch.equal(out, out_t)
