#!/usr/bin/env python3
"""Generate golden vectors from the reference's OWN embedded torch oracles.

Runs ONLY in the build container (needs /root/reference).  The reference's test
files cannot be imported (top-level ``import sgl_kernel`` + work at import time),
so the pure-torch oracle functions are lifted out of the files with ``ast`` and
executed here on seeded inputs.  What is committed under tests/golden/ is data:
inputs (or the seed recipe + an input checksum for large cases) and the outputs
the reference's oracle produced.  No reference source text is stored.

    python tests/golden/make_golden.py            # regenerate everything
    python tests/golden/make_golden.py moe_fp8    # one family

Families -> reference oracle used
  moe_fp8   native_fused_moe + scaled_weight   /root/reference/test_moe_fp8_ext.py:22-25,70-91
            (masked -1 ids variant)            /root/reference/test_moe_offloading_cpu.py:29-52
  moe_int8  torch_w8a8_per_column_moe          /root/reference/test_moe_int8.py:16-94
  moe_bf16  torch_naive_moe                    /root/reference/test_moe.py:22-54
  topk      grouped_topk_native / biased       /root/reference/test_grouped_topk.py:9-39,
                                               /root/reference/test_biased_grouped_topk.py:9-47
  norm      forward_native (rmsnorm)           /root/reference/test_norm.py:15-33
  act       silu_and_mul                       /root/reference/test_activation.py:14-16
  gemm      native_w8a8_per_token_matmul etc.  /root/reference/test_gemm_int8.py:14-47,
                                               /root/reference/test_gemm_fp8.py:22-49
  attn      _run_sdpa_forward_extend/decode    /root/reference/test_extend.py:10-76,
                                               /root/reference/test_mla.py:12-66
  absorb    native_torch / native_torch_int8   /root/reference/test_absorb.py:20-109
  varlen    flash_attn_varlen_ref              /root/reference/test_flash_attn_varlen.py:14-46
"""
import ast
import hashlib
import math
import os
import sys

import torch
import torch.nn.functional as F
from safetensors.torch import save_file

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def lift(path, names, extra_globals=None):
    """Compile the named top-level FunctionDefs / ClassDefs of a reference file into a fresh namespace."""
    src = open(os.path.join(REF, path)).read()
    tree = ast.parse(src)
    keep = [n for n in tree.body if isinstance(n, (ast.FunctionDef, ast.ClassDef)) and n.name in names]
    missing = set(names) - {n.name for n in keep}
    if missing:
        raise RuntimeError(f"{path}: functions not found: {missing}")
    mod = ast.Module(body=keep, type_ignores=[])
    ns = {"torch": torch, "F": F, "math": math}
    if extra_globals:
        ns.update(extra_globals)
    exec(compile(mod, os.path.join(REF, path), "exec"), ns)
    return ns


def checksum(*tensors):
    h = hashlib.sha256()
    for t in tensors:
        t = t.contiguous()
        if t.dtype in (torch.float8_e4m3fn, torch.bfloat16):
            t = t.view(torch.uint8 if t.dtype == torch.float8_e4m3fn else torch.int16)
        h.update(t.numpy().tobytes())
    return h.hexdigest()


def as_saveable(d):
    out = {}
    for k, v in d.items():
        v = v.contiguous()
        if v.dtype == torch.float8_e4m3fn:
            out[k + "__fp8"] = v.view(torch.uint8)
        else:
            out[k] = v
    return out


def save(name, tensors, meta):
    meta = {k: str(v) for k, v in meta.items()}
    save_file(as_saveable(tensors), os.path.join(OUT, name + ".safetensors"), metadata=meta)
    sz = os.path.getsize(os.path.join(OUT, name + ".safetensors"))
    print(f"  wrote {name}.safetensors ({sz/1024:.0f} KiB) meta={meta}")


# --------------------------------------------------------------------------------------
# input recipes (shared with tests/ via tests/recipes.py — keep in sync: this file imports it)
# --------------------------------------------------------------------------------------
sys.path.insert(0, os.path.join(OUT, ".."))
import recipes  # noqa: E402


def gen_moe_fp8():
    for case in recipes.MOE_FP8_CASES:
        name, M, N, K, E, topk, bn, bk, masked, seed, full = case
        inp = recipes.moe_fp8_inputs(M, N, K, E, topk, bn, bk, masked, seed)
        if masked:
            ns = lift("test_moe_offloading_cpu.py", ["SiluAndMul", "scaled_weight", "native_fused_moe"],
                      {"BLOCK_N": bn, "BLOCK_K": bk})
        else:
            ns = lift("test_moe_fp8_ext.py", ["SiluAndMul", "scaled_weight", "native_fused_moe"],
                      {"BLOCK_N": bn, "BLOCK_K": bk})
        w1_scaled = ns["scaled_weight"](inp["w1"], inp["w1s"])
        w2_scaled = ns["scaled_weight"](inp["w2"], inp["w2s"])
        ids = inp["topk_ids"] if masked else inp["topk_ids"].to(torch.int64)
        ref = ns["native_fused_moe"](inp["a"], w1_scaled, w2_scaled, inp["topk_weight"], ids, topk)
        ref = ref.float()
        meta = dict(M=inp["a"].shape[0], N=N, K=K, E=E, topk=topk, block_n=bn, block_k=bk, masked=int(masked),
                    seed=seed, full=int(full),
                    input_sha256=checksum(inp["a"], inp["w1"], inp["w2"], inp["w1s"], inp["w2s"],
                                          inp["topk_weight"], inp["topk_ids"]))
        tensors = {"ref_out_f32": ref}
        if full:
            tensors.update(inp)
        save("moe_fp8_" + name, tensors, meta)


def gen_moe_int8():
    ns = lift("test_moe_int8.py", ["silu_and_mul", "per_token_quant_int8", "native_w8a8_per_token_matmul",
                                   "torch_w8a8_per_column_moe"])
    for case in recipes.MOE_INT8_CASES:
        name, M, N, K, E, topk, seed, full = case
        inp = recipes.moe_int8_inputs(M, N, K, E, topk, seed)
        ref = ns["torch_w8a8_per_column_moe"](inp["a"], inp["w1"], inp["w2"], inp["w1s"], inp["w2s"],
                                              inp["topk_weight"], inp["topk_ids"].to(torch.int64), topk)
        meta = dict(M=M, N=N, K=K, E=E, topk=topk, seed=seed, full=int(full),
                    input_sha256=checksum(inp["a"], inp["w1"], inp["w2"], inp["w1s"], inp["w2s"],
                                          inp["topk_weight"], inp["topk_ids"]))
        tensors = {"ref_out": ref}
        if full:
            tensors.update(inp)
        save("moe_int8_" + name, tensors, meta)


def gen_moe_bf16():
    ns = lift("test_moe.py", ["SiluAndMul", "torch_naive_moe"])
    for case in recipes.MOE_BF16_CASES:
        name, M, N, K, E, topk, renorm, seed, full = case
        inp = recipes.moe_bf16_inputs(M, N, K, E, topk, seed)
        ref = ns["torch_naive_moe"](inp["a"], inp["w1"], inp["w2"], inp["score"], topk, renorm)
        meta = dict(M=M, N=N, K=K, E=E, topk=topk, renorm=int(renorm), seed=seed, full=int(full),
                    input_sha256=checksum(inp["a"], inp["w1"], inp["w2"], inp["score"]))
        tensors = {"ref_out": ref}
        if full:
            tensors.update(inp)
        save("moe_bf16_" + name, tensors, meta)


def gen_topk():
    ns = lift("test_grouped_topk.py", ["grouped_topk_native"])
    nsb = lift("test_biased_grouped_topk.py", ["biased_grouped_topk"])
    for case in recipes.TOPK_CASES:
        name, M, E, G, topk, topk_group, renorm, biased, seed = case
        inp = recipes.topk_inputs(M, E, biased, seed)
        if biased:
            w, ids = nsb["biased_grouped_topk"](inp["hidden"].float(), inp["gating"].float(), inp["bias"].float(),
                                                topk, renorm, G, topk_group)
        else:
            w, ids = ns["grouped_topk_native"](inp["hidden"].float(), inp["gating"].float(), topk, renorm, G, topk_group)
        tensors = dict(inp)
        tensors["ref_w"] = w.float()
        tensors["ref_ids"] = ids.to(torch.int32)
        save("topk_" + name, tensors, dict(M=M, E=E, G=G, topk=topk, topk_group=topk_group, renorm=int(renorm),
                                            biased=int(biased), seed=seed))


def gen_gemm():
    ns = lift("test_gemm_int8.py", ["per_token_quant_int8", "native_w8a8_per_token_matmul"])
    for name, M, N, K, has_bias, seed in recipes.GEMM_INT8_CASES:
        inp = recipes.gemm_int8_inputs(M, N, K, has_bias, seed)
        Aq, As = ns["per_token_quant_int8"](inp["A"])
        ref = ns["native_w8a8_per_token_matmul"](Aq, inp["Bq"], As, inp["Bs"], inp.get("bias"), torch.bfloat16)
        save("gemm_int8_" + name, {"ref_out": ref, "ref_Aq": Aq, "ref_As": As.float().reshape(-1)},
             dict(M=M, N=N, K=K, seed=seed, input_sha256=checksum(inp["A"], inp["Bq"], inp["Bs"])))
    nsf = lift("test_gemm_fp8.py", ["scaled_weight"], {"BLOCK_N": 64, "BLOCK_K": 128})
    for name, M, N, K, has_bias, chunk, seed in recipes.GEMM_FP8_CASES:
        inp = recipes.gemm_fp8_inputs(M, N, K, has_bias, chunk, seed)
        # /root/reference/test_gemm_fp8.py:41-49 (inline): bf16 dequantised weight, bf16 matmul, bf16 bias add
        ws = nsf["scaled_weight"](inp["w"], inp["scales"]).view(N, K).to(torch.bfloat16)
        ref = torch.matmul(inp["data"].to(torch.bfloat16), ws.T)
        if has_bias:
            ref = ref + inp["bias"].to(torch.bfloat16)
        # and the fp32 formulation of the same product (what the implementation is held to more tightly)
        ref32 = inp["data"].float() @ nsf["scaled_weight"](inp["w"], inp["scales"]).view(N, K).T
        if has_bias:
            ref32 = ref32 + inp["bias"]
        save("gemm_fp8_" + name, {"ref_out_bf16": ref, "ref_out_f32": ref32},
             dict(M=M, N=N, K=K, seed=seed, input_sha256=checksum(inp["data"].contiguous(), inp["w"], inp["scales"])))
    for name, M, N, K, has_bias, seed in recipes.GEMM_BF16_CASES:
        inp = recipes.gemm_bf16_inputs(M, N, K, has_bias, seed)
        # /root/reference/test_gemm.py:15-20 (inline)
        ref = torch.matmul(inp["mat1"].float(), inp["mat2"].float().t())
        if has_bias:
            ref.add_(inp["bias"].bfloat16())
        save("gemm_bf16_" + name, {"ref_out": ref.bfloat16()},
             dict(M=M, N=N, K=K, seed=seed, input_sha256=checksum(inp["mat1"], inp["mat2"])))


def gen_shared():
    ns = lift("test_shared_experts.py", ["SiluAndMul", "per_token_quant_int8", "native_w8a8_per_token_matmul",
                                         "torch_naive_moe", "torch_w8a8_per_column_moe"])
    for name, m, n, k, rsf, seed in recipes.SHARED_CASES:
        inp = recipes.shared_inputs(m, n, k, seed)
        ref = ns["torch_naive_moe"](inp["hs"].float(), inp["w1"].float(), inp["w2"].float(), inp["fused"].float(), rsf)
        w1q, w1s = ns["per_token_quant_int8"](inp["w1"])
        w2q, w2s = ns["per_token_quant_int8"](inp["w2"])
        ref8 = ns["torch_w8a8_per_column_moe"](inp["hs"].float(), w1q, w2q, w1s, w2s, inp["fused"].float(), rsf)
        save("shared_" + name, {"ref_bf16": ref.bfloat16(), "ref_int8": ref8.bfloat16(), "w1q": w1q, "w2q": w2q,
                                "w1s": w1s.float().reshape(-1), "w2s": w2s.float().reshape(-1)},
             dict(m=m, n=n, k=k, rsf=rsf, seed=seed, input_sha256=checksum(inp["hs"], inp["w1"], inp["w2"], inp["fused"])))
    nsf = lift("test_moe_fp8_ext.py", ["SiluAndMul", "scaled_weight"], {"BLOCK_N": 64, "BLOCK_K": 128})
    for name, M, N, K, rsf, seed in recipes.SHARED_FP8_CASES:
        inp = recipes.shared_fp8_inputs(M, N, K, seed)
        # /root/reference/test_moe_fp8_ext.py:39-56 (inline, fp32)
        w1sc = nsf["scaled_weight"](inp["w1"][None], inp["w1s"][None]).view(2 * N, K)
        w2sc = nsf["scaled_weight"](inp["w2"][None], inp["w2s"][None]).view(K, N)
        ic1 = nsf["SiluAndMul"](torch.matmul(inp["a"].float(), w1sc.transpose(0, 1)))
        ref = torch.matmul(ic1, w2sc.transpose(0, 1)) + inp["fused"].float() * rsf
        save("shared_fp8_" + name, {"ref_out_f32": ref},
             dict(M=M, N=N, K=K, rsf=rsf, seed=seed,
                  input_sha256=checksum(inp["a"], inp["w1"], inp["w2"], inp["w1s"], inp["w2s"], inp["fused"])))


def gen_rows():
    from typing import Optional, Tuple, Union
    ns = lift("test_norm.py", ["forward_native"], {"Optional": Optional, "Tuple": Tuple, "Union": Union})
    for name, rows, hidden, dtype, seed in recipes.NORM_CASES:
        inp = recipes.norm_inputs(rows, hidden, dtype, seed)
        out = ns["forward_native"](inp["x"], inp["w"], 1e-6)
        out2, res2 = ns["forward_native"](inp["x"], inp["w"], 1e-6, inp["res"])
        save("norm_" + name, {"ref_out": out, "ref_fused_out": out2, "ref_fused_res": res2},
             dict(rows=rows, hidden=hidden, dtype=str(dtype), seed=seed, input_sha256=checksum(
                 *(t.view(torch.int16) for t in (inp["x"], inp["w"], inp["res"])))))
    nsa = lift("test_activation.py", ["forward_native"])
    for name, rows, two_d, dtype, seed in recipes.ACT_CASES:
        inp = recipes.act_inputs(rows, two_d, dtype, seed)
        save("act_" + name, {"ref_out": nsa["forward_native"](inp["x"])},
             dict(rows=rows, two_d=two_d, dtype=str(dtype), seed=seed, input_sha256=checksum(inp["x"].view(torch.int16))))


def gen_attn():
    from torch.nn.functional import scaled_dot_product_attention
    ns = lift("test_extend.py", ["_run_sdpa_forward_extend"], {"scaled_dot_product_attention": scaled_dot_product_attention})
    for name, B, N_CTX, HQ, HKV, D, DV, mla, seed in recipes.EXTEND_CASES:
        inp = recipes.extend_inputs(B, N_CTX, HQ, HKV, D, DV, mla, seed)
        o_ref = torch.empty(inp["q_extend"].shape[0], HQ, DV, dtype=torch.bfloat16)
        ns["_run_sdpa_forward_extend"](inp["q_extend"], o_ref, inp["k_buffer"], inp["v_buffer"], inp["req_to_tokens"],
                                       inp["b_req_idx"], inp["b_seq_len"], inp["b_prefix"], inp["b_extend"],
                                       scaling=1.0 / D ** 0.5, enable_gqa=HQ != HKV, causal=True)
        save("extend_" + name, {"ref_out": o_ref},
             dict(B=B, HQ=HQ, HKV=HKV, D=D, DV=DV, mla=int(mla), seed=seed,
                  input_sha256=checksum(inp["q_extend"], inp["k_buffer"], inp["v_buffer"], inp["req_to_tokens"])))
    nsd = lift("test_mla.py", ["_run_sdpa_forward_decode"], {"scaled_dot_product_attention": scaled_dot_product_attention})
    for name, B, HQ, HKV, D, DV, seq_len, v_alias, seed in recipes.DECODE_CASES:
        inp = recipes.decode_inputs(B, HQ, HKV, D, DV, seq_len, v_alias, seed)
        kb = inp["k_buffer"].clone()
        if v_alias:
            vb, value = kb.narrow(2, 0, DV), inp["key"].narrow(2, 0, DV)
        else:
            vb, value = inp["v_buffer"].clone(), inp["value"]
            vb[inp["loc"]] = value      # the reference oracle only writes the key (MLA: v aliases k); GQA needs v too
        o_ref = torch.zeros(B, HQ, DV, dtype=torch.bfloat16)
        nsd["_run_sdpa_forward_decode"](inp["q"], o_ref, kb, vb, inp["key"], inp["loc"], inp["req_to_token"],
                                        inp["b_req_idx"], inp["b_seq_len"], scaling=1.0 / D ** 0.5, enable_gqa=HQ != HKV)
        save("decode_" + name, {"ref_out": o_ref},
             dict(B=B, HQ=HQ, HKV=HKV, D=D, DV=DV, seq_len=seq_len, v_alias=int(v_alias), seed=seed,
                  input_sha256=checksum(inp["q"], inp["k_buffer"], inp["key"], inp["loc"])))


def gen_attn_big():
    """BASELINE.json config 3 sizes (seqlen <= 8k): the reference oracle runs on the whole problem, the golden keeps its
    output for a sample of token rows (all heads) -- the inputs are regenerated from the seed by tests/recipes.py."""
    from torch.nn.functional import scaled_dot_product_attention
    ns = lift("test_extend.py", ["_run_sdpa_forward_extend"], {"scaled_dot_product_attention": scaled_dot_product_attention})
    for name, B, N_CTX, HQ, HKV, D, DV, mla, seed in recipes.EXTEND_BIG_CASES:
        inp = recipes.extend_inputs_fixed(B, N_CTX, HQ, HKV, D, DV, mla, seed)
        T = inp["q_extend"].shape[0]
        o_ref = torch.empty(T, HQ, DV, dtype=torch.bfloat16)
        ns["_run_sdpa_forward_extend"](inp["q_extend"], o_ref, inp["k_buffer"], inp["v_buffer"], inp["req_to_tokens"],
                                       inp["b_req_idx"], inp["b_seq_len"], inp["b_prefix"], inp["b_extend"],
                                       scaling=1.0 / D ** 0.5, enable_gqa=HQ != HKV, causal=True)
        rows = recipes.sample_rows(T, 160, seed)
        save("extend_" + name, {"rows": rows, "ref_out_rows": o_ref[rows].clone()},
             dict(B=B, N_CTX=N_CTX, HQ=HQ, HKV=HKV, D=D, DV=DV, mla=int(mla), seed=seed, tokens=T,
                  input_sha256=checksum(inp["q_extend"], inp["k_buffer"], inp["v_buffer"], inp["req_to_tokens"])))
    nsd = lift("test_mla.py", ["_run_sdpa_forward_decode"], {"scaled_dot_product_attention": scaled_dot_product_attention})
    for name, B, HQ, HKV, D, DV, seq_len, v_alias, seed in recipes.DECODE_BIG_CASES:
        inp = recipes.decode_inputs(B, HQ, HKV, D, DV, seq_len, v_alias, seed)
        kb = inp["k_buffer"].clone()
        if v_alias:
            vb, value = kb.narrow(2, 0, DV), inp["key"].narrow(2, 0, DV)
        else:
            vb, value = inp["v_buffer"].clone(), inp["value"]
            vb[inp["loc"]] = value
        o_ref = torch.zeros(B, HQ, DV, dtype=torch.bfloat16)
        nsd["_run_sdpa_forward_decode"](inp["q"], o_ref, kb, vb, inp["key"], inp["loc"], inp["req_to_token"],
                                        inp["b_req_idx"], inp["b_seq_len"], scaling=1.0 / D ** 0.5, enable_gqa=HQ != HKV)
        save("decode_" + name, {"ref_out": o_ref},
             dict(B=B, HQ=HQ, HKV=HKV, D=D, DV=DV, seq_len=seq_len, v_alias=int(v_alias), seed=seed,
                  input_sha256=checksum(inp["q"], inp["key"], inp["loc"])))


def gen_absorb():
    """qkv_proj_with_rope: native_torch / native_torch_int8 of /root/reference/test_absorb.py:65-109 (and their helpers
    :20-63), run with the file's own module constants (:10-17)."""
    d = recipes.ABSORB_DIMS
    consts = dict(kv_lora_rank=d["kv_lora_rank"], qk_head_dim=d["qk_nope_head_dim"] + d["qk_rope_head_dim"],
                  qk_nope_head_dim=d["qk_nope_head_dim"], qk_rope_head_dim=d["qk_rope_head_dim"],
                  rotary_dim=d["qk_rope_head_dim"], num_heads=d["num_heads"], q_lora_rank=d["q_lora_rank"])
    ns = lift("test_absorb.py", ["layernorm", "_rotate_gptj", "per_token_quant_int8", "native_w8a8_per_token_matmul",
                                 "rotary_emb", "native_torch", "native_torch_int8"], consts)
    for name, B, hidden, seed in recipes.ABSORB_CASES:
        inp = recipes.absorb_inputs(B, hidden, seed)
        bf = torch.bfloat16
        q_in = torch.zeros(B, d["num_heads"], d["kv_lora_rank"] + d["qk_rope_head_dim"], dtype=bf)
        q, k, v = ns["native_torch"](q_in, inp["hidden_states"], inp["q_a_proj_weight"], inp["norm_weight1"],
                                     inp["q_b_proj_weight"], inp["w_kc"].transpose(1, 2), inp["kv_a_proj_weight"],
                                     inp["norm_weight2"], inp["pos"], inp["cos_sin_cache"])
        w1q, w1s = ns["per_token_quant_int8"](inp["q_a_proj_weight"])
        w2q, w2s = ns["per_token_quant_int8"](inp["q_b_proj_weight"])
        w3q, w3s = ns["per_token_quant_int8"](inp["kv_a_proj_weight"])
        q_in8 = torch.zeros_like(q_in)
        q8, k8, v8 = ns["native_torch_int8"](q_in8, inp["hidden_states"], w1q, w1s, inp["norm_weight1"], w2q, w2s,
                                             inp["w_kc"].transpose(1, 2), w3q, w3s, inp["norm_weight2"], inp["pos"],
                                             inp["cos_sin_cache"])
        save("absorb_" + name, {"q": q.clone(), "k": k.clone(), "v": v.clone(), "q_int8": q8.clone(), "k_int8": k8.clone(),
                                "v_int8": v8.clone()},
             dict(B=B, hidden=hidden, seed=seed, input_sha256=checksum(inp["hidden_states"], inp["q_a_proj_weight"], inp["w_kc"])))


def gen_varlen():
    """flash_attn_varlen_func: flash_attn_varlen_ref of /root/reference/test_flash_attn_varlen.py:14-46."""
    ns = lift("test_flash_attn_varlen.py", ["flash_attn_varlen_ref"])
    for name, batch, mq, mk, H, Hkv, D, DV, causal, varlen, seed in recipes.VARLEN_CASES:
        inp = recipes.varlen_inputs(batch, mq, mk, H, Hkv, D, DV, varlen, seed)
        ref = ns["flash_attn_varlen_ref"](inp["q"], inp["k"], inp["v"], inp["cu_q"], inp["cu_k"], is_causal=causal,
                                          enable_gqa=H != Hkv)
        save("varlen_" + name, {"ref_out": ref.contiguous()},
             dict(batch=batch, H=H, Hkv=Hkv, D=D, DV=DV, causal=int(causal), seed=seed,
                  input_sha256=checksum(inp["q"], inp["k"], inp["v"], inp["cu_q"], inp["cu_k"])))


def gen_bmm():
    """bmm_cpu: the reference's expectation is torch.bmm(matA, matB) on bf16 (/root/reference/test_bmm_fp8.py:57, matB =
    mat2 viewed [B, K, N])."""
    for name, B, M, N, K, chunk, seed in recipes.BMM_CASES:
        inp = recipes.bmm_inputs(B, M, N, K, chunk, seed)
        ref = torch.bmm(inp["mat1"], inp["mat2"].transpose(1, 2))
        save("bmm_" + name, {"ref_out": ref.contiguous()},
             dict(B=B, M=M, N=N, K=K, seed=seed, input_sha256=checksum(inp["mat1"].contiguous(), inp["mat2"])))


def gen_mxfp4():
    """mxfp4_scaled_mm_cpu: MXFP4QuantizeUtil (quantize / dequantize) of /root/reference/test_mxfp4.py:14-127 and the
    expectation matmul(A.float(), Bdq.float().t()).bfloat16() (+ bias) of :166-168,196-198."""
    ns = lift("test_mxfp4.py", ["MXFP4QuantizeUtil"], {"block_size": 32})
    util = ns["MXFP4QuantizeUtil"]
    for name, M, N, K, kind, has_bias, seed in recipes.MXFP4_CASES:
        inp = recipes.mxfp4_inputs(M, N, K, kind, has_bias, seed, util.quantize)
        dq = util.dequantize(inp["wq"], torch.bfloat16, inp["ws"])
        ref = torch.matmul(inp["a"].float(), dq.float().t()).bfloat16()
        if inp["bias"] is not None:
            ref.add_(inp["bias"].view(1, -1))
        tensors = {"ref_out": ref, "dq": dq.contiguous(), "wq": inp["wq"], "ws": inp["ws"]}
        save("mxfp4_" + name, tensors, dict(M=M, N=N, K=K, kind=kind, seed=seed, input_sha256=checksum(inp["a"], inp["wq"], inp["ws"])))


FAMILIES = {"attn_big": gen_attn_big, "moe_fp8": gen_moe_fp8, "moe_int8": gen_moe_int8, "moe_bf16": gen_moe_bf16, "topk": gen_topk, "gemm": gen_gemm,
            "shared": gen_shared, "rows": gen_rows, "attn": gen_attn, "absorb": gen_absorb, "varlen": gen_varlen, "bmm": gen_bmm, "mxfp4": gen_mxfp4}


if __name__ == "__main__":
    which = sys.argv[1:] or list(FAMILIES)
    for fam in which:
        print(f"[{fam}]")
        FAMILIES[fam]()
