"""Torch restatement of the reference's attention oracles (test infrastructure): explicit fp32 masked softmax per
sequence instead of the reference's scaled_dot_product_attention calls — same mathematics."""
import torch


def _attend(q, k, v, scale, visible, logit_cap=0.0):
    """q [HQ,Lq,D], k [HKV,Lk,D], v [HKV,Lk,DV] -> [HQ,Lq,DV]; visible [Lq,Lk] bool; GQA by head repetition."""
    HQ, HKV = q.shape[0], k.shape[0]
    rep = HQ // HKV
    k = k.float().repeat_interleave(rep, dim=0)
    v = v.float().repeat_interleave(rep, dim=0)
    s = torch.einsum("hqd,hkd->hqk", q.float(), k) * scale
    if logit_cap > 0:
        s = logit_cap * torch.tanh(s / logit_cap)
    s = s.masked_fill(~visible.unsqueeze(0), float("-inf"))
    return torch.einsum("hqk,hkd->hqd", torch.softmax(s, dim=-1), v)


def extend_attention(q_extend, k_buffer, v_buffer, req_to_tokens, b_req_idx, b_seq_len, b_prefix_len, b_extend_len, scale,
                     logit_cap=0.0):
    """/root/reference/test_extend.py:10-76: per sequence, queries = its extend tokens, keys/values = the first seq_len
    entries of its page list (the extend tokens are already in the buffers); query i sees keys [0, prefix + i].
    Returns fp32 [extend tokens, HQ, DV]."""
    T, HQ, _ = q_extend.shape
    DV = v_buffer.shape[2]
    out = torch.zeros(T, HQ, DV)
    start = 0
    for b in range(b_seq_len.shape[0]):
        L, P, Eq = int(b_seq_len[b]), int(b_prefix_len[b]), int(b_extend_len[b])
        toks = req_to_tokens[int(b_req_idx[b]), :L].long()
        k, v = k_buffer[toks].movedim(0, 1), v_buffer[toks].movedim(0, 1)
        q = q_extend[start:start + Eq].movedim(0, 1)
        visible = torch.arange(L).unsqueeze(0) <= (P + torch.arange(Eq)).unsqueeze(1)
        out[start:start + Eq] = _attend(q, k, v, scale, visible, logit_cap).movedim(0, 1)
        start += Eq
    return out


def decode_attention(q, k_buffer, v_buffer, key, value, loc, req_to_token, b_req_idx, b_seq_len, scale, logit_cap=0.0):
    """/root/reference/test_mla.py:12-66: write the new token's key/value at loc, then one query per request over
    its whole page list.  Mutates k_buffer / v_buffer like the reference.  Returns fp32 [B, HQ, DV]."""
    k_buffer[loc.long()] = key
    v_buffer[loc.long()] = value
    B, HQ, _ = q.shape
    out = torch.zeros(B, HQ, v_buffer.shape[2])
    for b in range(B):
        L = int(b_seq_len[b])
        toks = req_to_token[int(b_req_idx[b]), :L].long()
        k, v = k_buffer[toks].movedim(0, 1), v_buffer[toks].movedim(0, 1)
        out[b] = _attend(q[b].unsqueeze(1), k, v, scale, torch.ones(1, L, dtype=torch.bool), logit_cap).squeeze(1)
    return out


def flash_attn_varlen(q, k, v, cu_seqlens_q, cu_seqlens_k, is_causal):
    """/root/reference/test_flash_attn_varlen.py:14-46: per sequence b, queries = rows cu_q[b]:cu_q[b+1] of q [Tq,H,D],
    keys/values = rows cu_k[b]:cu_k[b+1] of k [Tk,Hkv,D] / v [Tk,Hkv,DV]; scale 1/sqrt(D); is_causal = the top-left aligned
    mask of scaled_dot_product_attention (query i sees keys 0..i)."""
    cu_q, cu_k = cu_seqlens_q.tolist(), cu_seqlens_k.tolist()
    out = torch.zeros(q.shape[0], q.shape[1], v.shape[2], dtype=torch.float32)
    scale = 1.0 / q.shape[-1] ** 0.5
    for b in range(len(cu_k) - 1):
        qs, qe, ks, ke = cu_q[b], cu_q[b + 1], cu_k[b], cu_k[b + 1]
        Lq, Lk = qe - qs, ke - ks
        if is_causal:
            visible = torch.arange(Lk).view(1, Lk) <= torch.arange(Lq).view(Lq, 1)
        else:
            visible = torch.ones(Lq, Lk, dtype=torch.bool)
        o = _attend(q[qs:qe].transpose(0, 1), k[ks:ke].transpose(0, 1), v[ks:ke].transpose(0, 1), scale, visible)
        out[qs:qe] = o.transpose(0, 1)
    return out
