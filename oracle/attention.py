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
