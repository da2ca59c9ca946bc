"""Torch restatement of the reference's expert-routing oracles (test infrastructure).

Adds what the reference leaves open: a tie rule.  The reference's torch.topk(sorted=False) returns an unspecified
order and breaks ties arbitrarily; here larger value first and, among equal values, lower index first (stable sort),
experts of non-selected groups only after all experts of the selected groups.
"""
import torch


def _topk_stable(values, k):
    """top-k by (value desc, index asc); returns (values, indices)."""
    order = torch.sort(values, dim=-1, descending=True, stable=True).indices[..., :k]
    return values.gather(-1, order), order


def grouped_topk(gating, topk, renormalize, num_expert_group, topk_group):
    """/root/reference/test_grouped_topk.py:9-39 (softmax scores, group max, masked_fill(0.0))."""
    scores = torch.softmax(gating.float(), dim=-1)
    M, E = scores.shape
    G = num_expert_group
    group_scores = scores.view(M, G, -1).max(dim=-1).values
    _, gidx = _topk_stable(group_scores, topk_group)
    gmask = torch.zeros_like(group_scores, dtype=torch.bool).scatter_(1, gidx, True)
    emask = gmask.unsqueeze(-1).expand(M, G, E // G).reshape(M, E)
    # selection key: selected-group experts by score, the rest strictly below everything (they weigh 0 anyway)
    key = torch.where(emask, scores, torch.full_like(scores, -1.0))
    _, ids = _topk_stable(key, topk)
    w = torch.where(emask.gather(1, ids), scores.gather(1, ids), torch.zeros(M, topk))
    if renormalize:
        w = w / w.sum(dim=-1, keepdim=True)
    return w.float(), ids.to(torch.int32)


def biased_grouped_topk(gating, bias, topk, renormalize, num_expert_group, topk_group):
    """/root/reference/test_biased_grouped_topk.py:9-47 (sigmoid + bias, top-2-sum group score, -inf mask)."""
    scores = gating.float().sigmoid()
    M, E = scores.shape
    G = num_expert_group
    choice = scores + bias.float().unsqueeze(0)
    top2, _ = _topk_stable(choice.view(M, G, -1), 2)
    group_scores = top2.sum(dim=-1)
    _, gidx = _topk_stable(group_scores, topk_group)
    gmask = torch.zeros_like(group_scores, dtype=torch.bool).scatter_(1, gidx, True)
    emask = gmask.unsqueeze(-1).expand(M, G, E // G).reshape(M, E)
    key = choice.masked_fill(~emask, float("-inf"))
    _, ids = _topk_stable(key, topk)
    w = scores.gather(1, ids)
    if renormalize:
        w = w / w.sum(dim=-1, keepdim=True)
    return w.float(), ids.to(torch.int32)
