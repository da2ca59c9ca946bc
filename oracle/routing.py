"""Torch restatement of the reference's expert-routing oracles (test infrastructure).

Adds what the reference leaves open: a tie rule.  The reference's torch.topk(sorted=False) returns an unspecified
order and breaks ties arbitrarily; here larger value first and, among equal values, lower index first (stable sort),
experts of non-selected groups only after all experts of the selected groups.  Softmax variant: EQUAL fp32 scores that
come from different logits (softmax saturates: many scores round to the same value or to 0) are ranked by logit first,
index second.  Softmax is monotone, so that rule is the same as ranking by the logits directly -- which makes the ids a
function of exact input comparisons, independent of any exp() implementation; the function checks that the scores it
computed are indeed non-increasing along the logit order.
"""
import torch


def _topk_stable(values, k):
    """top-k by (value desc, index asc); returns (values, indices)."""
    order = torch.sort(values, dim=-1, descending=True, stable=True).indices[..., :k]
    return values.gather(-1, order), order


def grouped_topk(gating, topk, renormalize, num_expert_group, topk_group):
    """/root/reference/test_grouped_topk.py:9-39 (softmax scores, group max, masked_fill(0.0))."""
    logits = gating.float()
    scores = torch.softmax(logits, dim=-1)
    M, E = scores.shape
    G = num_expert_group
    # group score = max score of the group (test_grouped_topk.py:21-23) = the score of its max logit; ties by logit, index
    group_logit = logits.view(M, G, -1).max(dim=-1).values
    group_scores = scores.view(M, G, -1).max(dim=-1).values
    _, gidx = _topk_stable(group_logit, topk_group)
    gs = group_scores.gather(1, torch.sort(group_logit, dim=-1, descending=True, stable=True).indices)
    assert (gs[:, :-1] >= gs[:, 1:]).all(), "softmax scores are not monotone in the logits (group level)"
    gmask = torch.zeros_like(group_scores, dtype=torch.bool).scatter_(1, gidx, True)
    emask = gmask.unsqueeze(-1).expand(M, G, E // G).reshape(M, E)
    # selection key: selected-group experts by score (equal scores: by logit), the rest strictly below everything (they
    # weigh 0 anyway, masked_fill(0.0) in the reference) in index order
    key = torch.where(emask, logits, torch.full_like(logits, float("-inf")))
    order = torch.sort(key, dim=-1, descending=True, stable=True).indices
    ids = order[:, :topk]
    n_sel = emask.sum(dim=1, keepdim=True)
    along = torch.where(torch.arange(E).unsqueeze(0) < n_sel, scores.gather(1, order), torch.zeros_like(scores))
    assert (along[:, :-1] >= along[:, 1:]).all(), "softmax scores are not monotone in the logits"
    w = torch.where(emask.gather(1, ids), scores.gather(1, ids), torch.zeros(M, topk))
    if renormalize:
        w = w / w.sum(dim=-1, keepdim=True)
    return w.float(), ids.to(torch.int32)


def biased_grouped_topk(gating, bias, topk, renormalize, num_expert_group, topk_group, key_nudge=None):
    """/root/reference/test_biased_grouped_topk.py:9-47 (sigmoid + bias, top-2-sum group score, -inf mask).
    key_nudge (tests only): added to the ranking key, to ask whether another id set is within a few ulps of this one."""
    scores = gating.float().sigmoid()
    M, E = scores.shape
    G = num_expert_group
    choice = scores + bias.float().unsqueeze(0)
    if key_nudge is not None:
        choice = choice + key_nudge
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
