"""Expert-parallel fused_experts over torch.distributed (RCCL all-to-all over xGMI on MI355X).

The reference has no EP code; what it pins is the LOCAL contract EP needs: `topk_ids == -1` marks experts that are
not resident and must contribute nothing (/root/reference/test_moe_offloading_cpu.py:12-15,62-68).  This module
builds the exchange around that contract:

  rank r owns experts [r*E/G, (r+1)*E/G).  Every rank starts with its own tokens and their routing.
  dispatch : each token row is sent ONCE to every rank that owns at least one of its experts, together with its
             topk ids rewritten to the destination's local numbering (-1 elsewhere) and its routing weights;
             variable splits, `all_to_all_single` (direct all-to-all uses all 7 xGMI links at once; a ring would
             be bound by one link)
  local    : fused_experts on the received rows — the very same HIP path, -1 slots skipped
  combine  : partial rows (already weighted, summed over the local experts) return by the inverse all-to-all and
             are added per token in ascending rank order in fp32, one bf16 rounding (deterministic)

The split sizes of a variable all-to-all must be known on the host, so one tiny count exchange + host read
happens per call.
"""
import torch
import torch.distributed as dist


class ExpertParallelMoE:
    def __init__(self, num_experts, local_experts_fn, group=None):
        """local_experts_fn(hidden[R,K] bf16, topk_w[R,topk] f32, local_ids[R,topk] i32) -> [R,K] bf16."""
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        if num_experts % self.world != 0:
            raise ValueError(f"num_experts ({num_experts}) must be divisible by the EP world size ({self.world})")
        self.num_experts = num_experts
        self.experts_per_rank = num_experts // self.world
        self.local_fn = local_experts_fn
        self.last_stats = {}
        # gloo has no device all-to-all: with it (CPU tests, single-GPU rehearsals of the multi-rank flow) the payloads
        # are staged through host memory; with RCCL ("nccl") they stay on the GPU
        self.host_staged = dist.get_backend(group) == "gloo"

    def _all_to_all(self, out, inp, out_splits=None, in_splits=None):
        if self.host_staged and out.is_cuda:
            o = torch.empty(out.shape, dtype=out.dtype)
            dist.all_to_all_single(o, inp.cpu(), out_splits, in_splits, group=self.group)
            out.copy_(o)
        else:
            dist.all_to_all_single(out, inp, out_splits, in_splits, group=self.group)

    def plan(self, topk_ids):
        """Which (rank, token) pairs exchange rows.  Returns (send_tok, send_rank, send_counts[G])."""
        G, epr = self.world, self.experts_per_rank
        valid = (topk_ids >= 0) & (topk_ids < self.num_experts)
        dest = torch.where(valid, topk_ids // epr, torch.zeros_like(topk_ids)).long()
        member = torch.zeros(topk_ids.shape[0], G, dtype=torch.int32, device=topk_ids.device)
        member.scatter_add_(1, dest, valid.to(torch.int32))
        pairs = (member.t() > 0).nonzero()            # rows sorted by rank, then token (row-major order)
        send_rank, send_tok = pairs[:, 0], pairs[:, 1]
        send_counts = (member > 0).sum(dim=0)
        return send_tok, send_rank, send_counts

    def __call__(self, hidden, topk_weights, topk_ids):
        G, epr = self.world, self.experts_per_rank
        M, K = hidden.shape
        topk = topk_ids.shape[1]
        topk_ids = topk_ids.to(torch.int32)
        send_tok, send_rank, send_counts = self.plan(topk_ids)

        recv_counts = torch.empty_like(send_counts)
        self._all_to_all(recv_counts, send_counts)
        send_l = send_counts.tolist()
        recv_l = recv_counts.tolist()                  # host sync: split sizes must live on the host
        R_recv = sum(recv_l)

        # payload 1: token rows.  payload 2: per-row routing (local ids | weights bit-cast), one exchange
        rows = hidden.index_select(0, send_tok)
        ids_sel = topk_ids.index_select(0, send_tok)
        lo = (send_rank * epr).to(torch.int32).unsqueeze(1)
        local_ids = torch.where((ids_sel >= lo) & (ids_sel < lo + epr), ids_sel - lo, torch.full_like(ids_sel, -1))
        meta = torch.cat([local_ids, topk_weights.float().index_select(0, send_tok).view(torch.int32)], dim=1)

        recv_rows = torch.empty(R_recv, K, dtype=hidden.dtype, device=hidden.device)
        recv_meta = torch.empty(R_recv, 2 * topk, dtype=torch.int32, device=hidden.device)
        self._all_to_all(recv_rows, rows, recv_l, send_l)
        self._all_to_all(recv_meta, meta, recv_l, send_l)

        partial = self.local_fn(recv_rows, recv_meta[:, topk:].contiguous().view(torch.float32),
                                recv_meta[:, :topk].contiguous())

        back = torch.empty(rows.shape[0], K, dtype=hidden.dtype, device=hidden.device)
        self._all_to_all(back, partial.contiguous(), send_l, recv_l)

        # fixed-order reduce: segment d of `back` holds at most one row per token; rows are added per token in rank
        # order in fp32 and rounded once
        out = self._reduce(back, send_tok, send_rank, send_l, M)
        self.last_stats = dict(rows_sent=int(sum(send_l)), rows_received=int(R_recv),
                               bytes_sent=int(sum(send_l)) * K * hidden.element_size())
        return out

    def _reduce(self, back, send_tok, send_rank, send_l, M):
        G, K = self.world, back.shape[1]
        if back.is_cuda and back.dtype == torch.bfloat16 and K % 8 == 0:
            # one HIP launch: table[m][d] = row of `back` that rank d returned for token m (-1: none)
            from . import _lib
            table = torch.full((M, G), -1, dtype=torch.int32, device=back.device)
            table[send_tok, send_rank] = torch.arange(back.shape[0], dtype=torch.int32, device=back.device)
            out = torch.empty(M, K, dtype=back.dtype, device=back.device)
            _lib.check(_lib.lib().sglk_ep_reduce_rows(back.data_ptr(), back.stride(0), table.data_ptr(), G, out.data_ptr(),
                                                      out.stride(0), M, K, torch.cuda.current_stream(back.device).cuda_stream),
                       "ep_reduce_rows")
            return out
        # host tensors (the gloo tests of the exchange logic): the same sums, same order, in torch
        out = torch.zeros(M, K, dtype=torch.float32, device=back.device)
        off = 0
        for d in range(G):
            n = send_l[d]
            if n:
                out.index_add_(0, send_tok[off:off + n], back[off:off + n].float())
            off += n
        return out.to(back.dtype)


def masked_allgather_reference(hidden, topk_weights, topk_ids, num_experts, local_experts_fn, group=None):
    """The cheap EP formulation pinned by the reference's -1 contract, used as the cross-check of the all-to-all
    path: all-gather every rank's tokens, mask non-local experts to -1, run the local experts, sum the partial
    outputs over ranks and keep the own slice."""
    G = dist.get_world_size(group)
    r = dist.get_rank(group)
    epr = num_experts // G

    def gather(t):
        parts = [torch.empty_like(t) for _ in range(G)]
        dist.all_gather(parts, t.contiguous(), group=group)
        return torch.cat(parts, dim=0)

    all_h, all_w, all_ids = gather(hidden), gather(topk_weights.float()), gather(topk_ids.to(torch.int32))
    lo = r * epr
    local_ids = torch.where((all_ids >= lo) & (all_ids < lo + epr), all_ids - lo, torch.full_like(all_ids, -1))
    partial = local_experts_fn(all_h, all_w, local_ids).float()
    dist.all_reduce(partial, group=group)
    M = hidden.shape[0]
    return partial[r * M:(r + 1) * M].to(hidden.dtype)
