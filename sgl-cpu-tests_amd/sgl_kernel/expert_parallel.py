"""Expert-parallel fused_experts over torch.distributed (RCCL all-to-all over xGMI on MI355X).

The reference has no EP code; what it pins is the LOCAL contract EP needs: `topk_ids == -1` marks experts that are
not resident and must contribute nothing (/root/reference/test_moe_offloading_cpu.py:12-15,62-68).  This module
builds the exchange around that contract:

  rank r owns experts [r*E/G, (r+1)*E/G).  Every rank starts with its own tokens and their routing.
  plan     : ONE HIP launch (sglk_ep_plan): per destination rank the tokens that route at least one slot to it, in
             ascending token order (deterministic), as counts[G] and pos[M][G]
  dispatch : ONE all-to-all of ONE payload: row = [token row bf16 | topk ids rewritten to the destination's local
             numbering, -1 elsewhere | topk routing weights], packed by one HIP launch (sglk_ep_pack); direct all-to-all
             uses all 7 xGMI links at once (a ring would be bound by one link)
  local    : fused_experts on the received rows -- the very same HIP path, -1 slots skipped; the token rows are read in
             place from the payload (a row-strided view)
  combine  : partial rows (already weighted, summed over the local experts) return by the inverse all-to-all and are
             added per token in ascending rank order in fp32, one bf16 rounding (sglk_ep_reduce_rows; deterministic)

Split sizes.  A variable all-to-all needs its split sizes on the host: `capacity_factor=None` (exact mode) exchanges the
counts and reads them once per call (one host sync).  `capacity_factor=c` (0 < c <= 1) sends fixed segments of
ceil(c * M) rows per destination instead: no count exchange, NO host read, hipGraph-capturable; unused rows carry ids = -1
and cost wire bytes (c = 1 can never overflow: a token goes to a rank at most once; an overflow is flagged in
`last_overflow` and the surplus tokens lose that rank's experts).

Rounding.  Every rank rounds its partial sum to bf16 before it travels, and the G partials are added in fp32 and rounded
once more, whereas one GPU adds all topk slot rows in fp32 and rounds once.  The G-GPU result therefore differs from the
1-GPU result by at most G extra bf16 roundings of partial sums: |diff| <= 2^-8 * (sum_d |partial_d| + |result|) (asserted in
tests/test_expert_parallel_gloo.py); it meets the reference's predicate allclose(rtol=atol=1e-2) like the 1-GPU result.
"""
import ctypes
import math

import torch
import torch.distributed as dist


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def plan_torch(topk_ids, num_experts, world, capacity=0):
    """Torch formulation of sglk_ep_plan (CPU tensors of the gloo tests; the GPU test compares the kernel with it)."""
    epr = num_experts // world
    M = topk_ids.shape[0]
    valid = (topk_ids >= 0) & (topk_ids < num_experts)
    dest = torch.where(valid, topk_ids // epr, torch.zeros_like(topk_ids)).long()
    member = torch.zeros(M, world, dtype=torch.int32, device=topk_ids.device)
    member.scatter_add_(1, dest, valid.to(torch.int32))
    member = member > 0
    counts = member.sum(dim=0).to(torch.int32)
    pos = torch.cumsum(member.to(torch.int32), dim=0).to(torch.int32) - 1
    pos = torch.where(member, pos, torch.full_like(pos, -1))
    overflow = 0
    if capacity > 0:
        for d in range(world):
            if int(counts[d]) > capacity:
                overflow |= 1 << d
        pos = torch.where(pos >= capacity, torch.full_like(pos, -1), pos)
        seg_start = torch.arange(world + 1, dtype=torch.int32, device=topk_ids.device) * capacity
    else:
        seg_start = torch.zeros(world + 1, dtype=torch.int32, device=topk_ids.device)
        seg_start[1:] = torch.cumsum(counts, 0)
    return counts, seg_start, pos, overflow


def pack_torch(hidden, topk_weights, topk_ids, pos, seg_start, counts, row_bytes, num_experts, world, capacity=0):
    """Torch formulation of sglk_ep_pack: uint8 payload [rows][row_bytes]."""
    M, K = hidden.shape
    topk = topk_ids.shape[1]
    epr = num_experts // world
    rows = int(seg_start[world])
    payload = torch.zeros(rows, row_bytes, dtype=torch.uint8, device=hidden.device)
    for d in range(world):
        sel = torch.nonzero(pos[:, d] >= 0).flatten()
        dst = (int(seg_start[d]) + pos[sel, d]).long()
        lo = d * epr
        ids = topk_ids[sel]
        local = torch.where((ids >= lo) & (ids < lo + epr), ids - lo, torch.full_like(ids, -1)).to(torch.int32)
        payload[dst, :2 * K] = hidden[sel].contiguous().view(torch.uint8).reshape(-1, 2 * K)
        payload[dst, 2 * K:2 * K + 4 * topk] = local.contiguous().view(torch.uint8).reshape(-1, 4 * topk)
        payload[dst, 2 * K + 4 * topk:2 * K + 8 * topk] = topk_weights[sel].float().contiguous().view(torch.uint8).reshape(-1, 4 * topk)
        if capacity > 0:
            pad = torch.arange(int(seg_start[d]) + int(min(int(counts[d]), capacity)), int(seg_start[d]) + capacity)
            payload[pad, 2 * K:2 * K + 4 * topk] = torch.full((pad.numel(), topk), -1, dtype=torch.int32).view(torch.uint8).reshape(-1, 4 * topk)
    return payload


class ExpertParallelMoE:
    def __init__(self, num_experts, local_experts_fn, group=None, capacity_factor=None, profile=False):
        """local_experts_fn(hidden[R,K] bf16 (may be row-strided), topk_w[R,topk] f32, local_ids[R,topk] i32) -> [R,K] bf16."""
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        if num_experts % self.world != 0:
            raise ValueError(f"num_experts ({num_experts}) must be divisible by the EP world size ({self.world})")
        if capacity_factor is not None and not (0.0 < capacity_factor <= 1.0):
            raise ValueError("capacity_factor must be in (0, 1] (1 = a segment can hold every token of the rank)")
        self.num_experts = num_experts
        self.experts_per_rank = num_experts // self.world
        self.local_fn = local_experts_fn
        self.capacity_factor = capacity_factor
        self.last_stats = {}
        self.last_overflow = None       # capacity mode: device int32 bit mask of destinations that overflowed (lazy: no sync)
        self.profile = profile
        self._events = []
        # gloo has no device all-to-all: with it (CPU tests, single-GPU rehearsals of the multi-rank flow) the payloads
        # are staged through host memory; with RCCL ("nccl") they stay on the GPU
        self.host_staged = dist.get_backend(group) == "gloo"

    # ---- collectives -------------------------------------------------------------------------------------------------
    def _all_to_all(self, out, inp, out_splits=None, in_splits=None):
        if self.host_staged and out.is_cuda:
            o = torch.empty(out.shape, dtype=out.dtype)
            dist.all_to_all_single(o, inp.cpu(), out_splits, in_splits, group=self.group)
            out.copy_(o)
        else:
            dist.all_to_all_single(out, inp, out_splits, in_splits, group=self.group)

    # ---- plan / pack: HIP kernels for device tensors, the torch formulation for host tensors (gloo tests) ------------------
    def plan(self, topk_ids, capacity=0):
        """counts[G], seg_start[G+1], pos[M][G] (see sglk_ep_plan in include/sglk.h)."""
        G, E = self.world, self.num_experts
        if not topk_ids.is_cuda:
            counts, seg_start, pos, overflow = plan_torch(topk_ids, E, G, capacity)
            self.last_overflow = torch.tensor([overflow], dtype=torch.int32)
            return counts, seg_start, pos
        from . import _lib
        M, topk = topk_ids.shape
        dev = topk_ids.device
        counts = torch.empty(G, dtype=torch.int32, device=dev)
        seg_start = torch.empty(G + 1, dtype=torch.int32, device=dev)
        pos = torch.empty(M, G, dtype=torch.int32, device=dev)
        overflow = torch.empty(1, dtype=torch.int32, device=dev)
        _lib.check(_lib.lib().sglk_ep_plan(_ptr(topk_ids), M, topk, E, G, capacity, _ptr(counts), _ptr(seg_start), _ptr(pos),
                                           _ptr(overflow), ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "ep_plan")
        self.last_overflow = overflow
        return counts, seg_start, pos

    def pack(self, hidden, topk_weights, topk_ids, pos, seg_start, counts, rows, row_bytes, capacity=0):
        G, E = self.world, self.num_experts
        if not hidden.is_cuda:
            return pack_torch(hidden, topk_weights, topk_ids, pos, seg_start, counts, row_bytes, E, G, capacity)
        from . import _lib
        M, K = hidden.shape
        topk = topk_ids.shape[1]
        payload = torch.empty(rows, row_bytes, dtype=torch.uint8, device=hidden.device)
        _lib.check(_lib.lib().sglk_ep_pack(_ptr(hidden), hidden.stride(0), _ptr(topk_ids), _ptr(topk_weights), _ptr(pos),
                                           _ptr(seg_start), _ptr(counts), _ptr(payload), row_bytes, M, K, topk, E, G, capacity,
                                           ctypes.c_void_p(torch.cuda.current_stream(hidden.device).cuda_stream)), "ep_pack")
        return payload

    def _mark(self, evs):
        if self.profile and evs is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            evs.append(e)

    def __call__(self, hidden, topk_weights, topk_ids):
        G = self.world
        M, K = hidden.shape
        topk = topk_ids.shape[1]
        if hidden.stride(1) != 1:
            hidden = hidden.contiguous()
        topk_ids = topk_ids.to(torch.int32).contiguous()
        topk_weights = topk_weights.float().contiguous()
        row_bytes = (2 * K + 8 * topk + 15) // 16 * 16
        capacity = 0 if self.capacity_factor is None else max(1, math.ceil(self.capacity_factor * M))
        evs = [] if (self.profile and hidden.is_cuda) else None
        self._mark(evs)

        counts, seg_start, pos = self.plan(topk_ids, capacity)
        if capacity:
            send_l = recv_l = None                      # equal splits, nothing to read back
            rows_send = rows_recv = G * capacity
        else:
            both = torch.empty(2 * G, dtype=torch.int32, device=counts.device)
            both[:G] = counts
            self._all_to_all(both[G:], counts)
            both_l = both.tolist()                      # the ONE host sync of the exact mode: split sizes live on the host
            send_l, recv_l = both_l[:G], both_l[G:]
            rows_send, rows_recv = sum(send_l), sum(recv_l)
        payload = self.pack(hidden, topk_weights, topk_ids, pos, seg_start, counts, rows_send, row_bytes, capacity)
        self._mark(evs)

        recv = torch.empty(rows_recv, row_bytes, dtype=torch.uint8, device=hidden.device)
        self._all_to_all(recv, payload, recv_l, send_l)
        self._mark(evs)

        # the received token rows are used in place (row stride = row_bytes / 2 elements); routing is a small copy
        recv_rows = recv.view(torch.bfloat16)[:, :K] if recv.is_cuda else recv[:, :2 * K].contiguous().view(torch.bfloat16)
        meta = recv[:, 2 * K:2 * K + 8 * topk].contiguous().view(torch.int32)
        partial = self.local_fn(recv_rows, meta[:, topk:].contiguous().view(torch.float32), meta[:, :topk].contiguous())
        self._mark(evs)

        back = torch.empty(rows_send, K, dtype=hidden.dtype, device=hidden.device)
        self._all_to_all(back, partial.contiguous(), send_l, recv_l)
        out = self._reduce(back, pos, seg_start, M)
        self._mark(evs)
        if evs is not None:
            self._events.append(evs)
        self.last_stats = dict(rows_sent=int(rows_send), rows_received=int(rows_recv),
                               bytes_sent=int(rows_send) * row_bytes, bytes_returned=int(rows_recv) * K * hidden.element_size(),
                               capacity=capacity)
        return out

    def phase_ms(self):
        """Mean milliseconds per call of (plan+pack, dispatch all-to-all, local experts, return all-to-all + reduce); needs
        profile=True; synchronises."""
        if not self._events:
            return None
        torch.cuda.synchronize()
        names = ("plan_pack", "dispatch", "experts", "combine")
        tot = [0.0] * 4
        for evs in self._events:
            for i in range(4):
                tot[i] += evs[i].elapsed_time(evs[i + 1])
        n = len(self._events)
        self._events = []
        return {k: round(v / n, 4) for k, v in zip(names, tot)}

    def _reduce(self, back, pos, seg_start, M):
        """out[m] = sum over ranks d (ascending) of the row rank d returned for token m; fp32 sum, one rounding."""
        G, K = self.world, back.shape[1]
        table = torch.where(pos >= 0, pos + seg_start[:G].unsqueeze(0), torch.full_like(pos, -1)).contiguous()
        if back.is_cuda and back.dtype == torch.bfloat16 and K % 8 == 0:
            from . import _lib
            out = torch.empty(M, K, dtype=back.dtype, device=back.device)
            _lib.check(_lib.lib().sglk_ep_reduce_rows(back.data_ptr(), back.stride(0), table.data_ptr(), G, out.data_ptr(),
                                                      out.stride(0), M, K, torch.cuda.current_stream(back.device).cuda_stream),
                       "ep_reduce_rows")
            return out
        # host tensors (the gloo tests of the exchange logic): the same sums, same order, in torch
        out = torch.zeros(M, K, dtype=torch.float32, device=back.device)
        for d in range(G):
            sel = torch.nonzero(table[:, d] >= 0).flatten()
            if sel.numel():
                out.index_add_(0, sel, back[table[sel, d].long()].float())
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
