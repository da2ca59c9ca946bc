"""Tensor-parallel collectives of the reference harness (/root/reference/test_allreduce.py:86-132):
`initialize(world_size, rank)`, `shm_allreduce(tensor, group, op)` (in place) and `shm_allgather(tensor, group, dim)`.

The reference moves CPU tensors through a shared-memory segment between the ranks of one host.  Here every rank owns one
MI355X and the tensors live in HBM.  bf16 sums go through XgmiAllReduce (below: direct peer reads over all 7 xGMI links,
one-shot / two-shot) once one is registered for the group; everything else maps to torch.distributed (backend "nccl" = RCCL).  A
"gloo" group works too - CPU tensors directly, GPU tensors staged through the host - which is what the CPU tests and
single-GPU rehearsals use.  No sglang process-group wrapper is needed: `group` is a torch.distributed group (or None for
the default one)."""
import torch
import torch.distributed as dist

_state = {"world_size": None, "rank": None}
_xgmi = {}      # group -> XgmiAllReduce registered for it (shm_allreduce uses it for bf16 SUM on GPU tensors)


class XgmiAllReduce:
    """Direct all-reduce between the GPUs of one node over xGMI peer memory (sglk_allreduce_sum_bf16, include/sglk.h): every
    rank maps its peers' staging regions through HIP IPC; one-shot for small messages, two-shot (reduce-scatter + all-gather of
    slices) for large ones, fp32 sums in ascending rank order -- bit-identical on every rank.  The 64-byte IPC handles travel
    through the torch.distributed group once, at construction; no collective of the group is used afterwards.

    Status: exercised with 2 and 4 processes SHARING one GPU (tests/test_xgmi_allreduce_gpu.py); NOT yet run across the GPUs of a
    node (no 8-GPU box in this build's reach), so the cross-GPU visibility of the staged data -- system-scope fences at the end of
    the producing kernels + sc0 sc1 loads in the consumers -- is unverified there.  Opt-in for that reason: nothing uses it
    unless the caller constructs one.

    Failure model: a rank that waits longer than SGLK_AR_WAIT_MS (default 30 s) for a peer gives up, leaves its output unwritten
    and sets a status word.  `all_reduce` / `shm_allreduce` read a host mirror of that word at the NEXT call (no stream
    synchronisation) and raise; `check()` synchronises and raises at once.  After a failure the ranks' epochs are out of step:
    every rank calls `resync()` (a collective on the group) before the communicator is used again."""

    def __init__(self, group=None, max_bytes=16 << 20, register=True):
        import ctypes
        from . import _lib
        self._lib, self._ct = _lib, ctypes
        L = _lib.lib()
        self.group = group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        if self.world > 8:
            raise RuntimeError("XgmiAllReduce: at most 8 ranks (one node)")
        self.cap = (int(max_bytes) + 255) // 256 * 256
        self.device = torch.device("cuda", torch.cuda.current_device())
        data, flags = ctypes.c_void_p(), ctypes.c_void_p()
        _lib.check(L.sglk_comm_alloc(4 * self.cap, 0, ctypes.byref(data)), "comm_alloc(data)")
        _lib.check(L.sglk_comm_alloc(256, 1, ctypes.byref(flags)), "comm_alloc(flags)")
        self._own = (data.value, flags.value)
        hd, hf = ctypes.create_string_buffer(64), ctypes.create_string_buffer(64)
        _lib.check(L.sglk_ipc_export(data, hd), "ipc_export(data)")
        _lib.check(L.sglk_ipc_export(flags, hf), "ipc_export(flags)")
        everyone = [None] * self.world
        dist.all_gather_object(everyone, (bytes(hd.raw), bytes(hf.raw)), group=group)
        self._opened = []
        pd, pf = (ctypes.c_void_p * self.world)(), (ctypes.c_void_p * self.world)()
        for r, (d, f) in enumerate(everyone):
            if r == self.rank:
                pd[r], pf[r] = data.value, flags.value
                continue
            od, of = ctypes.c_void_p(), ctypes.c_void_p()
            _lib.check(L.sglk_ipc_open(d, ctypes.byref(od)), f"ipc_open(data of rank {r})")
            _lib.check(L.sglk_ipc_open(f, ctypes.byref(of)), f"ipc_open(flags of rank {r})")
            pd[r], pf[r] = od.value, of.value
            self._opened += [od.value, of.value]
        self._pd, self._pf = pd, pf
        self.status = torch.zeros(1, dtype=torch.int32, device=self.device)
        self._host_status = torch.zeros(1, dtype=torch.int32).pin_memory()   # mirror, refreshed behind every call (async copy)
        self.epoch = 0
        dist.barrier(group=group)          # everybody has mapped everybody before the first call
        if register:
            _xgmi[group] = self

    def supports(self, t):
        """Decided from what every rank sees alike (device kind, dtype, element count): ranks must never split between this path
        and torch.distributed for one call.  Layout and alignment are local properties -- all_reduce stages such tensors."""
        return t.is_cuda and t.dtype == torch.bfloat16 and t.numel() % 8 == 0 and t.numel() * 2 <= self.cap

    def all_reduce(self, t, algo=0):
        """In-place sum over the ranks (bf16).  algo: 0 by size, 1 one-shot, 2 two-shot."""
        if not self.supports(t):
            raise RuntimeError(f"XgmiAllReduce: bf16 CUDA tensors of a multiple of 8 elements, at most {self.cap} bytes")
        if int(self._host_status[0]) != 0:      # an earlier call gave up on a peer (mirror of the device word; no synchronisation)
            raise RuntimeError("XgmiAllReduce: a peer did not arrive within SGLK_AR_WAIT_MS in an earlier call; that call's output "
                               "was not written.  Call resync() on every rank of the group before using the communicator again")
        work = t if (t.is_contiguous() and t.data_ptr() % 16 == 0) else t.contiguous().clone()   # clone(): a fresh, aligned block
        self.epoch += 1
        ct = self._ct
        stream = torch.cuda.current_stream(t.device)
        self._lib.check(self._lib.lib().sglk_allreduce_sum_bf16(
            self._pd, self._pf, self.rank, self.world, self.cap, ct.c_void_p(work.data_ptr()), ct.c_void_p(work.data_ptr()), work.numel(),
            self.epoch & 0xFFFFFFFF, algo, ct.c_void_p(self.status.data_ptr()), ct.c_void_p(stream.cuda_stream)), "allreduce_sum_bf16")
        self._host_status.copy_(self.status, non_blocking=True)
        if work is not t:
            t.copy_(work)
        return t

    def check(self):
        """Synchronises; raises if a peer failed to arrive in some call since the last check / resync."""
        if int(self.status.item()) != 0:
            raise RuntimeError("XgmiAllReduce: a peer did not arrive within SGLK_AR_WAIT_MS; call resync() on every rank")

    def resync(self):
        """Collective on the group, after a failed call: drains this rank's stream, moves every rank to one common epoch above
        all epochs in use (the flag words only ever grow, so nothing has to be cleared) and clears the status."""
        torch.cuda.synchronize()
        on_gpu = dist.get_backend(self.group) != "gloo"
        e = torch.tensor([self.epoch], dtype=torch.int64, device=self.device if on_gpu else "cpu")
        dist.all_reduce(e, op=dist.ReduceOp.MAX, group=self.group)
        self.epoch = int(e.item()) + 2
        self.status.zero_()
        self._host_status.zero_()
        torch.cuda.synchronize()
        dist.barrier(group=self.group)

    def close(self):
        """Collective: every rank unmaps its peers and frees its own regions.  The barrier keeps a fast rank from freeing memory
        a slower peer's kernels are still reading."""
        L = self._lib.lib()
        torch.cuda.synchronize()
        try:
            dist.barrier(group=self.group)
        except Exception:           # the group is already gone (interpreter shutdown): nothing left to wait for
            pass
        for p in self._opened:
            L.sglk_ipc_close(p)
        self._opened = []
        if self._own:
            L.sglk_comm_free(self._own[0])
            L.sglk_comm_free(self._own[1])
            self._own = None
        _xgmi.pop(self.group, None)


def initialize(world_size, rank):
    """/root/reference/test_allreduce.py:85-86.  The process group must exist already (the reference creates it with
    init_distributed_environment just before); this only checks that the numbers agree with it."""
    if not dist.is_initialized():
        raise RuntimeError("sgl_kernel.initialize: torch.distributed is not initialised (create the process group first)")
    if dist.get_world_size() != world_size or dist.get_rank() != rank:
        raise RuntimeError(f"sgl_kernel.initialize: (world_size, rank) = ({world_size}, {rank}) but the default group says "
                           f"({dist.get_world_size()}, {dist.get_rank()})")
    _state["world_size"], _state["rank"] = world_size, rank


def _host_staged(t, group):
    return t.is_cuda and dist.get_backend(group) == "gloo"


def shm_allreduce(tensor, group=None, op=dist.ReduceOp.SUM):
    """In-place all-reduce (/root/reference/test_allreduce.py:103-105).  Returns None like the reference."""
    if not tensor.is_contiguous():
        raise RuntimeError("shm_allreduce: tensor must be contiguous")
    comm = _xgmi.get(group)
    if comm is not None and op == dist.ReduceOp.SUM and comm.supports(tensor):
        comm.all_reduce(tensor)            # direct xGMI peer reads (XgmiAllReduce), not a ring
    elif _host_staged(tensor, group):
        h = tensor.cpu()
        dist.all_reduce(h, op=op, group=group)
        tensor.copy_(h)
    else:
        dist.all_reduce(tensor, op=op, group=group)


def shm_allgather(tensor, group=None, dim=0):
    """All-gather along `dim` (/root/reference/test_allreduce.py:125: get_tp_group().all_gather(tensor, dim)): the result
    has size world * tensor.size(dim) there, rank r's block at position r."""
    world = dist.get_world_size(group)
    if dim < 0:
        dim += tensor.dim()
    src = tensor.contiguous()
    staged = _host_staged(src, group)
    if staged:
        src = src.cpu()
    flat = torch.empty(world * src.numel(), dtype=src.dtype, device=src.device)
    dist.all_gather_into_tensor(flat, src.view(-1), group=group)
    out = flat.view((world,) + tuple(src.shape))
    out = out.movedim(0, dim).reshape(src.shape[:dim] + (world * src.shape[dim],) + src.shape[dim + 1:])
    return out.to(tensor.device) if staged else out
