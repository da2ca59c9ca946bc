"""Tensor-parallel collectives of the reference harness (/root/reference/test_allreduce.py:86-132):
`initialize(world_size, rank)`, `shm_allreduce(tensor, group, op)` (in place) and `shm_allgather(tensor, group, dim)`.

The reference moves CPU tensors through a shared-memory segment between the ranks of one host.  Here every rank owns one
MI355X and the tensors live in HBM, so the same calls map to RCCL over xGMI through torch.distributed (backend "nccl"):
the 10 MiB bf16 message of the reference bench (1024 x 5120) is one all-reduce across the 7 point-to-point links.  A
"gloo" group works too - CPU tensors directly, GPU tensors staged through the host - which is what the CPU tests and
single-GPU rehearsals use.  No sglang process-group wrapper is needed: `group` is a torch.distributed group (or None for
the default one)."""
import torch
import torch.distributed as dist

_state = {"world_size": None, "rank": None}


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
    if _host_staged(tensor, group):
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
