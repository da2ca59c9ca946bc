"""sgl_kernel for MI355X (gfx950): the reference harness's `import sgl_kernel` resolves to this package.

Importing it registers the `torch.ops.sgl_kernel.*` operators with the reference's exact signatures
(/root/reference/bench_moe.py:5-6 etc.) and forwards each call through ctypes to hand-written HIP kernels in
libsglk.so (C-ABI: include/sglk.h).  There is no CPU compute path: operators given CPU tensors stage them
through the GPU (host buffers in, host buffers out) and raise if no GPU is present.
"""
from . import _lib  # noqa: F401
from . import _ops  # noqa: F401  (registers torch.ops.sgl_kernel.*)
from . import common_ops  # noqa: F401
from .version import __version__  # noqa: F401
