"""Pointer plumbing between numpy / torch buffers and the C ABI.

PyTorch-ROCm is used here only as the owner of device memory and streams.
"""

from __future__ import annotations

import numpy as np

from . import _lib


def is_torch(x) -> bool:
    return type(x).__module__.startswith("torch")


def on_device(x) -> bool:
    return is_torch(x) and x.is_cuda


def mem_flag(x) -> int:
    return _lib.MEM_DEVICE if on_device(x) else _lib.MEM_HOST


def ptr(x) -> int | None:
    """Raw address of a contiguous numpy array / torch tensor (None passes NULL)."""
    if x is None:
        return None
    if is_torch(x):
        if not x.is_contiguous():
            raise ValueError("tensor must be contiguous")
        return x.data_ptr()
    if isinstance(x, np.ndarray):
        if not x.flags["C_CONTIGUOUS"]:
            raise ValueError("array must be C-contiguous")
        return x.ctypes.data
    raise TypeError(f"unsupported buffer type {type(x)!r}")


def current_stream(x=None) -> int | None:
    """hipStream_t of torch's current stream on x's device (NULL stream for host buffers)."""
    if x is not None and on_device(x):
        import torch

        return torch.cuda.current_stream(x.device).cuda_stream
    return None


def same_side(*xs) -> int:
    """All non-None buffers must live on the same side of PCIe; returns the mem flag."""
    flags = {mem_flag(x) for x in xs if x is not None}
    if len(flags) > 1:
        raise ValueError("mixing host and device buffers in one call")
    return flags.pop() if flags else _lib.MEM_HOST
