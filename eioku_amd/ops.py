"""Thin wrappers over single-kernel C-ABI entry points (building blocks; used by the parity tests)."""
from __future__ import annotations

import numpy as np

from . import _lib
from ._buffers import current_stream, ptr


def conv2d_f16(x_nhwc, weight_oihw: np.ndarray, bias: np.ndarray | None, *, stride: int = 1, silu: bool = True,
               in_coff: int = 0, cin: int | None = None, residual=None, res_coff: int = 0,
               out=None, out_coff: int = 0, out_f32: bool = False):
    """K4: one NHWC fp16 convolution (k in {1,3}, pad k//2) with fused bias/SiLU/residual.

    ``x_nhwc``: CUDA fp16 tensor (n,h,w,C); the conv reads channels [in_coff, in_coff+cin).
    ``out``: optional preallocated CUDA fp16 tensor (n,ho,wo,C_out_total) written at ``out_coff``.
    """
    import torch

    lib = _lib.load()
    _lib.init()
    n, h, w, ctot = (int(s) for s in x_nhwc.shape)
    cout, cin_w, k, _ = weight_oihw.shape
    cin = cin_w if cin is None else cin
    assert cin == cin_w
    w32 = np.ascontiguousarray(weight_oihw, dtype=np.float32)
    b32 = None if bias is None else np.ascontiguousarray(bias, dtype=np.float32)
    ho = (h + 2 * (k // 2) - k) // stride + 1
    wo = (w + 2 * (k // 2) - k) // stride + 1
    o32 = None
    if out_f32:
        o32 = torch.empty((n, ho, wo, cout), dtype=torch.float32, device=x_nhwc.device)
    elif out is None:
        out = torch.empty((n, ho, wo, cout), dtype=torch.float16, device=x_nhwc.device)
    _lib.check(lib.eioku_conv2d_f16(ptr(x_nhwc), n, h, w, ctot, in_coff, cin, ptr(w32), ptr(b32), cout, k, stride,
                                    int(silu), ptr(residual), 0 if residual is None else int(residual.shape[-1]),
                                    res_coff, ptr(out), 0 if out is None else int(out.shape[-1]), out_coff,
                                    ptr(o32), current_stream(x_nhwc)), "eioku_conv2d_f16")
    return o32 if out_f32 else out
