"""Synthetic bench/test inputs generated directly in HBM.

Counter-based splitmix64: element ``i`` of stream ``seed`` is ``mix(seed + (i+1)*GOLDEN)``.
The host-side helpers below only build the tiny per-frame parameter tables; the bulk data is
produced by the kernels in ``csrc/synth.hip``.
"""

from __future__ import annotations

import numpy as np

from . import _lib
from ._buffers import current_stream, ptr

_MASK = (1 << 64) - 1


def _mix(z: int) -> int:
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _MASK
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _MASK
    return z ^ (z >> 31)


def splitmix64_scalar(seed: int, i: int) -> int:
    return _mix((seed + (i + 1) * 0x9E3779B97F4A7C15) & _MASK)


def frame_params(seed: int, n_frames: int, first_frame: int = 0, mean_len: int = 150, jitter: int = 60) -> np.ndarray:
    """(n,5) int32 ``[base_b, base_g, base_r, gx, gy]`` per frame; a new scene every 150+U{0..59} frames."""
    total = first_frame + n_frames
    out = np.empty((total, 5), dtype=np.int32)
    t = s = 0
    while t < total:
        ln = mean_len + splitmix64_scalar(seed ^ 0x5CE2E, s) % max(jitter, 1)
        r = splitmix64_scalar(seed ^ 0xBA5E, s)
        row = [32 + (r & 0xFF) % 192, 32 + ((r >> 8) & 0xFF) % 192, 32 + ((r >> 16) & 0xFF) % 192,
               (r >> 24) & 31, (r >> 32) & 31]
        out[t:t + ln] = row
        t += ln
        s += 1
    return out[first_frame:]


def frames_bgr(seed: int, n: int, h: int, w: int, device, first_frame: int = 0, params: np.ndarray | None = None):
    """uint8 CUDA tensor (n,h,w,3) of synthetic BGR frames."""
    import torch

    lib = _lib.load()
    _lib.init()
    if params is None:
        params = frame_params(seed, n, first_frame)
    p = torch.from_numpy(np.ascontiguousarray(params, dtype=np.int32)).to(device)
    out = torch.empty((n, h, w, 3), dtype=torch.uint8, device=device)
    _lib.check(lib.eioku_synth_frames_bgr(seed, first_frame, n, h, w, ptr(p), ptr(out), current_stream(out)),
               "eioku_synth_frames_bgr")
    return out


def normal_f32(seed: int, rows: int, dim: int, device, l2_normalise: bool = False):
    """float32 CUDA tensor (rows,dim) of approx-normal values (Irwin-Hall 4), optionally unit rows."""
    import torch

    lib = _lib.load()
    _lib.init()
    out = torch.empty((rows, dim), dtype=torch.float32, device=device)
    _lib.check(lib.eioku_synth_normal_f32(seed, rows, dim, int(l2_normalise), ptr(out), current_stream(out)),
               "eioku_synth_normal_f32")
    return out


def u64(seed: int, n: int, device, offset: int = 0):
    import torch

    lib = _lib.load()
    _lib.init()
    out = torch.empty((n,), dtype=torch.int64, device=device)
    _lib.check(lib.eioku_synth_u64(seed, offset, n, ptr(out), current_stream(out)), "eioku_synth_u64")
    return out


def bytes_u8(seed: int, n: int, device):
    import torch

    lib = _lib.load()
    _lib.init()
    out = torch.empty((n,), dtype=torch.uint8, device=device)
    _lib.check(lib.eioku_synth_bytes(seed, n, ptr(out), current_stream(out)), "eioku_synth_bytes")
    return out
