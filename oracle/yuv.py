"""OpenCV's 8-bit YUV 4:2:0 -> BGR conversion (``cv2.COLOR_YUV2BGR_I420`` / ``COLOR_YUV2BGR_NV12``) in numpy.

Test infrastructure (see ``oracle/__init__.py``).  What it stands for: the colour conversion inside ``cap.read()`` of the
reference's frame loops (``/root/reference/ml-service/src/services/model_manager.py:237-297,331-398``: OpenCV's FFmpeg
backend decodes to YUV 4:2:0 and converts to BGR on the CPU before the detector sees the frame).  The single-pass
ingest uploads the decoder's planes (1.5 B / pixel instead of 3) and converts on the device.  **Parity unpinned**: cv2
is not installed here; the arithmetic restates OpenCV's ``color_yuv.simd.hpp`` [PUBLIC-LIB] - BT.601 studio range,
20-bit fixed point:

    y = max(0, Y - 16) * 1220542
    B = sat8((y + 2116026 (U - 128)                    + 2^19) >> 20)
    G = sat8((y -  409993 (U - 128) - 852492 (V - 128) + 2^19) >> 20)
    R = sat8((y + 1673527 (V - 128)                    + 2^19) >> 20)

every 2 x 2 block of pixels sharing one (U, V) sample - and is pinned by known answers only (tests/test_oracle_yuv.py:
studio black / white, the coefficients' float images 1.164 / 2.018 / 0.391 / 0.813 / 1.596, grey ramps).
"""
from __future__ import annotations

import numpy as np

CY, CUB, CUG, CVG, CVR, SHIFT = 1220542, 2116026, -409993, -852492, 1673527, 20


def split_planes(planar: np.ndarray, h: int, w: int, layout: str = "i420"):
    """planar: uint8 (..., 3h/2, w) in OpenCV's Mat layout -> (Y (...,h,w), U (...,h/2,w/2), V (...,h/2,w/2))"""
    y = planar[..., :h, :]
    c = planar[..., h:, :].reshape(*planar.shape[:-2], h // 2, w)
    if layout == "nv12":
        return y, c[..., 0::2], c[..., 1::2]
    if layout != "i420":
        raise ValueError(layout)
    flat = planar[..., h:, :].reshape(*planar.shape[:-2], -1)
    q = (h // 2) * (w // 2)
    u = flat[..., :q].reshape(*planar.shape[:-2], h // 2, w // 2)
    v = flat[..., q:2 * q].reshape(*planar.shape[:-2], h // 2, w // 2)
    return y, u, v


def yuv420_to_bgr(planar: np.ndarray, h: int, w: int, layout: str = "i420") -> np.ndarray:
    y, u, v = split_planes(planar, h, w, layout)
    yy = np.maximum(0, y.astype(np.int64) - 16) * CY
    uu = np.repeat(np.repeat(u.astype(np.int64) - 128, 2, -2), 2, -1)
    vv = np.repeat(np.repeat(v.astype(np.int64) - 128, 2, -2), 2, -1)
    half = 1 << (SHIFT - 1)
    b = (yy + CUB * uu + half) >> SHIFT
    g = (yy + CUG * uu + CVG * vv + half) >> SHIFT
    r = (yy + CVR * vv + half) >> SHIFT
    return np.clip(np.stack([b, g, r], -1), 0, 255).astype(np.uint8)


def bgr_to_i420(frames_bgr: np.ndarray) -> np.ndarray:
    """A plausible encoder for synthetic test clips (NOT an OpenCV restatement): BT.601 studio-range Y per pixel, U / V
    from the 2 x 2 block mean; (n, 3h/2, w) uint8 planar I420."""
    f = frames_bgr.astype(np.float64)
    n, h, w, _ = f.shape
    r, g, b = f[..., 2], f[..., 1], f[..., 0]
    y = 16 + (65.481 * r + 128.553 * g + 24.966 * b) / 255
    u = 128 + (-37.797 * r - 74.203 * g + 112.0 * b) / 255
    v = 128 + (112.0 * r - 93.786 * g - 18.214 * b) / 255
    sub = lambda p: p.reshape(n, h // 2, 2, w // 2, 2).mean((2, 4))
    out = np.empty((n, h * 3 // 2, w), np.uint8)
    out[:, :h] = np.clip(np.rint(y), 0, 255)
    q = (h // 2) * (w // 2)
    flat = out[:, h:].reshape(n, -1)
    flat[:, :q] = np.clip(np.rint(sub(u)), 0, 255).reshape(n, -1)
    flat[:, q:2 * q] = np.clip(np.rint(sub(v)), 0, 255).reshape(n, -1)
    return out


def i420_to_nv12(planar: np.ndarray, h: int, w: int) -> np.ndarray:
    y, u, v = split_planes(planar, h, w, "i420")
    out = planar.copy()
    c = out[..., h:, :].reshape(*planar.shape[:-2], h // 2, w)
    c[..., 0::2] = u
    c[..., 1::2] = v
    return out
