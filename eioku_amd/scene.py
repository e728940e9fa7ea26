"""Scene-cut stage: HIP kernels K1/K2 plus the host arithmetic around them.

Mirrors what ``ModelManager.detect_scenes`` observes from its ffmpeg child process
(``/root/reference/ml-service/src/services/model_manager.py:736-828``) and the PySceneDetect
``ContentDetector`` that BASELINE.json's north_star names.  The per-frame integer sums come from
the GPU (``eioku_scene_sad_luma`` / ``eioku_scene_hsv_sums``); everything after them is a handful
of float64 operations per frame and stays on the host, in the order the libraries perform them.
"""

from __future__ import annotations

import numpy as np

from . import _lib
from ._buffers import current_stream, is_torch, on_device, ptr, same_side


def _out_like(ref, shape):
    if on_device(ref):
        import torch

        return torch.empty(shape, dtype=torch.int64, device=ref.device)
    return np.empty(shape, dtype=np.uint64)


def _finish(out, keep_on_device=False):
    """Device results come back as uint64 numpy (one small D2H copy, which synchronises) unless the
    caller keeps them in HBM (int64 tensor holding the uint64 bits) to stay asynchronous."""
    if is_torch(out):
        return out if keep_on_device else out.cpu().numpy().view(np.uint64)
    return out


def luma_sad(y_frames, prev=None, *, row_stride: int | None = None, frame_stride: int | None = None,
             shape: tuple[int, int, int] | None = None, keep_on_device: bool = False) -> np.ndarray:
    """K1: ``sad[t] = sum |Y_t - Y_{t-1}|`` (uint64, exact).  ``sad[0]`` is 0 without ``prev``.

    ``y_frames``: uint8 ``(n,h,w)`` numpy array (host, staged by the library) or CUDA tensor
    (HBM, zero copy).  ``row_stride`` / ``frame_stride`` / ``shape`` describe padded planes laid
    out in a flat buffer.
    """
    lib = _lib.load()
    _lib.init()
    if shape is None:
        n, h, w = (int(s) for s in y_frames.shape)
    else:
        n, h, w = shape
    row_stride = w if row_stride is None else int(row_stride)
    frame_stride = row_stride * h if frame_stride is None else int(frame_stride)
    mem = same_side(y_frames, prev)
    out = _out_like(y_frames, (n,))
    _lib.check(lib.eioku_scene_sad_luma(ptr(y_frames), n, h, w, row_stride, frame_stride, ptr(prev),
                                        ptr(out), mem, current_stream(y_frames)),
               "eioku_scene_sad_luma")
    return _finish(out, keep_on_device)


def hsv_sums(bgr_frames, prev=None, *, keep_on_device: bool = False) -> np.ndarray:
    """K2: per-frame ``sum |c_t - c_{t-1}|`` for c in (hue, sat, val); uint64 ``(n,3)``, exact."""
    lib = _lib.load()
    _lib.init()
    n, h, w, c = (int(s) for s in bgr_frames.shape)
    if c != 3:
        raise ValueError("expected (n,h,w,3) BGR frames")
    mem = same_side(bgr_frames, prev)
    out = _out_like(bgr_frames, (n, 3))
    _lib.check(lib.eioku_scene_hsv_sums(ptr(bgr_frames), n, h, w, h * w * 3, ptr(prev), ptr(out), mem,
                                        current_stream(bgr_frames)), "eioku_scene_hsv_sums")
    return _finish(out, keep_on_device)


def luma_sad_bgr(bgr_frames, prev=None, *, keep_on_device: bool = False) -> np.ndarray:
    """K1 on decoded BGR frames ``(n,h,w,3)``: OpenCV BT.601 luma of every pixel (``frames.bgr_to_luma_bt601``), then the
    per-frame SAD; uint64 ``(n,)``, exact.  ``prev``: the BGR frame preceding frame 0 (same side of PCIe)."""
    lib = _lib.load()
    _lib.init()
    n, h, w, c = (int(s) for s in bgr_frames.shape)
    if c != 3:
        raise ValueError("expected (n,h,w,3) BGR frames")
    mem = same_side(bgr_frames, prev)
    out = _out_like(bgr_frames, (n,))
    _lib.check(lib.eioku_scene_sad_luma_bgr(ptr(bgr_frames), n, h, w, h * w * 3, ptr(prev), ptr(out), mem,
                                            current_stream(bgr_frames)), "eioku_scene_sad_luma_bgr")
    return _finish(out, keep_on_device)


def scene_scores_luma(y_frames, prev=None, prev_mafd: float = 0.0):
    """K1 + libavfilter's score in one C call (``eioku_scene_scores_luma``): ``(mafd, score)`` float64 arrays for
    ``(n,h,w)`` uint8 luma planes (numpy: staged; CUDA tensor: zero copy).  Synchronises the stream."""
    lib = _lib.load()
    _lib.init()
    n, h, w = (int(s) for s in y_frames.shape)
    mafd = np.zeros(n, np.float64)
    score = np.zeros(n, np.float64)
    _lib.check(lib.eioku_scene_scores_luma(ptr(y_frames), n, h, w, w, h * w, ptr(prev), float(prev_mafd), ptr(mafd), ptr(score),
                                           same_side(y_frames, prev), current_stream(y_frames)), "eioku_scene_scores_luma")
    return mafd, score


def content_detect(bgr_frames, prev=None, threshold: float = 27.0, min_scene_len: int = 15, mode: str = "legacy"):
    """ContentDetector end to end in one C call (``eioku_scene_content``): ``(cut frame indices, scores)`` for
    ``(n,h,w,3)`` BGR frames.  Synchronises the stream."""
    import ctypes as C

    if mode not in ("legacy", "suppress", "merge"):
        raise ValueError(f"unknown mode {mode!r}")
    lib = _lib.load()
    _lib.init()
    n, h, w, c = (int(s) for s in bgr_frames.shape)
    if c != 3:
        raise ValueError("expected (n,h,w,3) BGR frames")
    cuts = np.zeros(max(n, 1), np.int32)
    scores = np.zeros(n, np.float64)
    found = C.c_int(0)
    _lib.check(lib.eioku_scene_content(ptr(bgr_frames), n, h, w, ptr(prev), float(threshold), int(min_scene_len),
                                       1 if mode == "merge" else 0, ptr(cuts), n, C.byref(found), ptr(scores),
                                       same_side(bgr_frames, prev), current_stream(bgr_frames)), "eioku_scene_content")
    return [int(t) for t in cuts[:found.value]], scores


def yuv420_to_bgr(yuv, h: int, w: int, layout: str = "i420"):
    """Decoder planes -> BGR frames with OpenCV's integer BT.601 (``eioku_yuv420_to_bgr``): ``yuv`` uint8
    ``(n, 3h/2, w)`` in OpenCV's planar Mat layout (numpy: staged; CUDA tensor: zero copy) -> ``(n,h,w,3)`` on the same
    side.  ``layout``: ``"i420"`` (Y | U | V) or ``"nv12"`` (Y | interleaved UV)."""
    lib = _lib.load()
    _lib.init()
    if layout not in ("i420", "nv12"):
        raise ValueError(f"unknown layout {layout!r}")
    n = int(yuv.shape[0])
    if tuple(int(v) for v in yuv.shape[1:]) != (h * 3 // 2, w):
        raise ValueError(f"expected (n, {h * 3 // 2}, {w}) planar frames, got {tuple(yuv.shape)}")
    if on_device(yuv):
        import torch

        out = torch.empty((n, h, w, 3), dtype=torch.uint8, device=yuv.device)
    else:
        yuv = np.ascontiguousarray(yuv, dtype=np.uint8)
        out = np.empty((n, h, w, 3), np.uint8)
    _lib.check(lib.eioku_yuv420_to_bgr(ptr(yuv), n, h, w, 1 if layout == "nv12" else 0, ptr(out), same_side(yuv), current_stream(yuv)),
               "eioku_yuv420_to_bgr")
    return out


def bgr2hsv(bgr):
    """OpenCV-compatible 8-bit BGR->HSV image (parity/debug helper)."""
    lib = _lib.load()
    _lib.init()
    if on_device(bgr):
        import torch

        out = torch.empty_like(bgr)
    else:
        out = np.empty_like(bgr)
    npix = int(np.prod(bgr.shape[:-1]))
    _lib.check(lib.eioku_bgr2hsv(ptr(bgr), npix, ptr(out), same_side(bgr), current_stream(bgr)),
               "eioku_bgr2hsv")
    return out


# ---------------------------------------------------------------------------
# host arithmetic (float64, library order)
# ---------------------------------------------------------------------------

def ffmpeg_scene_scores(sad: np.ndarray, count: int, *, bitdepth: int = 8, prev_mafd: float = 0.0,
                        first_has_prev: bool = False):
    """libavfilter ``get_scene_score`` on the SAD series -> ``(mafd, score)`` float64 arrays.

    ``mafd = sad / count / 2**(bitdepth-8)``; ``score = clip(float32(min(mafd, |mafd - prev|) / 100))``
    (the value passes through ``av_clipf``, i.e. is rounded to float32).  The arithmetic lives behind the C ABI
    (``eioku_scene_scores_from_sad``, csrc/scene_host.hip): one implementation for every binder.
    """
    sad = np.ascontiguousarray(sad, dtype=np.uint64)
    n = len(sad)
    mafd = np.zeros(n, np.float64)
    score = np.zeros(n, np.float64)
    _lib.check(_lib.load().eioku_scene_scores_from_sad(ptr(sad), n, float(count), int(bitdepth), float(prev_mafd),
                                                       int(bool(first_has_prev)), ptr(mafd), ptr(score)),
               "eioku_scene_scores_from_sad")
    return mafd, score


def pts_time_string(frame_index: int, tb_num: int, tb_den: int, pts_per_frame: int = 1) -> str:
    """What vf_showinfo prints after ``pts_time:`` - ``"%.6g" % (av_q2d(tb) * pts)``."""
    return "%.6g" % ((tb_num / float(tb_den)) * (frame_index * pts_per_frame))


def content_scores(sums: np.ndarray, num_pixels: int, *, first_has_prev: bool = False) -> np.ndarray:
    """ContentDetector frame score: ``(dh + ds + dl + 0.0) / 3.0`` with ``d = sum / float(pixels)``
    (``eioku_scene_content_scores``)."""
    sums = np.ascontiguousarray(sums, dtype=np.uint64).reshape(-1, 3)
    out = np.zeros(len(sums), np.float64)
    _lib.check(_lib.load().eioku_scene_content_scores(ptr(sums), len(sums), float(num_pixels), int(bool(first_has_prev)), ptr(out)),
               "eioku_scene_content_scores")
    return out


def content_cuts(scores, threshold: float = 27.0, min_scene_len: int = 15, mode: str = "legacy") -> list[int]:
    """Cut frames from ContentDetector scores (``eioku_scene_content_cuts``).

    ``legacy``: PySceneDetect 0.6.0-0.6.3 rule (``score >= threshold`` and ``min_scene_len`` frames
    since the last cut, counting from the first frame); ``suppress`` is the same rule under its
    0.6.4+ name; ``merge``: the 0.6.4+ ``FlashFilter.Mode.MERGE`` state machine.
    """
    import ctypes as C

    if mode not in ("legacy", "suppress", "merge"):
        raise ValueError(f"unknown mode {mode!r}")
    scores = np.ascontiguousarray(scores, dtype=np.float64)
    n = len(scores)
    cuts = np.zeros(max(n, 1), np.int32)
    found = C.c_int(0)
    _lib.check(_lib.load().eioku_scene_content_cuts(ptr(scores), n, float(threshold), int(min_scene_len), 1 if mode == "merge" else 0,
                                                    ptr(cuts), n, C.byref(found)), "eioku_scene_content_cuts")
    return [int(t) for t in cuts[:found.value]]


def build_scenes(cut_timestamps_ms, duration_ms: int | None) -> list[dict]:
    """The reference's scene list, index quirk included (``model_manager.py:758-828``).

    Cuts t1..tn give ``{i-1: t_i -> t_(i+1)}`` for i < n plus ``{n: tn -> duration}``; no cut gives
    ``{0: 0 -> duration}``.  ``duration_ms=None`` is the ffprobe-failure fallback (last cut + 1000).
    """
    ts = [int(t) for t in cut_timestamps_ms]
    scenes = [{"scene_index": i, "start_ms": a, "end_ms": b, "duration_ms": b - a}
              for i, (a, b) in enumerate(zip(ts[:-1], ts[1:]))]
    last = ts[-1] if ts else 0
    if duration_ms is None:
        duration_ms = last + 1000
    if ts:
        scenes.append({"scene_index": len(ts), "start_ms": last, "end_ms": duration_ms,
                       "duration_ms": duration_ms - last})
    else:
        scenes.append({"scene_index": 0, "start_ms": 0, "end_ms": duration_ms, "duration_ms": duration_ms})
    return scenes
