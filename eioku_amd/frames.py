"""Frame sources for the drop-in ``ModelManager`` (the ``cv2.VideoCapture`` seam).

The reference reads frames with ``cv2.VideoCapture(video_path)`` - ``get(CAP_PROP_FPS)``,
``get(CAP_PROP_FRAME_COUNT)``, ``read()``, ``grab()``, ``release()``
(``/root/reference/ml-service/src/services/model_manager.py:237-299``) - and scene detection shells
out to ffmpeg.  Video *decode* is outside this round's scope, so this module
only provides the same small surface over what is available:

  * ``cv2`` itself when the deployment image has it (the reference's container does): BGR frames for the
    detectors and the decoder's luma plane for the scene stage (``Cv2FrameSource.luma_planes``);
  * ``.npy`` raw clips ``(n,h,w,3)`` uint8 BGR, memory mapped, with an optional ``<file>.json``
    sidecar ``{"fps": 29.97, "time_base": [1001, 30000], "duration": 6.673}``;
  * ``.y4m`` (YUV4MPEG2, 4:2:0 / 4:4:4 / mono, 8 bit) for the scene stage: exactly the luma plane the
    ffmpeg ``select`` filter scores.
"""

from __future__ import annotations

import json
from fractions import Fraction
from pathlib import Path

import numpy as np


class EndOfStream(RuntimeError):
    """A container's header promised more frames than its stream holds (CAP_PROP_FRAME_COUNT is an estimate)."""


class FrameSource:
    """Minimal ``cv2.VideoCapture`` look-alike."""

    fps: float = 30.0
    total_frames: int = 0
    time_base: tuple[int, int] = (1, 30)  # seconds per pts tick (vf_showinfo's pts_time = pts * tb)
    duration_s: float | None = None

    def read(self):  # -> (ok, frame_bgr)
        raise NotImplementedError

    def grab(self) -> bool:
        raise NotImplementedError

    def release(self) -> None:
        pass

    def luma_planes(self, start: int, count: int) -> np.ndarray:
        """``(count,h,w)`` uint8 luma of frames [start, start+count) for the scene stage."""
        raise NotImplementedError

    # Decoder planes: a source that can hand over the decoded 4:2:0 frame itself - 1.5 bytes per
    # pixel - names its layout here ("i420" / "nv12") and implements read_yuv(); the single-pass ingest then uploads the
    # planes and converts to BGR on the device (eioku_yuv420_to_bgr).  None: BGR only.
    yuv_layout: str | None = None

    def read_yuv(self):  # -> (ok, planar uint8 (3h/2, w) frame in OpenCV's Mat layout)
        raise NotImplementedError


def _fps_to_time_base(fps: float) -> tuple[int, int]:
    fr = Fraction(fps).limit_denominator(1001)
    return fr.denominator, fr.numerator


class NpyFrameSource(FrameSource):
    def __init__(self, path):
        self.path = Path(path)
        self.frames = np.load(self.path, mmap_mode="r")
        if self.frames.ndim != 4 or self.frames.shape[-1] != 3 or self.frames.dtype != np.uint8:
            raise ValueError(f"{path}: expected (n,h,w,3) uint8 BGR frames")
        meta = {}
        side = Path(str(self.path) + ".json")
        if side.exists():
            meta = json.loads(side.read_text())
        self.fps = float(meta.get("fps", 30.0))
        self.total_frames = int(self.frames.shape[0])
        self.time_base = tuple(meta.get("time_base", _fps_to_time_base(self.fps)))
        self.duration_s = meta.get("duration", self.total_frames / self.fps if self.fps else None)
        self.pos = 0

    def read(self):
        if self.pos >= self.total_frames:
            return False, None
        f = np.ascontiguousarray(self.frames[self.pos])
        self.pos += 1
        return True, f

    def grab(self):
        if self.pos >= self.total_frames:
            return False
        self.pos += 1
        return True

    def luma_planes(self, start, count):
        # raw BGR has no Y plane: OpenCV's COLOR_BGR2YUV_I420 luma stands in for the decoder's (one definition for every
        # BGR source, host and device: eioku_scene_sad_luma_bgr computes the same integers)
        return bgr_to_luma_bt601(np.asarray(self.frames[start:start + count]))


class Y4mSource(FrameSource):
    """YUV4MPEG2 reader (8-bit C420*, C444, Cmono): luma plane access for the scene stage."""

    def __init__(self, path):
        self.path = Path(path)
        with open(self.path, "rb") as f:
            header = f.readline()
        if not header.startswith(b"YUV4MPEG2"):
            raise ValueError(f"{path}: not a YUV4MPEG2 file")
        w = h = None
        fr = Fraction(30, 1)
        cs = "420"
        for tok in header.split()[1:]:
            t = tok.decode()
            if t[0] == "W":
                w = int(t[1:])
            elif t[0] == "H":
                h = int(t[1:])
            elif t[0] == "F":
                a, b = t[1:].split(":")
                fr = Fraction(int(a), int(b))
            elif t[0] == "C":
                cs = t[1:]
        if "p10" in cs or "p12" in cs or "p16" in cs:
            raise ValueError("only 8-bit y4m is supported")
        self.w, self.h = w, h
        cw, chh = (w, h) if cs.startswith("444") else ((0, 0) if cs.startswith("mono") else ((w + 1) // 2, (h + 1) // 2))
        self.frame_bytes = w * h + 2 * cw * chh
        self.header_len = len(header)
        size = self.path.stat().st_size
        self.total_frames = (size - self.header_len) // (6 + self.frame_bytes)  # "FRAME\n" + payload
        self.fps = float(fr)
        self.time_base = (fr.denominator, fr.numerator)
        self.duration_s = self.total_frames / self.fps
        self.pos = 0
        self._mm = np.memmap(self.path, dtype=np.uint8, mode="r")
        # 4:2:0 clips with even sizes hand their planes to the single-pass ingest as they are (I420)
        if cs.startswith("420") and w % 4 == 0 and h % 2 == 0:
            self.yuv_layout = "i420"

    def read_yuv(self):
        if self.yuv_layout is None:
            raise RuntimeError(f"{self.path}: only 4:2:0 y4m clips (w % 4 == 0, h % 2 == 0) carry planes the device converts")
        if self.pos >= self.total_frames:
            return False, None
        off = self.header_len + self.pos * (6 + self.frame_bytes) + 6
        self.pos += 1
        return True, np.asarray(self._mm[off:off + self.frame_bytes]).reshape(self.h * 3 // 2, self.w)

    def luma_planes(self, start, count):
        out = np.empty((count, self.h, self.w), dtype=np.uint8)
        for i in range(count):
            off = self.header_len + (start + i) * (6 + self.frame_bytes) + 6
            out[i] = self._mm[off:off + self.w * self.h].reshape(self.h, self.w)
        return out

    def grab(self):
        if self.pos >= self.total_frames:
            return False
        self.pos += 1
        return True

    def read(self):
        raise RuntimeError("y4m sources carry no BGR frames here; use them for scene detection")


def bgr_to_luma_bt601(frames_bgr: np.ndarray) -> np.ndarray:
    """OpenCV's 8-bit ``COLOR_BGR2YUV_I420`` luma (``RGB2YUV420p``: BT.601 studio range, 20-bit fixed
    point, ``Y = (269484 R + 528482 G + 102760 B + (16 << 20) + (1 << 19)) >> 20``) [PUBLIC-LIB]."""
    f = frames_bgr.astype(np.int64)
    y = (269484 * f[..., 2] + 528482 * f[..., 1] + 102760 * f[..., 0] + (16 << 20) + (1 << 19)) >> 20
    return y.astype(np.uint8)


class Cv2FrameSource(FrameSource):
    """``cv2.VideoCapture`` (what the reference opens, ``model_manager.py:237``).

    ``luma_planes`` serves the scene stage, which in the reference is an ffmpeg child scoring the DECODER's
    luma plane (``model_manager.py:736-755``).  A second capture of the same file is opened for it:

    * with ``CAP_PROP_CONVERT_RGB = 0`` OpenCV's FFmpeg backend hands back the decoded frame without the
      swscale BGR conversion - a single-channel ``(h, w)`` plane or a planar I420 ``(3h/2, w)`` image; the first
      ``h`` rows are exactly the Y plane ffmpeg scores (bit-exact with the reference's scene score);
    * a backend that ignores the property returns BGR; the luma is then recomputed with OpenCV's own
      ``COLOR_BGR2YUV_I420`` integer formula.  YUV -> BGR -> Y does not round-trip exactly (+-1-2 codes on some
      pixels), so the scene SCORE can differ from ffmpeg's in the last digits; cut decisions at the default
      threshold (a 70-code mean difference) are far from that noise.  ``luma_exact`` says which case applies.
    """

    def __init__(self, path):
        import cv2

        self._cv2 = cv2
        self.path = path
        self.cap = cv2.VideoCapture(path)
        self.fps = self.cap.get(cv2.CAP_PROP_FPS) or 30
        self.total_frames = int(self.cap.get(cv2.CAP_PROP_FRAME_COUNT))
        self.time_base = _fps_to_time_base(self.fps)
        self.duration_s = self.total_frames / self.fps if self.fps else None
        self._ycap = None
        self._ypos = 0
        self.luma_exact = None  # decided by the first luma frame

    def read(self):
        return self.cap.read()

    def grab(self):
        return self.cap.grab()

    def release(self):
        self.cap.release()
        if self._ycap is not None:
            self._ycap.release()
            self._ycap = None

    # ---- decoder planes for the single-pass ingest ---------------------------------------------------------
    def try_yuv(self) -> bool:
        """Re-open the capture with ``CAP_PROP_CONVERT_RGB = 0``; when the backend answers with planar 4:2:0 frames
        ``(3h/2, w)`` (OpenCV's FFmpeg backend does for yuv420p streams) those are what ``read_yuv`` returns from now on
        and ``yuv_layout`` names their layout (``EIOKU_CV2_RAW_LAYOUT``, default ``i420``: a shape cannot tell I420 from
        NV12).  False - and nothing changed - when the backend ignores the property or hands back something else."""
        import os

        cv2 = self._cv2
        cap = cv2.VideoCapture(self.path)
        cap.set(cv2.CAP_PROP_CONVERT_RGB, 0)
        hh = int(cap.get(cv2.CAP_PROP_FRAME_HEIGHT))
        ok, f = cap.read()
        cap.release()
        f = np.asarray(f) if ok else None
        if f is not None and f.ndim == 3 and f.shape[2] == 1:
            f = f.reshape(f.shape[0], f.shape[1])
        if f is None or f.ndim != 2 or hh <= 0 or f.shape[0] != hh * 3 // 2 or hh % 2 or f.shape[1] % 4:
            return False
        self.cap.release()
        self.cap = cv2.VideoCapture(self.path)
        self.cap.set(cv2.CAP_PROP_CONVERT_RGB, 0)
        self.yuv_layout = os.environ.get("EIOKU_CV2_RAW_LAYOUT", "i420")
        self._yuv_h = hh
        return True

    def read_yuv(self):
        if self.yuv_layout is None:
            raise RuntimeError("try_yuv() first: this capture hands back BGR")
        ok, f = self.cap.read()
        if not ok:
            return False, None
        f = np.asarray(f)
        return True, f.reshape(f.shape[0], f.shape[1])

    def _luma_of(self, frame) -> np.ndarray:
        f = np.asarray(frame)
        if f.ndim == 2 or (f.ndim == 3 and f.shape[2] == 1):
            f = f.reshape(f.shape[0], f.shape[1])
            if self._h is None:
                # (h, w) gray plane or (3h/2, w) planar 4:2:0: the capture's own frame height tells them apart
                hh = int(self._ycap.get(self._cv2.CAP_PROP_FRAME_HEIGHT)) or f.shape[0]
                self._h = hh if f.shape[0] in (hh, hh * 3 // 2) else f.shape[0]
            self.luma_exact = True
            return np.ascontiguousarray(f[: self._h])
        self.luma_exact = False
        return bgr_to_luma_bt601(f)

    def luma_planes(self, start, count):
        cv2 = self._cv2
        if self._ycap is None or start < self._ypos:
            if self._ycap is not None:
                self._ycap.release()
            self._ycap = cv2.VideoCapture(self.path)
            self._ycap.set(cv2.CAP_PROP_CONVERT_RGB, 0)
            self._ypos = 0
            self._h = None
        while self._ypos < start:  # sequential access is the only access a compressed stream offers
            if not self._ycap.grab():
                break
            self._ypos += 1
        out = []
        for _ in range(count):
            ok, frame = self._ycap.read()
            if not ok:
                break
            self._ypos += 1
            out.append(self._luma_of(frame))
        if not out:
            raise EndOfStream(f"{self.path!r}: no frame at index {start}")
        return np.stack(out)


def open_video(path: str) -> FrameSource:
    p = str(path)
    if p.endswith(".npy"):
        return NpyFrameSource(p)
    if p.endswith(".y4m"):
        return Y4mSource(p)
    try:
        return Cv2FrameSource(p)
    except ImportError as e:
        raise RuntimeError(f"cannot open {p!r}: no decoder available (cv2 not installed; raw .npy / .y4m clips are "
                           "supported natively)") from e
