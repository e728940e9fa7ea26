"""Detection stage host side: letterbox planning, the YOLOv8 handle, result unpacking.

Replaces what ``model(frame, conf=thr, verbose=False, device=device)`` does inside
``ModelManager.detect_objects`` / ``detect_faces``
(``/root/reference/ml-service/src/services/model_manager.py:270-291, 364-392``): Ultralytics
predictor defaults apply because the reference passes nothing else - imgsz 640, rect letterbox
(stride 32, pad 114), iou 0.7, max_det 300, class-aware NMS.  All arithmetic on pixels runs in the HIP
library; this module only derives the handful of scalars / coefficient tables that Ultralytics and
OpenCV compute in Python-float / C-float arithmetic on the host, in the same order.
"""

from __future__ import annotations

import os
import ctypes as C
import functools
import math
from dataclasses import dataclass

import numpy as np

from . import _lib, weights as W
from ._buffers import current_stream, on_device, ptr

DET_DTYPE = np.dtype([("x1", "<f4"), ("y1", "<f4"), ("x2", "<f4"), ("y2", "<f4"), ("conf", "<f4"),
                      ("cls", "<i4"), ("anchor", "<i4"), ("pad", "<i4")])
assert DET_DTYPE.itemsize == 32

INTER_RESIZE_COEF_SCALE = 2048


@dataclass
class LetterboxPlan:
    src_h: int
    src_w: int
    new_h: int
    new_w: int
    top: int
    left: int
    out_h: int
    out_w: int
    mode: int  # 0 copy, 1 bilinear (cv2.INTER_LINEAR fixed point), 2 area 2x2
    gain: float  # scale_boxes gain (Python float)
    pad_x: int
    pad_y: int
    xofs: np.ndarray | None = None
    yofs: np.ndarray | None = None
    xalpha: np.ndarray | None = None
    ybeta: np.ndarray | None = None

    def geom(self) -> np.ndarray:
        return np.array([self.new_h, self.new_w, self.top, self.left, self.out_h, self.out_w, self.mode,
                         self.pad_x, self.pad_y], dtype=np.int32)


def _linear_coeffs(src: int, dst: int, clamp_edges: bool):
    """cv2.resize INTER_LINEAR tap position / 11-bit weights along one axis (imgproc/resize.cpp).

    ``scale = 1/(dst/src)`` in double; ``f = (float)((d+0.5)*scale - 0.5)``; ``s = floor(f)``;
    ``f -= s`` (float).  Horizontally (``clamp_edges``) taps that fall off either end snap to the
    edge pixel with weight 1; vertically the row index is clamped later and the weights are kept.
    """
    scale = 1.0 / (dst / src)
    ofs = np.empty(dst, dtype=np.int32)
    coef = np.empty((dst, 2), dtype=np.int16)
    for d in range(dst):
        f = np.float32((d + 0.5) * scale - 0.5)
        s = int(math.floor(float(f)))
        f = np.float32(f - np.float32(s))
        if clamp_edges:
            if s < 0:
                f, s = np.float32(0), 0
            if s >= src - 1:
                f, s = np.float32(0), src - 1
        ofs[d] = s
        coef[d, 0] = int(np.rint(np.float32(np.float32(1.0) - f) * np.float32(INTER_RESIZE_COEF_SCALE)))
        coef[d, 1] = int(np.rint(f * np.float32(INTER_RESIZE_COEF_SCALE)))
    return ofs, coef


@functools.lru_cache(maxsize=64)
def letterbox_plan(h: int, w: int, imgsz: int = 640, stride: int = 32, auto: bool = True) -> LetterboxPlan:
    """Ultralytics ``LetterBox(new_shape=imgsz, auto=True, stride=32)`` + ``scale_boxes`` geometry.

    Cached per frame size (the tap tables are a Python loop over every destination row and column); treat the
    returned plan and its arrays as read-only."""
    r = min(imgsz / h, imgsz / w)
    new_w, new_h = int(round(w * r)), int(round(h * r))
    dw, dh = imgsz - new_w, imgsz - new_h
    if auto:
        dw, dh = dw % stride, dh % stride
    dw /= 2
    dh /= 2
    top, bottom = int(round(dh - 0.1)), int(round(dh + 0.1))
    left, right = int(round(dw - 0.1)), int(round(dw + 0.1))
    out_h, out_w = new_h + top + bottom, new_w + left + right
    # ops.scale_boxes(img1_shape=(out_h,out_w), boxes, img0_shape=(h,w))
    gain = min(out_h / h, out_w / w)
    pad_x = round((out_w - w * gain) / 2 - 0.1)
    pad_y = round((out_h - h * gain) / 2 - 0.1)
    plan = LetterboxPlan(h, w, new_h, new_w, top, left, out_h, out_w, 0, gain, int(pad_x), int(pad_y))
    if (new_h, new_w) == (h, w):
        plan.mode = 0
    elif h == 2 * new_h and w == 2 * new_w:
        plan.mode = 2  # cv2.resize swaps INTER_LINEAR for INTER_AREA at an exact 1/2 scale
    else:
        plan.mode = 1
        plan.xofs, plan.xalpha = _linear_coeffs(w, new_w, True)
        plan.yofs, plan.ybeta = _linear_coeffs(h, new_h, False)
    return plan


class Yolov8Detector:
    """A YOLOv8 network resident on one GPU (``eioku_yolo_t``)."""

    def __init__(self, variant: str = "n", nc: int = 80, state: dict | None = None, names: dict | None = None):
        lib = _lib.load()
        _lib.init()
        self._lib = lib
        self.variant, self.nc = variant, nc
        self.names = names if names is not None else (dict(enumerate(W.COCO_NAMES)) if nc == 80 else
                                                      {i: str(i) for i in range(nc)})
        ch, depth = W.YOLO_VARIANTS[variant]
        h = C.c_void_p()
        _lib.check(lib.eioku_yolo_create((C.c_int * 5)(*ch), (C.c_int * 4)(*depth), nc, C.byref(h)),
                   "eioku_yolo_create")
        self._h = h
        self._table = self._conv_table()
        expect = [(n, co, (8 if n == "model.0.conv" else ci), k, s) for n, co, ci, k, s in W.conv_table(variant, nc)]
        if self._table != expect:
            raise RuntimeError("library graph and weights.conv_table disagree")
        if state is not None:
            self.load_state(state)

    @classmethod
    def from_model_name(cls, model_name: str, path=None, seed: int | None = None):
        """``yolov8s.pt`` / ``yolov8n-face.pt``: weights from ``path``, or random-init when ``seed`` is given."""
        variant, nc, names = W.variant_from_model_name(model_name)
        if seed is not None:
            state = W.random_state(variant, nc, seed)
        else:
            state, own = W.load_checkpoint(path, variant, nc)
            if own is not None and len(own) == nc:
                names = own  # result.names of THIS checkpoint (a custom-trained model), not the COCO table
        return cls(variant, nc, state, names)

    def _conv_table(self):
        lib, out = self._lib, []
        for i in range(lib.eioku_yolo_num_convs(self._h)):
            name = C.create_string_buffer(128)
            co, ci, k, s = C.c_int(), C.c_int(), C.c_int(), C.c_int()
            _lib.check(lib.eioku_yolo_conv_info(self._h, i, name, 128, C.byref(co), C.byref(ci), C.byref(k),
                                                C.byref(s)), "eioku_yolo_conv_info")
            out.append((name.value.decode(), co.value, ci.value, k.value, s.value))
        return out

    def load_state(self, state: dict) -> None:
        self._state = dict(state)
        for i, (name, cout, cin, k, _) in enumerate(self._table):
            w, b = state[name]
            w = np.asarray(w, dtype=np.float32)
            if name == "model.0.conv":  # RGB -> 8-channel pixels, zero weights on the padding channels
                w8 = np.zeros((cout, 8, k, k), dtype=np.float32)
                w8[:, :3] = w
                w = w8
            if w.shape != (cout, cin, k, k):
                raise ValueError(f"{name}: weight shape {w.shape} != {(cout, cin, k, k)}")
            w = np.ascontiguousarray(w)
            b = np.ascontiguousarray(b, dtype=np.float32)
            _lib.check(self._lib.eioku_yolo_set_conv(self._h, i, ptr(w), ptr(b)), f"eioku_yolo_set_conv({name})")

    def close(self):
        if getattr(self, "_h", None):
            self._lib.eioku_yolo_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- raw network (parity tests, bench) ------------------------------------------------------
    def forward_raw(self, x_nhwc8):
        """fp16 CUDA tensor (n,h,w,8) -> ([box P3,P4,P5], [cls P3,P4,P5]) fp32 CUDA tensors (NHWC)."""
        import torch

        n, h, w, c = (int(s) for s in x_nhwc8.shape)
        assert c == 8 and x_nhwc8.dtype == torch.float16
        box = [torch.empty((n, h // s, w // s, 64), dtype=torch.float32, device=x_nhwc8.device) for s in (8, 16, 32)]
        cls = [torch.empty((n, h // s, w // s, self.nc), dtype=torch.float32, device=x_nhwc8.device) for s in (8, 16, 32)]
        bp = (C.c_void_p * 3)(*[t.data_ptr() for t in box])
        cp = (C.c_void_p * 3)(*[t.data_ptr() for t in cls])
        _lib.check(self._lib.eioku_yolo_forward(self._h, ptr(x_nhwc8), n, h, w, bp, cp, current_stream(x_nhwc8)),
                   "eioku_yolo_forward")
        return box, cls

    def last_conv_flops(self) -> float:
        f = C.c_double(0)
        _lib.check(self._lib.eioku_yolo_last_conv_flops(self._h, C.byref(f)), "eioku_yolo_last_conv_flops")
        return f.value

    def calibrate_random_head(self, frames_bgr, frac: float = 0.01, conf: float = 0.25):
        """Random-init models only (bench): rescale the six Detect output convs so that, on these
        frames, box logits have std 2 and class logits std 3 with about ``frac`` of the anchors above
        ``conf`` - the candidate density of a trained detector, which sets the decode/NMS workload."""
        import torch

        x, _ = letterbox_f16(frames_bgr)
        box, cls = self.forward_raw(x)
        scaled = [(c - c.mean()) * (3.0 / c.std()) for c in cls]
        top = torch.cat([c.amax(dim=-1).reshape(-1) for c in scaled])
        shift = float(math.log(conf / (1 - conf)) - torch.quantile(top[:: max(1, top.numel() // 1_000_000)].float(), 1.0 - frac))
        for l in range(3):
            w, b = self._state[f"model.22.cv2.{l}.2"]
            self._state[f"model.22.cv2.{l}.2"] = ((w * (2.0 / float(box[l].std()))).astype(np.float32), b * 0)
            w, b = self._state[f"model.22.cv3.{l}.2"]
            sc = 3.0 / float(cls[l].std())
            self._state[f"model.22.cv3.{l}.2"] = ((w * sc).astype(np.float32),
                                                  ((b - float(cls[l].mean())) * sc + shift).astype(np.float32))
        self.load_state(self._state)

    # ---- frames -> detections ------------------------------------------------------------------
    def detect(self, frames_bgr, conf: float = 0.25, iou: float = 0.7, max_det: int = 300, imgsz: int = 640,
               keep_on_device: bool = False):
        """uint8 BGR frames ``(n,h,w,3)`` (numpy = host staging, CUDA tensor = zero copy).

        Returns ``(dets, counts)``: structured array ``(n,max_det)`` of :data:`DET_DTYPE` and int32 ``(n,)``;
        with ``keep_on_device`` the raw CUDA tensors (uint8 view / int32) so the call stays asynchronous.
        """
        n, h, w, c = (int(s) for s in frames_bgr.shape)
        if c != 3:
            raise ValueError("expected (n,h,w,3) BGR frames")
        plan = letterbox_plan(h, w, imgsz)
        dev = on_device(frames_bgr)
        if dev:
            import torch

            dets = torch.empty((n, max_det, 32), dtype=torch.uint8, device=frames_bgr.device)
            counts = torch.empty((n,), dtype=torch.int32, device=frames_bgr.device)
        else:
            dets = np.zeros((n, max_det), dtype=DET_DTYPE)
            counts = np.zeros((n,), dtype=np.int32)
        geom = plan.geom()
        _lib.check(self._lib.eioku_yolo_detect(
            self._h, ptr(frames_bgr), n, h, w, ptr(geom), ptr(plan.xofs), ptr(plan.yofs), ptr(plan.xalpha),
            ptr(plan.ybeta), float(np.float32(plan.gain)), float(conf), float(iou), int(max_det), ptr(dets),
            ptr(counts), _lib.MEM_DEVICE if dev else _lib.MEM_HOST, current_stream(frames_bgr)), "eioku_yolo_detect")
        if dev and not keep_on_device:
            dets = dets.cpu().numpy().view(DET_DTYPE).reshape(n, max_det)
            counts = counts.cpu().numpy()
        return dets, counts


class PipelinedDetector:
    """``depth`` detector handles with the same weights, each on its own HIP stream: ``submit`` enqueues a batch
    and returns at once, ``result`` hands back ``(dets, counts)`` in submission order.  Consecutive batches then
    overlap on the GPU (the network's many small launches leave a single stream a quarter idle) and the upload of
    batch i+1 overlaps the kernels of batch i; the frame loop of ``ModelManager`` and ``bench.py`` both run on it.
    """

    def __init__(self, first: Yolov8Detector, depth: int = 2, device=None, streams=None):
        import torch

        if depth < 1:
            raise ValueError("depth must be >= 1")
        self.names = first.names
        self._handles = [first]
        for _ in range(1, depth):
            d = Yolov8Detector(first.variant, first.nc, None, first.names)
            d.load_state(first._state)
            self._handles.append(d)
        self._device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        # high priority: the detector's persistent kernels size their grids for the whole chip; workgroups of other
        # streams' kernels (scene, embed) that sit on a CU when such a grid arrives turn into a straggler round
        # (measured on the overlapped bench step: 1.505 -> 1.46 ms)
        # `streams`: lanes handed in by a caller that builds one pipeline after another (a stream's hardware queue is
        # fixed when it is created; the lanes of a later pipeline can land on queues that make them share one with the
        # caller's other streams: bench.py's cfg4 runs measured 41 k instead of 58 k frames/s after the earlier runs)
        self._streams = list(streams) if streams is not None else [
            torch.cuda.Stream(device=self._device, priority=-1) for _ in self._handles]
        if len(self._streams) != len(self._handles):
            raise ValueError("one stream per handle")
        self._pending = []  # (dets, counts, event, frames kept alive)
        self._n = 0

    @property
    def depth(self) -> int:
        return len(self._handles)

    def in_flight(self) -> int:
        return len(self._pending)

    def submit(self, frames_bgr, conf: float = 0.25, iou: float = 0.7, max_det: int = 300, imgsz: int = 640) -> None:
        import torch

        j = self._n % len(self._handles)
        self._n += 1
        s = self._streams[j]
        if on_device(frames_bgr):
            s.wait_stream(torch.cuda.current_stream(self._device))  # the caller produced the frames on its stream
        with torch.cuda.stream(s):
            f = frames_bgr if on_device(frames_bgr) else torch.from_numpy(np.ascontiguousarray(frames_bgr)).to(
                self._device, non_blocking=True)
            dets, counts = self._handles[j].detect(f, conf=conf, iou=iou, max_det=max_det, imgsz=imgsz, keep_on_device=True)
            ev = torch.cuda.Event()
            ev.record(s)
        self._pending.append((dets, counts, ev, f))

    def result(self):
        """Oldest submitted batch: ``(dets structured (n,max_det), counts int32 (n,))`` on the host."""
        dets, counts, ev, _ = self._pending.pop(0)
        ev.synchronize()
        n, max_det = int(dets.shape[0]), int(dets.shape[1])
        return dets.cpu().numpy().view(DET_DTYPE).reshape(n, max_det), counts.cpu().numpy()

    def result_on_device(self):
        """Oldest submitted batch as the raw CUDA tensors plus the event that marks them complete."""
        dets, counts, ev, _ = self._pending.pop(0)
        return dets, counts, ev

    def close(self):
        import torch

        if self._pending:
            torch.cuda.synchronize(self._device)
            self._pending.clear()
        for h in self._handles:
            h.close()


def letterbox_f16(frames_bgr, imgsz: int = 640):
    """K3 alone: CUDA uint8 (n,h,w,3) -> CUDA fp16 (n,out_h,out_w,8) network input (RGB/255 + 5 zero channels)."""
    import torch

    lib = _lib.load()
    _lib.init()
    n, h, w, _ = (int(s) for s in frames_bgr.shape)
    plan = letterbox_plan(h, w, imgsz)
    out = torch.empty((n, plan.out_h, plan.out_w, 8), dtype=torch.float16, device=frames_bgr.device)
    geom = plan.geom()
    _lib.check(lib.eioku_letterbox_f16(ptr(frames_bgr), n, h, w, ptr(geom), ptr(plan.xofs), ptr(plan.yofs),
                                       ptr(plan.xalpha), ptr(plan.ybeta), ptr(out), current_stream(frames_bgr)),
               "eioku_letterbox_f16")
    return out, plan


def postprocess(box_maps, cls_maps, plan: LetterboxPlan, conf: float, iou: float = 0.7, max_det: int = 300):
    """K6+K7 alone on CUDA fp32 Detect maps (NHWC) -> (dets structured array, counts)."""
    import torch

    lib = _lib.load()
    _lib.init()
    n = int(box_maps[0].shape[0])
    nc = int(cls_maps[0].shape[-1])
    hl = (C.c_int * 3)(*[int(b.shape[1]) for b in box_maps])
    wl = (C.c_int * 3)(*[int(b.shape[2]) for b in box_maps])
    dev = box_maps[0].device
    dets = torch.empty((n, max_det, 32), dtype=torch.uint8, device=dev)
    counts = torch.empty((n,), dtype=torch.int32, device=dev)
    bp = (C.c_void_p * 3)(*[t.data_ptr() for t in box_maps])
    cp = (C.c_void_p * 3)(*[t.data_ptr() for t in cls_maps])
    _lib.check(lib.eioku_yolo_postprocess(bp, cp, n, hl, wl, nc, float(conf), float(iou), int(max_det),
                                          float(np.float32(plan.gain)), plan.pad_x, plan.pad_y, plan.src_w, plan.src_h,
                                          ptr(dets), ptr(counts), current_stream(box_maps[0])), "eioku_yolo_postprocess")
    return dets.cpu().numpy().view(DET_DTYPE).reshape(n, max_det), counts.cpu().numpy()
