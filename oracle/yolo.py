"""Detection stage oracle (rows a2 / a3 of SURVEY.md §8): what ``model(frame, conf=...)`` computes.

Test infrastructure (see ``oracle/__init__.py``).  **Parity unpinned** for the arithmetic: cv2,
ultralytics and torchvision are not installed in the build container and no YOLOv8 checkpoint is
available, so the algorithms of the pinned versions (opencv-python 4.11.0.86, ultralytics 8.4.8,
torchvision 0.16.2 - ``ml-service/poetry.lock:1999,3865,3730``) are restated from their published
source [PUBLIC-LIB].  The orchestration around the call (sampling, timestamps, dict shape) is pinned
separately by ``tests/golden/ref_detect_loop.json``.

The network oracle is *the same fp16 network* the HIP path runs (BASELINE cfg2 is fp16): weights and
every stored activation rounded to fp16 (RNE), accumulation / bias / SiLU in fp32 on torch-CPU, the
three Detect output convs kept in fp32.  The reference itself runs fp32 (``half=False``); DESIGN.md
records the expected drift between the two.
"""

from __future__ import annotations

import math

import numpy as np

# --------------------------------------------------------------------------------------------
# letterbox: cv2.resize(INTER_LINEAR) + copyMakeBorder(114)   [PUBLIC-LIB: OpenCV resize.cpp]
# --------------------------------------------------------------------------------------------


def _axis_coeffs(src: int, dst: int, horizontal: bool):
    scale = 1.0 / (float(dst) / float(src))
    ofs, a0, a1 = [], [], []
    for d in range(dst):
        f = np.float32((d + 0.5) * scale - 0.5)
        s = math.floor(float(f))
        f = np.float32(f - np.float32(s))
        if horizontal:
            if s < 0:
                f, s = np.float32(0.0), 0
            if s >= src - 1:
                f, s = np.float32(0.0), src - 1
        ofs.append(s)
        # saturate_cast<short>(float * 2048) == cvRound (round half to even)
        a0.append(int(np.rint(np.float32((np.float32(1.0) - f) * np.float32(2048.0)))))
        a1.append(int(np.rint(np.float32(f * np.float32(2048.0)))))
    return np.array(ofs), np.array(a0, dtype=np.int64), np.array(a1, dtype=np.int64)


def resize_linear_u8(src: np.ndarray, dst_w: int, dst_h: int) -> np.ndarray:
    """``cv2.resize(src, (dst_w, dst_h), interpolation=cv2.INTER_LINEAR)`` for uint8 HxWxC."""
    sh, sw = src.shape[:2]
    if (sh, sw) == (dst_h, dst_w):
        return src.copy()
    if sw == 2 * dst_w and sh == 2 * dst_h:
        # resize(): INTER_LINEAR with iscale_x == iscale_y == 2 is replaced by INTER_AREA
        s = src.astype(np.int32)
        return ((s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    xo, xa0, xa1 = _axis_coeffs(sw, dst_w, True)
    yo, yb0, yb1 = _axis_coeffs(sh, dst_h, False)
    s = src.astype(np.int64)
    x1 = np.minimum(xo + 1, sw - 1)
    # HResizeLinear<uchar,int,short>: D = S[sx]*a0 + S[sx+cn]*a1
    hrow = s[:, xo, :] * xa0[None, :, None] + s[:, x1, :] * xa1[None, :, None]
    r0 = np.clip(yo, 0, sh - 1)
    r1 = np.clip(yo + 1, 0, sh - 1)
    s0, s1 = hrow[r0], hrow[r1]
    # VResizeLinear<uchar,...>: ((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2
    out = (((yb0[:, None, None] * (s0 >> 4)) >> 16) + ((yb1[:, None, None] * (s1 >> 4)) >> 16) + 2) >> 2
    return out.astype(np.uint8)


def letterbox(img: np.ndarray, imgsz: int = 640, stride: int = 32, auto: bool = True):
    """ultralytics ``LetterBox(new_shape, auto=True, stride=32)(image=img)`` -> (img, (top, left))."""
    h, w = img.shape[:2]
    r = min(imgsz / h, imgsz / w)
    new_unpad = int(round(w * r)), int(round(h * r))
    dw, dh = imgsz - new_unpad[0], imgsz - new_unpad[1]
    if auto:
        dw, dh = np.mod(dw, stride), np.mod(dh, stride)
    dw /= 2
    dh /= 2
    if (w, h) != new_unpad:
        img = resize_linear_u8(img, new_unpad[0], new_unpad[1])
    top, bottom = int(round(dh - 0.1)), int(round(dh + 0.1))
    left, right = int(round(dw - 0.1)), int(round(dw + 0.1))
    out = np.full((img.shape[0] + top + bottom, img.shape[1] + left + right, 3), 114, dtype=np.uint8)
    out[top:top + img.shape[0], left:left + img.shape[1]] = img
    return out, (top, left)


def preprocess(frames_bgr: np.ndarray, imgsz: int = 640, fp16: bool = True):
    """BGR u8 (n,h,w,3) -> float32 NCHW RGB /255, fp16-valued unless ``fp16=False`` (``im.float(); im /= 255``)."""
    import torch

    lb = np.stack([letterbox(f, imgsz)[0] for f in frames_bgr])
    x = torch.from_numpy(lb[..., ::-1].copy()).permute(0, 3, 1, 2).float()
    return (x / 255.0).half().float() if fp16 else x / 255.0


# --------------------------------------------------------------------------------------------
# network (torch-CPU, fp16 storage emulation)   [PUBLIC-LIB: ultralytics nn/modules]
# --------------------------------------------------------------------------------------------

def _h(t):
    return t.half().float()


class Net:
    """``fp16=True`` (default): the fp16 network the HIP path runs.  ``fp16=False``: the reference's own arithmetic
    (Ultralytics predicts with ``half=False``, ``model_manager.py:270-275``): fp32 weights and activations - the
    validation mode SURVEY.md 7 asks for, used to bound the fp16 build's drift from the reference."""

    def __init__(self, state: dict, variant_ch, variant_depth, nc: int, fp16: bool = True):
        import torch

        self.nc = nc
        self.fp16 = fp16
        self.ch, self.depth = variant_ch, variant_depth
        rw = (lambda t: t.half().float()) if fp16 else (lambda t: t)
        self.p = {k: (rw(torch.from_numpy(np.asarray(w, np.float32))), torch.from_numpy(np.asarray(b, np.float32)))
                  for k, (w, b) in state.items()}

    def _h(self, t):
        return t.half().float() if self.fp16 else t

    def conv(self, name, x, stride=1, act=True, keep_f32=False):
        import torch
        import torch.nn.functional as F

        w, b = self.p[name]
        y = F.conv2d(x, w, b, stride=stride, padding=w.shape[-1] // 2)
        if act:
            y = y * torch.sigmoid(y)
        return y if keep_f32 else self._h(y)

    def c2f(self, p, x, n, shortcut):
        import torch

        y = list(self.conv(f"{p}.cv1.conv", x).chunk(2, 1))
        for i in range(n):
            t = self.conv(f"{p}.m.{i}.cv2.conv", self.conv(f"{p}.m.{i}.cv1.conv", y[-1]))
            y.append(self._h(y[-1] + t) if shortcut else t)
        return self.conv(f"{p}.cv2.conv", torch.cat(y, 1))

    def sppf(self, x):
        import torch
        import torch.nn.functional as F

        y = [self.conv("model.9.cv1.conv", x)]
        for _ in range(3):
            y.append(F.max_pool2d(y[-1], 5, 1, 2))
        return self.conv("model.9.cv2.conv", torch.cat(y, 1))

    def forward(self, x, penultimate: bool = False):
        """x: float32 NCHW (fp16-valued).  Returns (box[3], cls[3]) float32 NHWC numpy arrays; with ``penultimate``
        also the inputs of the six Detect output convs (``[box feat[3], cls feat[3]]``, NHWC) - tests build
        well-conditioned heads on top of them."""
        import torch
        import torch.nn.functional as F

        d0, d1, d2, d3 = self.depth
        with torch.no_grad():
            x0 = self.conv("model.0.conv", x, 2)
            x1 = self.conv("model.1.conv", x0, 2)
            x2 = self.c2f("model.2", x1, d0, True)
            x3 = self.conv("model.3.conv", x2, 2)
            x4 = self.c2f("model.4", x3, d1, True)
            x5 = self.conv("model.5.conv", x4, 2)
            x6 = self.c2f("model.6", x5, d2, True)
            x7 = self.conv("model.7.conv", x6, 2)
            x8 = self.c2f("model.8", x7, d3, True)
            x9 = self.sppf(x8)
            x12 = self.c2f("model.12", torch.cat([F.interpolate(x9, scale_factor=2, mode="nearest"), x6], 1), d0, False)
            x15 = self.c2f("model.15", torch.cat([F.interpolate(x12, scale_factor=2, mode="nearest"), x4], 1), d0, False)
            x18 = self.c2f("model.18", torch.cat([self.conv("model.16.conv", x15, 2), x12], 1), d0, False)
            x21 = self.c2f("model.21", torch.cat([self.conv("model.19.conv", x18, 2), x9], 1), d0, False)
            box, cls, fb, fc = [], [], [], []
            for l, f in enumerate((x15, x18, x21)):
                b = self.conv(f"model.22.cv2.{l}.1.conv", self.conv(f"model.22.cv2.{l}.0.conv", f))
                c = self.conv(f"model.22.cv3.{l}.1.conv", self.conv(f"model.22.cv3.{l}.0.conv", f))
                box.append(self.conv(f"model.22.cv2.{l}.2", b, act=False, keep_f32=True).permute(0, 2, 3, 1).contiguous().numpy())
                cls.append(self.conv(f"model.22.cv3.{l}.2", c, act=False, keep_f32=True).permute(0, 2, 3, 1).contiguous().numpy())
                if penultimate:
                    fb.append(b.permute(0, 2, 3, 1).contiguous().numpy())
                    fc.append(c.permute(0, 2, 3, 1).contiguous().numpy())
        if penultimate:
            return box, cls, fb, fc
        return box, cls


# --------------------------------------------------------------------------------------------
# decode + NMS + scale_boxes   [PUBLIC-LIB: ultralytics Detect / ops.non_max_suppression, torchvision nms]
# --------------------------------------------------------------------------------------------

def decode(box_maps, cls_maps):
    """Detect head inference path in fp32, per image: xyxy boxes (letterboxed px) and class scores.

    ``dfl``: softmax over 16 bins, expectation with arange(16) accumulated in bin order;
    ``dist2bbox(xywh=True) * stride``; then ``xywh2xyxy`` as ``non_max_suppression`` applies it.
    Returns ``(boxes (n,A,4) f32, scores (n,A,nc) f32)`` with A = anchors of P3|P4|P5.
    """
    f32 = np.float32
    boxes, scores = [], []
    for l, (bm, cm) in enumerate(zip(box_maps, cls_maps)):
        n, h, w, _ = bm.shape
        stride = f32(8 << l)
        lg = bm.reshape(n, h * w, 4, 16).astype(f32)
        m = lg.max(axis=-1, keepdims=True)
        e = np.exp((lg - m).astype(f32)).astype(f32)
        s = np.zeros(e.shape[:-1], dtype=f32)
        for i in range(16):
            s = (s + e[..., i]).astype(f32)
        d = np.zeros(e.shape[:-1], dtype=f32)
        for i in range(16):
            d = (d + ((e[..., i] / s).astype(f32) * f32(i)).astype(f32)).astype(f32)
        ax = (np.arange(w, dtype=f32) + f32(0.5))[None, :].repeat(h, 0).reshape(-1)
        ay = (np.arange(h, dtype=f32) + f32(0.5))[:, None].repeat(w, 1).reshape(-1)
        x1, y1 = ax[None] - d[..., 0], ay[None] - d[..., 1]
        x2, y2 = ax[None] + d[..., 2], ay[None] + d[..., 3]
        cx, cy = ((x1 + x2) / f32(2)) * stride, ((y1 + y2) / f32(2)) * stride
        bw, bh = (x2 - x1) * stride, (y2 - y1) * stride
        hw, hh = bw / f32(2), bh / f32(2)
        boxes.append(np.stack([cx - hw, cy - hh, cx + hw, cy + hh], -1).astype(f32))
        scores.append((f32(1) / (f32(1) + np.exp(-cm.reshape(n, h * w, -1).astype(f32)))).astype(f32))
    return np.concatenate(boxes, 1), np.concatenate(scores, 1)


def nms_torchvision(boxes: np.ndarray, order: np.ndarray, iou_thres: float) -> list[int]:
    """``torchvision.ops.nms`` CPU kernel on boxes visited in ``order`` (fp32 arithmetic)."""
    f32 = np.float32
    x1, y1, x2, y2 = (boxes[:, i] for i in range(4))
    areas = ((x2 - x1) * (y2 - y1)).astype(f32)
    suppressed = np.zeros(len(order), dtype=bool)
    keep = []
    thr = f32(iou_thres)
    for a in range(len(order)):
        if suppressed[a]:
            continue
        i = order[a]
        keep.append(int(i))
        rest = order[a + 1:]
        xx1 = np.maximum(x1[i], x1[rest]); yy1 = np.maximum(y1[i], y1[rest])
        xx2 = np.minimum(x2[i], x2[rest]); yy2 = np.minimum(y2[i], y2[rest])
        w = np.maximum(f32(0), (xx2 - xx1).astype(f32)); h = np.maximum(f32(0), (yy2 - yy1).astype(f32))
        inter = (w * h).astype(f32)
        with np.errstate(divide="ignore", invalid="ignore"):
            ovr = (inter / ((areas[i] + areas[rest]).astype(f32) - inter).astype(f32)).astype(f32)
        suppressed[a + 1:] |= ovr > thr
    return keep


def non_max_suppression(boxes, scores, conf_thres: float, iou_thres: float = 0.7, max_det: int = 300,
                        max_wh: float = 7680.0, max_nms: int = 30000):
    """Per image: [(anchor_index, xyxy f32, conf f32, cls int)] in kept order (single-label, class aware).

    Ties in confidence are visited in ascending anchor order (torch's argsort leaves them unspecified).
    """
    f32 = np.float32
    out = []
    for b, s in zip(boxes, scores):
        conf = s.max(axis=1)
        j = s.argmax(axis=1)
        cand = np.nonzero(conf > f32(conf_thres))[0]
        order = cand[np.lexsort((cand, -conf[cand].astype(np.float64)))][:max_nms]
        off = (j[order].astype(f32) * f32(max_wh))[:, None]
        nb = (b[order] + off).astype(f32)
        keep = nms_torchvision(nb, np.arange(len(order)), iou_thres)[:max_det]
        out.append([(int(order[k]), b[order[k]].copy(), f32(conf[order[k]]), int(j[order[k]])) for k in keep])
    return out


def scale_boxes(box: np.ndarray, lb_shape, orig_shape) -> np.ndarray:
    """``ops.scale_boxes(img1_shape=lb_shape, boxes, img0_shape=orig_shape)`` + ``clip_boxes`` (fp32)."""
    f32 = np.float32
    gain = min(lb_shape[0] / orig_shape[0], lb_shape[1] / orig_shape[1])
    pad_x = round((lb_shape[1] - orig_shape[1] * gain) / 2 - 0.1)
    pad_y = round((lb_shape[0] - orig_shape[0] * gain) / 2 - 0.1)
    b = box.astype(f32).copy()
    b[[0, 2]] = (b[[0, 2]] - f32(pad_x)).astype(f32)
    b[[1, 3]] = (b[[1, 3]] - f32(pad_y)).astype(f32)
    b = (b / f32(gain)).astype(f32)
    b[[0, 2]] = np.clip(b[[0, 2]], f32(0), f32(orig_shape[1]))
    b[[1, 3]] = np.clip(b[[1, 3]], f32(0), f32(orig_shape[0]))
    return b


def detect(net: Net, frames_bgr: np.ndarray, conf: float, iou: float = 0.7, max_det: int = 300, imgsz: int = 640):
    """End to end: list (per frame) of dicts {anchor, xyxy (orig px, f32), conf, cls}."""
    x = preprocess(frames_bgr, imgsz)
    box_maps, cls_maps = net.forward(x)
    boxes, scores = decode(box_maps, cls_maps)
    res = non_max_suppression(boxes, scores, conf, iou, max_det)
    h, w = frames_bgr.shape[1:3]
    lb_shape = x.shape[2:]
    return [[{"anchor": a, "xyxy": scale_boxes(b, lb_shape, (h, w)), "conf": float(c), "cls": k}
             for a, b, c, k in per] for per in res], (box_maps, cls_maps, boxes, scores)
