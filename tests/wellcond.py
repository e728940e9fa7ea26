"""Which detections of a head-map pair are decided by MARGIN and which by rounding (test infrastructure).

The HIP detector and the oracle run the same fp16 network, but MFMA accumulation order differs from torch-CPU's and
every layer rounds to fp16: head logits drift by ~1e-3 of a map's RMS.  A random-init YOLOv8 answers neighbouring
anchors with near-identical scores and boxes, so WHICH of two neighbours survives NMS can hinge on that drift; the
detections that do not hinge on it must match exactly - kept anchor, class and order (BASELINE.json: "post-NMS box
indices bit-exact").  `stably_kept` decides membership from measured drift bounds:

  an anchor the NMS kept is margin-stable when (1) its confidence clears the threshold by more than the conf drift
  `dc`, (2) its class leads the runner-up class by more than 2 dc, and (3) no candidate that could be visited
  BEFORE it (confidence within 2 dc of it or above) and could carry its class overlaps it by more than iou - `du`.

Greedy NMS keeps such an anchor under every perturbation of size (dc, du): nothing that may precede it can suppress it.
"""
from __future__ import annotations

import numpy as np

from eioku_amd import weights as W
from oracle import yolo as oy


def blob_frames(seed: int, n: int, h: int, w: int) -> np.ndarray:
    """Textured frames (8x8 colour blocks + noise): features, scores and boxes vary from anchor to anchor."""
    rng = np.random.default_rng(seed)
    blobs = rng.integers(0, 256, (h // 8 + 1, w // 8 + 1, 3))
    up = np.repeat(np.repeat(blobs, 8, 0), 8, 1)[:h, :w]
    out = []
    for i in range(n):
        f = np.roll(up, (37 * i, 101 * i), (0, 1)) + rng.integers(-20, 21, (h, w, 3))
        out.append(np.clip(f, 0, 255).astype(np.uint8))
    return np.stack(out)


def calibrated_state(frames, variant="n", nc=80, seed=7, frac=0.08, conf=0.25):
    """Random weights whose Detect logits are O(1) on THESE frames (a random net's logit scale depends on its
    input): rescale the six output convs so box logits have std 2 and class logits std 3, shifted so that about
    `frac` of the anchors pass `conf`."""
    state = W.random_state(variant, nc, seed=seed)
    box, cls = oy.Net(state, *W.YOLO_VARIANTS[variant], nc).forward(oy.preprocess(frames))
    scaled = [(c - c.mean()) * (3.0 / c.std()) for c in cls]
    top = np.concatenate([c.max(axis=-1).reshape(-1) for c in scaled])
    shift = float(np.log(conf / (1 - conf)) - np.quantile(top, 1.0 - frac))
    for l in range(3):
        w, b = state[f"model.22.cv2.{l}.2"]
        state[f"model.22.cv2.{l}.2"] = ((w * (2.0 / box[l].std())).astype(np.float32), (b * 0).astype(np.float32))
        w, b = state[f"model.22.cv3.{l}.2"]
        sc = 3.0 / cls[l].std()
        state[f"model.22.cv3.{l}.2"] = ((w * sc).astype(np.float32),
                                        ((b - cls[l].mean()) * sc + shift).astype(np.float32))
    return state


def iou_matrix(a, b):
    a, b = a.astype(np.float64), b.astype(np.float64)
    x1 = np.maximum(a[:, None, 0], b[None, :, 0])
    y1 = np.maximum(a[:, None, 1], b[None, :, 1])
    x2 = np.minimum(a[:, None, 2], b[None, :, 2])
    y2 = np.minimum(a[:, None, 3], b[None, :, 3])
    inter = np.clip(x2 - x1, 0, None) * np.clip(y2 - y1, 0, None)
    aa = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    ab = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    return inter / np.maximum(aa[:, None] + ab[None, :] - inter, 1e-30)


def stably_kept(boxes, scores, kept, thr, dc, du, iou_thr=0.7):
    """The margin-stable subset of `kept` (one image's NMS result on `boxes` (A,4) / `scores` (A,nc)), in order."""
    conf = scores.max(1)
    cls = scores.argmax(1)
    out = []
    for a in kept:
        if conf[a] <= thr + dc:
            continue
        s = np.sort(scores[a])
        if len(s) > 1 and s[-1] - s[-2] <= 2 * dc:
            continue
        ahead = np.nonzero((scores[:, cls[a]] > conf[a] - 2 * dc) & (scores[:, cls[a]] > thr - dc))[0]
        ahead = ahead[ahead != a]
        if len(ahead) and iou_matrix(boxes[a:a + 1], boxes[ahead]).max() >= iou_thr - du:
            continue
        out.append(int(a))
    return out


def drift(boxes_a, scores_a, boxes_b, scores_b, thr):
    """Measured perturbation between two decodes of one image: (max |conf difference| over every anchor and class,
    max |IoU difference| over every pair of anchors that is a candidate (conf > thr / 2) in either)."""
    dc = float(np.abs(scores_a - scores_b).max())
    cand = np.nonzero((scores_a.max(1) > thr / 2) | (scores_b.max(1) > thr / 2))[0]
    du = 0.0
    if len(cand) > 1:
        du = float(np.abs(iou_matrix(boxes_a[cand], boxes_a[cand]) - iou_matrix(boxes_b[cand], boxes_b[cand])).max())
    return dc, du
