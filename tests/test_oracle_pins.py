"""CPU pins of the arithmetic oracles by INDEPENDENT derivations ("bit-exact K3 proves agreement
with itself").  cv2 / ultralytics / faiss are not installable here, so each check derives the expected values from
the library's DOCUMENTED definition by a different route than the oracle's integer code:

  * cv2.resize INTER_LINEAR: the documented half-pixel-centre bilinear definition evaluated in float64 (the fixed
    point result may differ from its rounding by at most one code), and the closed form at the reference's own frame
    size: 1920x1080 -> 640x360 is scale 3 exactly, every tap lands ON a source pixel, so the resize must be the pure
    decimation src[3y+1, 3x+1].  oracle/yolo.py and eioku_amd/detect.py (two restatements of resize.cpp) must both
    satisfy them, and must agree with a third coefficient derivation written with exact fractions below.
  * cv2 COLOR_BGR2HSV (8 bit): the documented float formula (V = max, S = 255 (V - min) / V, H = 30 (G - B) / (V - min)
    ...) rounded, over a lattice: the table algorithm may differ by one code where the float value sits near .5.
  * Ultralytics LetterBox geometry at the reference's sizes; scale_boxes round trip.
  * FAISS: squared distances, ascending, -1 / FLT_MAX padding, smaller id first on exact ties.
  * the fp16 network's drift from the reference's fp32 arithmetic (half=False) stays inside the documented bound.
"""
from fractions import Fraction

import numpy as np
import pytest

from eioku_amd import detect as D, weights as W
from oracle import knn as oknn, prng, scene as oscene, yolo as oy


# ---------------------------------------------------------------------------------------------------------------
# cv2.resize(INTER_LINEAR)
# ---------------------------------------------------------------------------------------------------------------
def _fraction_coeffs(src, dst, horizontal):
    """Third derivation of the tap tables: the source coordinate as an exact fraction, then ONE rounding to float32
    where resize.cpp casts to float.  (dx + 0.5) * scale - 0.5 with scale = 1.0 / (dst / src) evaluated in double."""
    scale = 1.0 / (dst / src)
    out = []
    for d in range(dst):
        exact = Fraction(2 * d + 1, 2) * Fraction(scale) - Fraction(1, 2)  # the double-precision product, exactly
        f64 = float(exact)  # == the C double expression: one rounding (products of doubles round once)
        f = np.float32(f64)
        s = int(np.floor(f))
        f = np.float32(f - np.float32(s))
        if horizontal and s < 0:
            s, f = 0, np.float32(0)
        if horizontal and s >= src - 1:
            s, f = src - 1, np.float32(0)
        a1 = int(np.rint(np.float32(f * np.float32(2048))))
        a0 = int(np.rint(np.float32((np.float32(1) - f) * np.float32(2048))))
        out.append((s, a0, a1))
    return out


@pytest.mark.parametrize("src,dst", [(1920, 640), (1080, 360), (854, 640), (480, 360), (517, 397), (1000, 640), (333, 213)])
def test_three_derivations_of_the_tap_tables_agree(src, dst):
    for horizontal in (True, False):
        third = _fraction_coeffs(src, dst, horizontal)
        xo, a0, a1 = oy._axis_coeffs(src, dst, horizontal)
        po, pc = D._linear_coeffs(src, dst, horizontal)
        assert [(int(o), int(x), int(y)) for o, x, y in zip(xo, a0, a1)] == third
        assert [(int(o), int(c[0]), int(c[1])) for o, c in zip(po, pc)] == third
        assert all(x + y == 2048 for _, x, y in third)


def test_1080p_resize_is_pure_decimation():
    """scale 3: fx = 3 dx + 1 exactly -> weight 2048 on src[3y+1, 3x+1], and the 11-bit pipeline returns S itself."""
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (1080, 1920, 3), dtype=np.uint8)
    out = oy.resize_linear_u8(img, 640, 360)
    assert np.array_equal(out, img[1::3, 1::3])


@pytest.mark.parametrize("h,w,dh,dw", [(480, 854, 360, 640), (333, 517, 247, 384), (200, 1000, 128, 640)])
def test_fixed_point_resize_is_within_one_code_of_float64_bilinear(h, w, dh, dw):
    """OpenCV docs: dst(x, y) samples src at ((x + 0.5) * sw/dw - 0.5, ...) bilinearly, borders replicated."""
    rng = np.random.default_rng(h + w)
    img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    got = oy.resize_linear_u8(img, dw, dh).astype(np.float64)
    fx = (np.arange(dw) + 0.5) * (w / dw) - 0.5
    fy = (np.arange(dh) + 0.5) * (h / dh) - 0.5
    x0 = np.floor(fx).astype(int)
    y0 = np.floor(fy).astype(int)
    ax = fx - x0
    ay = fy - y0
    cx0, cx1 = np.clip(x0, 0, w - 1), np.clip(x0 + 1, 0, w - 1)
    cy0, cy1 = np.clip(y0, 0, h - 1), np.clip(y0 + 1, 0, h - 1)
    s = img.astype(np.float64)
    top = s[cy0][:, cx0] * (1 - ax)[None, :, None] + s[cy0][:, cx1] * ax[None, :, None]
    bot = s[cy1][:, cx0] * (1 - ax)[None, :, None] + s[cy1][:, cx1] * ax[None, :, None]
    want = top * (1 - ay)[:, None, None] + bot * ay[:, None, None]
    err = got - want
    # rounding (0.5) + the two truncating shifts of VResizeLinear ((b * (S >> 4)) >> 16 per tap: a small negative bias)
    assert np.abs(err).max() <= 1.0 and abs(err.mean()) < 0.2 and np.abs(err).mean() < 0.35


def test_letterbox_geometry_known_answers():
    """Ultralytics rect inference (auto=True, stride 32): the reference's 16:9 sources all land on 384x640."""
    for h, w in [(1080, 1920), (480, 854), (720, 1280), (2160, 3840)]:
        img = np.zeros((h, w, 3), np.uint8)
        out, (top, left) = oy.letterbox(img)
        assert out.shape == (384, 640, 3) and (top, left) == (12, 0)
        assert (out[:12] == 114).all() and (out[-12:] == 114).all() and (out[12:-12] == 0).all()
        p = D.letterbox_plan(h, w)
        assert (p.out_h, p.out_w, p.top, p.left, p.new_h, p.new_w) == (384, 640, 12, 0, 360, 640)
    out, (top, left) = oy.letterbox(np.zeros((1920, 1080, 3), np.uint8))  # portrait
    assert out.shape == (640, 384, 3) and (top, left) == (0, 12)
    out, pad = oy.letterbox(np.zeros((640, 640, 3), np.uint8))
    assert out.shape == (640, 640, 3) and pad == (0, 0)


def test_scale_boxes_round_trip_at_reference_sizes():
    for h, w in [(1080, 1920), (480, 854)]:
        p = D.letterbox_plan(h, w)
        box = np.array([100.0, 50.0, 900.0, 700.0], np.float32) * np.float32(min(h / 1080, w / 1920))
        lb = np.array([box[0] * p.gain + p.pad_x, box[1] * p.gain + p.pad_y, box[2] * p.gain + p.pad_x, box[3] * p.gain + p.pad_y],
                      np.float32)
        back = oy.scale_boxes(lb, (p.out_h, p.out_w), (h, w))
        assert np.allclose(back, box, atol=2e-3)
        assert p.gain == pytest.approx(min(384 / h, 640 / w)) and (p.pad_x, p.pad_y) == (0, 12)


# ---------------------------------------------------------------------------------------------------------------
# COLOR_BGR2HSV
# ---------------------------------------------------------------------------------------------------------------
def _hsv_float(bgr):
    b, g, r = (bgr[..., i].astype(np.float64) for i in range(3))
    v = np.maximum(np.maximum(b, g), r)
    mn = np.minimum(np.minimum(b, g), r)
    diff = v - mn
    s = np.where(v > 0, 255.0 * diff / np.where(v > 0, v, 1), 0.0)
    d = np.where(diff > 0, diff, 1)
    h = np.where(v == r, (g - b) / d, np.where(v == g, 2 + (b - r) / d, 4 + (r - g) / d)) * 30.0  # 60 deg / 2
    h = np.where(diff > 0, h, 0.0)
    h = np.where(h < 0, h + 180.0, h)
    return h, s, v


def test_hsv_table_algorithm_tracks_the_documented_float_formula():
    g = np.arange(0, 256, 5, dtype=np.uint8)
    bgr = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)
    got = oscene.bgr2hsv_u8(bgr).astype(np.float64)
    h, s, v = _hsv_float(bgr)
    assert np.array_equal(got[:, 2], v)
    assert np.abs(got[:, 1] - s).max() <= 0.53                             # S: rounded, 12-bit reciprocal table
    dh = np.abs(got[:, 0] - h)
    dh = np.minimum(dh, 180 - dh)                                          # hue is circular: 179.6 rounds to 180 -> 0
    assert dh.max() <= 0.7 and (dh > 0.5).mean() < 0.01  # + the 12-bit hue reciprocal's error (up to 0.18 at H ~ 180)
    assert got[:, 0].max() <= 180 and (got[:, 0] == 180).mean() < 1e-3
    for px, want in [((50, 100, 150), (15, 170, 150)), ((200, 40, 10), (115, 242, 200)), ((10, 200, 40), (55, 242, 200)),
                     ((0, 255, 255), (30, 255, 255)), ((255, 255, 0), (90, 255, 255)), ((255, 0, 255), (150, 255, 255)),
                     ((37, 37, 37), (0, 0, 37))]:
        assert tuple(int(x) for x in oscene.bgr2hsv_u8(np.array([px], np.uint8))[0]) == want, px


# ---------------------------------------------------------------------------------------------------------------
# FAISS IndexFlatL2 documented semantics
# ---------------------------------------------------------------------------------------------------------------
def test_flat_l2_documented_semantics():
    xb = np.eye(6, 8, dtype=np.float32)
    xb[4] = xb[1]  # exact duplicate
    q = np.zeros((2, 8), np.float32)
    q[0, 1] = 1.0
    q[1, 7] = 2.0
    Dd, I = oknn.search(xb, q, 8)
    assert I.dtype == np.int64  # distances: float64 ground truth here; the product returns float32 / FLT_MAX padding
    assert list(I[0, :2]) == [1, 4] and Dd[0, 0] == 0 and Dd[0, 1] == 0          # tie: smaller id first
    assert np.allclose(Dd[0, 2:6], 2.0)                                        # SQUARED L2: |e_i - e_j|^2 = 2
    assert np.all(I[:, 6:] == -1) and np.all(np.isinf(Dd[:, 6:]))               # fewer than k rows: -1 / +inf-like
    assert np.all(np.diff(Dd[:, :6], axis=1) >= 0)
    assert np.allclose(Dd[1, :6], 5.0)                                         # |2 e_7 - e_i|^2 = 4 + 1


# ---------------------------------------------------------------------------------------------------------------
# fp16 build vs the reference's fp32 arithmetic
# ---------------------------------------------------------------------------------------------------------------
FP16_VS_FP32_MAX, FP16_VS_FP32_MEAN = 0.04, 0.008  # of the map's RMS; measured 1.2e-2 / 2.0e-3 on YOLOv8n


def test_fp16_network_drift_from_fp32_reference_arithmetic_is_bounded():
    """The reference predicts in fp32 (half=False); BASELINE cfg2 asks for fp16.  The oracle's fp32 mode is the
    reference's arithmetic on the same weights: head logits of the fp16 network stay within the documented bound,
    which is what limits end-to-end index parity to margin-stable detections (tests/test_yolo_gpu.py)."""
    frames = prng.synth_frames_bgr(21, 1, 240, 427)
    state = W.random_state("n", 80, seed=7)
    h = oy.Net(state, *W.YOLO_VARIANTS["n"], 80, fp16=True).forward(oy.preprocess(frames))
    f = oy.Net(state, *W.YOLO_VARIANTS["n"], 80, fp16=False).forward(oy.preprocess(frames, fp16=False))
    for a, b in zip(h[0] + h[1], f[0] + f[1]):
        rms = float(np.sqrt((b.astype(np.float64) ** 2).mean()))
        err = np.abs(a - b)
        assert err.max() <= FP16_VS_FP32_MAX * rms and err.mean() <= FP16_VS_FP32_MEAN * rms, (err.max() / rms, err.mean() / rms)
