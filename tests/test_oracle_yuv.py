"""CPU: known answers for the YUV 4:2:0 -> BGR oracle (OpenCV's integer BT.601; cv2 is not installed here)."""
import numpy as np

from oracle import yuv as oyuv


def one(y, u, v):
    p = np.zeros((3, 2), np.uint8)  # a 2 x 2 frame: Y plane 2 x 2, then one U and one V sample (I420)
    p[:2] = y
    p[2, 0], p[2, 1] = u, v
    return oyuv.yuv420_to_bgr(p, 2, 2)[0, 0]


def test_studio_range_anchors_and_coefficients():
    assert list(one(16, 128, 128)) == [0, 0, 0] and list(one(235, 128, 128)) == [255, 255, 255]
    assert list(one(0, 128, 128)) == [0, 0, 0] and list(one(255, 128, 128)) == [255, 255, 255]  # super-black / white clip
    assert list(one(126, 128, 128)) == [128, 128, 128]  # (126 - 16) * 1.164 = 128.04
    # the fixed-point coefficients are the float BT.601 ones to 6 digits
    f = 1 << oyuv.SHIFT
    assert np.allclose([oyuv.CY / f, oyuv.CUB / f, -oyuv.CUG / f, -oyuv.CVG / f, oyuv.CVR / f], [1.164, 2.018, 0.391, 0.813, 1.596], atol=5e-4)
    # saturated primaries of BT.601 (Y, Cb, Cr): red (81, 90, 240), green (145, 54, 34), blue (41, 240, 110) -> within 2 codes
    assert np.abs(one(81, 90, 240).astype(int) - [0, 0, 255]).max() <= 2
    assert np.abs(one(145, 54, 34).astype(int) - [0, 255, 0]).max() <= 2
    assert np.abs(one(41, 240, 110).astype(int) - [255, 0, 0]).max() <= 2


def test_layouts_and_chroma_sharing():
    rng = np.random.default_rng(3)
    h, w = 6, 8
    p = rng.integers(0, 256, (2, h * 3 // 2, w), dtype=np.uint8)
    bgr = oyuv.yuv420_to_bgr(p, h, w, "i420")
    assert bgr.shape == (2, h, w, 3)
    assert np.array_equal(oyuv.yuv420_to_bgr(oyuv.i420_to_nv12(p, h, w), h, w, "nv12"), bgr)
    # grey chroma: B = G = R, monotone in Y; the four pixels of a block share U / V
    g = p.copy()
    g[:, h:] = 128
    out = oyuv.yuv420_to_bgr(g, h, w)
    assert np.array_equal(out[..., 0], out[..., 1]) and np.array_equal(out[..., 1], out[..., 2])
    q = p.copy()
    q[:, :h] = 100
    o2 = oyuv.yuv420_to_bgr(q, h, w)
    assert np.array_equal(o2[:, 0::2, 0::2], o2[:, 1::2, 1::2]) and np.array_equal(o2[:, 0::2, 0::2], o2[:, 0::2, 1::2])


def test_round_trip_of_the_test_encoder_is_close():
    rng = np.random.default_rng(5)
    f = np.repeat(np.repeat(rng.integers(0, 256, (2, 4, 6, 3), dtype=np.uint8), 2, 1), 2, 2)  # 2 x 2 blocks: no chroma loss
    back = oyuv.yuv420_to_bgr(oyuv.bgr_to_i420(f), 8, 12)
    assert np.abs(back.astype(int) - f.astype(int)).max() <= 3
