"""CPU checks of oracle/yolo.py (the detection-stage checker) against known answers.

Ultralytics / OpenCV / torchvision are not importable here (third-party, absent), so the
restatement is anchored on hand-computable cases of the published algorithms it follows: LetterBox
geometry of the reference's frame sizes, the DFL expectation, torchvision's greedy NMS and
``ops.scale_boxes``; the torch-CPU cross-check of the network itself runs in tests/test_yolo_gpu.py.
"""
import numpy as np

from eioku_amd import detect as D, weights as W
from oracle import yolo as oy


def test_u8_over_255_as_fp16_equals_multiply_by_reciprocal():
    """The fused stem (k_conv3x3_c8) converts bytes with v * (1/255) where K3 and Ultralytics divide by 255:
    after the fp16 rounding the two agree for every byte value, so the fused network input is bit-identical."""
    v = np.arange(256, dtype=np.float32)
    div = (v / np.float32(255.0)).astype(np.float16)
    mul = (v * np.float32(1.0 / 255.0)).astype(np.float16)
    assert np.array_equal(div.view(np.uint16), mul.view(np.uint16))


def test_letterbox_geometry_matches_plan_and_reference_sizes():
    for h, w in [(1080, 1920), (720, 1280), (480, 854), (640, 640), (470, 640), (96, 160), (333, 517)]:
        img = np.zeros((h, w, 3), np.uint8)
        out = oy.letterbox(img)[0]
        p = D.letterbox_plan(h, w)
        assert out.shape == (p.out_h, p.out_w, 3)
    p = D.letterbox_plan(1080, 1920)
    assert (p.new_h, p.new_w, p.top, p.left, p.out_h, p.out_w) == (360, 640, 12, 0, 384, 640)


def test_letterbox_pads_with_114_and_keeps_pixels_in_copy_mode():
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (470, 640, 3), dtype=np.uint8)
    out = oy.letterbox(img)[0]
    assert out.shape == (480, 640, 3)
    assert np.array_equal(out[5:475], img)
    assert (out[:5] == 114).all() and (out[475:] == 114).all()


def test_resize_half_is_2x2_area_average():
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (8, 12, 3), dtype=np.uint8)
    got = oy.resize_linear_u8(img, 6, 4)
    want = (img.reshape(4, 2, 6, 2, 3).astype(np.int32).sum(axis=(1, 3)) + 2) >> 2
    assert np.array_equal(got, want.astype(np.uint8))


def test_dfl_uniform_logits_give_the_mean_bin_and_centre_box():
    box = [np.zeros((1, 2, 2, 64), np.float32), np.zeros((1, 1, 1, 64), np.float32), np.zeros((1, 1, 1, 64), np.float32)]
    cls = [np.full((1, 2, 2, 3), -20.0, np.float32), np.zeros((1, 1, 1, 3), np.float32), np.zeros((1, 1, 1, 3), np.float32)]
    boxes, scores = oy.decode(box, cls)
    assert boxes.shape == (1, 6, 4) and scores.shape == (1, 6, 3)
    # uniform softmax over 16 bins -> expected distance 7.5 cells on every side; anchor (0.5,0.5) at stride 8
    assert np.allclose(boxes[0, 0], [(0.5 - 7.5) * 8, (0.5 - 7.5) * 8, (0.5 + 7.5) * 8, (0.5 + 7.5) * 8], atol=1e-4)
    assert np.allclose(boxes[0, 5], [(0.5 - 7.5) * 32, (0.5 - 7.5) * 32, (0.5 + 7.5) * 32, (0.5 + 7.5) * 32], atol=1e-3)
    assert np.allclose(scores[0, 4], 0.5) and scores[0, 0].max() < 1e-8


def test_nms_is_greedy_class_aware_and_strict_about_the_threshold():
    b = np.array([[0, 0, 10, 10], [1, 0, 11, 10], [0, 0, 10, 10], [50, 50, 60, 60], [0, 0, 10, 5]], np.float32)
    s = np.zeros((5, 2), np.float32)
    s[0, 0], s[1, 0], s[2, 1], s[3, 0], s[4, 0] = 0.9, 0.8, 0.85, 0.3, 0.6
    out = oy.non_max_suppression(b[None], s[None], 0.25, 0.7, 300)[0]
    # 1 overlaps 0 with IoU 9/11 > 0.7 (same class) -> suppressed; 2 is another class -> kept;
    # 4 overlaps 0 with IoU exactly 0.5 -> kept; order = descending confidence
    assert [a for a, *_ in out] == [0, 2, 4, 3]
    assert [c for *_, c in out] == [0, 1, 0, 0]
    # IoU exactly AT the threshold does not suppress (torchvision: ovr > thr)
    out = oy.non_max_suppression(b[None], s[None], 0.25, 0.5, 300)[0]
    assert 4 in [a for a, *_ in out]
    assert len(oy.non_max_suppression(b[None], s[None], 0.25, 0.7, 2)[0]) == 2


def test_scale_boxes_undoes_the_letterbox_and_clips():
    # 1080x1920 -> 384x640: gain 1/3, pad_y = round(12 - 0.1) = 12
    got = oy.scale_boxes(np.array([64.0, 12.0 + 30.0, 640.0 + 5, 384.0], np.float32), (384, 640), (1080, 1920))
    assert np.allclose(got, [192.0, 90.0, 1920.0, 1080.0], atol=1e-3)


def test_margin_stable_detections_survive_every_perturbation_within_the_margin():
    """tests/wellcond.py: an anchor declared margin-stable for drift (dc, du) stays kept, with its class, when every
    class logit moves by a random amount that changes confidences by less than dc (soundness of the criterion the
    GPU end-to-end test scopes its exact index comparison to)."""
    import wellcond

    frames = wellcond.blob_frames(5, 1, 135, 240)
    state = wellcond.calibrated_state(frames, "n", 3, seed=7, frac=0.08)
    net = oy.Net(state, *W.YOLO_VARIANTS["n"], 3)
    ref, (box_maps, cls_maps, boxes, scores) = oy.detect(net, frames, 0.25)
    kept = [d["anchor"] for d in ref[0]]
    dc = 0.01
    stable = wellcond.stably_kept(boxes[0], scores[0], kept, 0.25, dc, 1e-4)
    assert 5 <= len(stable) <= len(kept) and set(stable) <= set(kept)
    cls_of = {d["anchor"]: d["cls"] for d in ref[0]}
    rng = np.random.default_rng(0)
    for _ in range(6):
        # |d sigmoid| <= |d logit| / 4: logits move by up to 3.9 dc, confidences by less than dc
        pert = [c + rng.uniform(-3.9 * dc, 3.9 * dc, c.shape).astype(np.float32) for c in cls_maps]
        b2, s2 = oy.decode(box_maps, pert)
        assert np.abs(s2 - scores).max() < dc
        out = oy.non_max_suppression(b2, s2, 0.25)[0]
        got = {a: c for a, _, _, c in out}
        assert all(a in got and got[a] == cls_of[a] for a in stable)
