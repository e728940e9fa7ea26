"""GPU parity for the detection stage (through the C ABI) against oracle/yolo.py.

Bars (written here, as the tier asks):
  K3 letterbox            bit-exact (integer resize, one fp32 divide, one RNE to fp16)
  K4/K5 network           fp16 network vs the same fp16 network on torch-CPU: the fp32 accumulation
                          order differs and every one of the ~25 layers on a path rounds to fp16
                          (ulp 2**-11), so head logits drift by ~1e-3 of the map's RMS on average
                          (measured 0.9-1.7e-3) and < 1.2e-2 at worst; bars: max <= HEAD_MAX*rms,
                          mean <= HEAD_MEAN*rms
  K6+K7 decode + NMS      given IDENTICAL head maps: kept anchor indices, order and classes exact;
                          conf / box coordinates within BOX_RTOL (device expf vs numpy exp, ulps)
  end to end              same kept set whenever the oracle's own result is stable under a
                          perturbation of the size of the network tolerance
"""
import numpy as np
import pytest

from eioku_amd import detect as D, weights as W
from oracle import prng, yolo as oy

pytestmark = pytest.mark.gpu

HEAD_MAX, HEAD_MEAN = 0.025, 0.005
BOX_RTOL = 2e-5


def _dev(a, gpu):
    import torch

    return torch.from_numpy(np.ascontiguousarray(a)).to(gpu)


# ---------------------------------------------------------------------------------------------
# K3
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("h,w", [(480, 854), (1080, 1920), (720, 1280), (384, 640), (96, 160), (333, 517),
                                 (640, 640), (1280, 720), (200, 1000)])
def test_letterbox_bit_exact(gpu, h, w):
    rng = np.random.default_rng(h * 7 + w)
    frames = rng.integers(0, 256, (2, h, w, 3), dtype=np.uint8)
    got, plan = D.letterbox_f16(_dev(frames, gpu))
    got = got.cpu().numpy()
    want = oy.preprocess(frames).permute(0, 2, 3, 1).numpy().astype(np.float16)  # NHWC RGB
    assert got.shape == (2, plan.out_h, plan.out_w, 8)
    assert np.array_equal(got[..., :3], want)
    assert not got[..., 3:].any()


def test_letterbox_geometry_of_reference_sizes():
    p = D.letterbox_plan(1080, 1920)
    assert (p.new_h, p.new_w, p.top, p.left, p.out_h, p.out_w) == (360, 640, 12, 0, 384, 640)
    p = D.letterbox_plan(480, 854)
    assert (p.out_h, p.out_w, p.mode) == (384, 640, 1)
    p = D.letterbox_plan(720, 1280)
    assert p.mode == 2  # exact 1/2 scale -> area path
    p = D.letterbox_plan(640, 640)
    assert p.mode == 0 and (p.top, p.left) == (0, 0)


# ---------------------------------------------------------------------------------------------
# K4/K5: whole network
# ---------------------------------------------------------------------------------------------
def _check_heads(got, want, tag):
    for g, r in zip(got, want):
        g = g.cpu().numpy()
        err = np.abs(g - r)
        rms = float(np.sqrt((r.astype(np.float64) ** 2).mean()))
        assert g.shape == r.shape
        assert err.max() <= HEAD_MAX * rms, (tag, float(err.max()), rms)
        assert err.mean() <= HEAD_MEAN * rms, (tag, float(err.mean()), rms)


@pytest.mark.parametrize("variant,nc,n,h,w", [("n", 80, 2, 96, 160), ("n", 1, 1, 64, 96), ("s", 80, 1, 64, 64),
                                               ("m", 80, 1, 64, 96), ("n", 80, 1, 384, 640)])
def test_network_heads_match_fp16_oracle(gpu, variant, nc, n, h, w):
    import torch

    state = W.random_state(variant, nc, seed=7)
    det = D.Yolov8Detector(variant, nc, state)
    rng = np.random.default_rng(3)
    x = np.zeros((n, h, w, 8), dtype=np.float16)
    x[..., :3] = rng.random((n, h, w, 3)).astype(np.float16)
    box, cls = det.forward_raw(_dev(x, gpu))
    net = oy.Net(state, *W.YOLO_VARIANTS[variant], nc)
    rbox, rcls = net.forward(torch.from_numpy(x[..., :3].astype(np.float32)).permute(0, 3, 1, 2))
    _check_heads(box, rbox, (variant, "box"))
    _check_heads(cls, rcls, (variant, "cls"))
    assert det.last_conv_flops() > 0
    det.close()


# ---------------------------------------------------------------------------------------------
# K6 + K7 on identical head maps
# ---------------------------------------------------------------------------------------------
def _random_heads(seed, n, hl, wl, nc, cls_mu, cls_sigma):
    rng = np.random.default_rng(seed)
    box = [(2.0 * rng.standard_normal((n, h, w, 64))).astype(np.float32) for h, w in zip(hl, wl)]
    cls = [(cls_mu + cls_sigma * rng.standard_normal((n, h, w, nc))).astype(np.float32) for h, w in zip(hl, wl)]
    return box, cls


def _stable(boxes, scores, conf, iou, max_det, ref):
    """Is the oracle's own answer unchanged by ulp-scale perturbations of the two thresholds?"""
    for dc, di in ((1e-6, 0), (-1e-6, 0), (0, 1e-5), (0, -1e-5)):
        alt = oy.non_max_suppression(boxes, scores, conf + dc, iou + di, max_det)
        if [[a for a, *_ in per] for per in alt] != [[a for a, *_ in per] for per in ref]:
            return False
    return True


@pytest.mark.parametrize("seed,nc,cls_mu,cls_sigma,conf,max_det", [
    (1, 80, -6.0, 1.5, 0.25, 300),   # sparse, trained-model-like
    (2, 80, -3.0, 1.5, 0.25, 300),   # dense: thousands of candidates, max_det truncation
    (3, 1, -1.0, 1.5, 0.5, 300),     # single class (face): heavy same-class suppression
    (4, 80, -6.0, 1.5, 0.9, 300),    # nothing (or almost nothing) passes
    (5, 3, 0.0, 2.0, 0.001, 50),     # val-style low threshold: every anchor is a candidate
])
def test_decode_nms_indices_exact(gpu, seed, nc, cls_mu, cls_sigma, conf, max_det):
    n, hl, wl = 3, (48, 24, 12), (80, 40, 20)
    box, cls = _random_heads(seed, n, hl, wl, nc, cls_mu, cls_sigma)
    plan = D.letterbox_plan(480, 854)
    dets, counts = D.postprocess([_dev(b, gpu) for b in box], [_dev(c, gpu) for c in cls], plan, conf, 0.7, max_det)
    boxes, scores = oy.decode(box, cls)
    ref = oy.non_max_suppression(boxes, scores, conf, 0.7, max_det)
    if not _stable(boxes, scores, conf, 0.7, max_det, ref):
        pytest.skip("oracle result itself flips under ulp-scale threshold perturbation (ill-conditioned seed)")
    for i in range(n):
        k = int(counts[i])
        assert k == len(ref[i])
        assert list(dets["anchor"][i, :k]) == [a for a, *_ in ref[i]]
        assert list(dets["cls"][i, :k]) == [c for *_, c in ref[i]]
        if k:
            rconf = np.array([c for _, _, c, _ in ref[i]], dtype=np.float32)
            assert np.allclose(dets["conf"][i, :k], rconf, rtol=BOX_RTOL, atol=0)
            rb = np.stack([oy.scale_boxes(b, (plan.out_h, plan.out_w), (480, 854)) for _, b, _, _ in ref[i]])
            gb = np.stack([dets[f][i, :k] for f in ("x1", "y1", "x2", "y2")], -1)
            assert np.allclose(gb, rb, rtol=BOX_RTOL, atol=2e-3)


def test_nms_keeps_at_most_max_det_and_suppresses_duplicates(gpu):
    """Hand-built maps: identical strong boxes on neighbouring anchors collapse to one per class."""
    n, hl, wl, nc = 1, (8, 4, 2), (8, 4, 2), 2
    box = [np.zeros((n, h, w, 64), np.float32) for h, w in zip(hl, wl)]
    cls = [np.full((n, h, w, nc), -20.0, np.float32) for h, w in zip(hl, wl)]
    for b in box:  # DFL: put all mass on bin 4 for every side -> ltrb = 4 cells
        b.reshape(n, -1, 4, 16)[..., 4] = 30.0
    cls[0][0, 3, 3, 0] = 5.0
    cls[0][0, 3, 4, 0] = 4.0   # overlaps the first (IoU = 7/9 > 0.7) -> suppressed
    cls[0][0, 3, 4, 1] = -20.0
    cls[0][0, 6, 6, 1] = 3.0   # other class, far away -> kept
    plan = D.letterbox_plan(64, 64, imgsz=64)
    dets, counts = D.postprocess([_dev(b, gpu) for b in box], [_dev(c, gpu) for c in cls], plan, 0.25, 0.7, 300)
    assert counts[0] == 2
    assert list(dets["anchor"][0, :2]) == [3 * 8 + 3, 6 * 8 + 6]
    assert list(dets["cls"][0, :2]) == [0, 1]
    ref = oy.non_max_suppression(*oy.decode(box, cls), 0.25, 0.7, 300)
    assert [a for a, *_ in ref[0]] == [27, 54]
    # anchor centre (3.5,3.5) +- 4 cells at stride 8 = [-4,-4,60,60], clipped to the 64x64 frame
    assert np.allclose([dets["x1"][0, 0], dets["y1"][0, 0], dets["x2"][0, 0], dets["y2"][0, 0]], [0, 0, 60, 60])


# ---------------------------------------------------------------------------------------------
# end to end
# ---------------------------------------------------------------------------------------------
def _calibrated_state(frames, variant="n", nc=80, seed=7, frac=0.015):
    """Random weights whose Detect logits are O(1) on THESE frames (a random net's logit scale depends
    on its input): rescale the six output convs so box logits have std 2 and class logits std 3,
    shifted so that about `frac` of the anchors pass conf 0.25 and none saturates."""
    state = W.random_state(variant, nc, seed=seed)
    box, cls = oy.Net(state, *W.YOLO_VARIANTS[variant], nc).forward(oy.preprocess(frames))
    scaled = [(c - c.mean()) * (3.0 / c.std()) for c in cls]
    top = np.concatenate([c.max(axis=-1).reshape(-1) for c in scaled])
    shift = float(np.log(0.25 / 0.75) - np.quantile(top, 1.0 - frac))
    for l in range(3):
        w, b = state[f"model.22.cv2.{l}.2"]
        state[f"model.22.cv2.{l}.2"] = ((w * (2.0 / box[l].std())).astype(np.float32), (b * 0).astype(np.float32))
        w, b = state[f"model.22.cv3.{l}.2"]
        sc = 3.0 / cls[l].std()
        state[f"model.22.cv3.{l}.2"] = ((w * sc).astype(np.float32),
                                        ((b - cls[l].mean()) * sc + shift).astype(np.float32))
    return state


def test_detect_end_to_end_vs_oracle(gpu):
    frames = prng.synth_frames_bgr(21, 2, 240, 427)
    state = _calibrated_state(frames)
    det = D.Yolov8Detector("n", 80, state)
    conf = 0.25
    dets, counts = det.detect(_dev(frames, gpu), conf=conf)
    dets_h, counts_h = det.detect(frames, conf=conf)  # host staging path gives the same bytes
    assert np.array_equal(counts, counts_h) and np.array_equal(dets, dets_h)
    net = oy.Net(state, *W.YOLO_VARIANTS["n"], 80)
    ref, (_, _, boxes, scores) = oy.detect(net, frames, conf)
    # Network drift (HEAD_MAX) moves scores by ~1e-2 and boxes by a fraction of a pixel.  On smooth
    # frames neighbouring anchors carry near-identical boxes and scores, so WHICH of them wins NMS
    # is ill-conditioned; the detection it stands for is not.  Bar: every oracle detection with more
    # conf margin than the drift has a HIP detection of the same class with IoU >= 0.7 (the NMS radius) and conf
    # within 0.03, and vice versa, for at least 80 % of them (NMS chains near IoU 0.7 flip too).  (Exact index parity is asserted on identical head maps above.)
    def iou(a, b):
        iw = max(0.0, min(a[2], b[2]) - max(a[0], b[0]))
        ih = max(0.0, min(a[3], b[3]) - max(a[1], b[1]))
        u = (a[2] - a[0]) * (a[3] - a[1]) + (b[2] - b[0]) * (b[3] - b[1]) - iw * ih
        return iw * ih / u if u > 0 else 0.0

    for i in range(2):
        got = [(np.array([dets[f][i, k] for f in ("x1", "y1", "x2", "y2")], dtype=np.float64),
                float(dets["conf"][i, k]), int(dets["cls"][i, k])) for k in range(counts[i])]
        want = [(d["xyxy"].astype(np.float64), d["conf"], d["cls"]) for d in ref[i]]
        assert len(want) >= 5, "calibration should leave the oracle some detections"

        def matched(x, pool):
            return any(c == x[2] and iou(x[0], b) >= 0.7 and abs(s - x[1]) < 0.03 for b, s, c in pool)

        sure_w = [x for x in want if x[1] > conf + 0.03]
        sure_g = [x for x in got if x[1] > conf + 0.03]
        assert sum(matched(x, got) for x in sure_w) >= 0.8 * len(sure_w), (len(sure_w), len(got))
        assert sum(matched(x, want) for x in sure_g) >= 0.8 * len(sure_g), (len(sure_g), len(want))
    det.close()


@pytest.mark.parametrize("h,w", [(470, 640), (640, 470), (640, 472), (940, 1280), (1280, 940), (96, 160)])
def test_fused_stem_equals_letterbox_then_network(gpu, h, w):
    """detect() lets the stem read the BGR frames itself in the copy / exact-half letterbox modes (the fp16
    network input is never materialised); K3 -> eioku_yolo_forward -> K6/K7 is the unfused route.  Same
    arithmetic, so the detections must be the same BYTES (114 padding rows, image borders and the bilinear
    mode, which stays unfused, included)."""
    frames = _dev(prng.synth_frames_bgr(33, 3, h, w), gpu)
    det = D.Yolov8Detector("n", 80, W.random_state("n", 80, seed=3))
    det.calibrate_random_head(frames, frac=0.02)
    dets, counts = det.detect(frames, conf=0.25)
    x, plan = D.letterbox_f16(frames)
    box, cls = det.forward_raw(x)
    dets2, counts2 = D.postprocess(box, cls, plan, 0.25, 0.7, 300)
    assert counts.sum() > 0
    assert np.array_equal(counts, counts2)
    for i in range(len(counts)):
        assert np.array_equal(dets[i, :counts[i]], dets2[i, :counts[i]])
    # a threshold that lets a large share of the anchors through: detect() evaluates the box branch's last conv
    # per passing anchor (gathered rows) and reads {max logit, argmax} words -- still the dense route's bytes
    dets, counts = det.detect(frames, conf=0.02, max_det=1000)
    dets2, counts2 = D.postprocess(box, cls, plan, 0.02, 0.7, 1000)
    assert counts.min() > 20 and np.array_equal(counts, counts2)
    for i in range(len(counts)):
        assert np.array_equal(dets[i, :counts[i]], dets2[i, :counts[i]])
    det.close()


@pytest.mark.parametrize("variant", ["n", "s"])
def test_fused_3x3_1x1_pairs_are_bit_identical_to_separate_launches(gpu, variant, tmp_path):
    """model.1 -> model.2.cv1 and model.3 -> model.4.cv1 run as ONE launch (the 1x1 consumes the 3x3's tile on chip),
    the shallow C2f Bottlenecks (3x3 -> 3x3 + x, 16 / 32 channels) run as one launch with the intermediate in LDS
    (EIOKU_CONV_CHAIN; the 16-channel C2f's closing 1x1 joins that launch, EIOKU_CHAIN_CAT), and the neck's 1x1 convs read the half-resolution tensor in place instead of an upsampled copy.
    Same fp16 rounding of the intermediate, same MFMA k order: the Detect maps must equal, bit for bit, those of a
    process that runs every layer as its own launch (EIOKU_CONV_POST=0; the switch is read once per process)."""
    import os
    import subprocess
    import sys

    code = (
        "import sys, numpy as np, torch\n"
        "sys.path.insert(0, %r)\n"
        "from eioku_amd import detect as D, weights as W\n"
        "from oracle import prng\n"
        "f = torch.from_numpy(prng.synth_frames_bgr(41, 3, 96, 160)).cuda()\n"
        "det = D.Yolov8Detector(%r, 80, W.random_state(%r, 80, seed=9))\n"
        "x, _ = D.letterbox_f16(f)\n"
        "box, cls = det.forward_raw(x)\n"
        "np.savez(sys.argv[1], *[t.cpu().numpy() for t in box + cls])\n"
    ) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), variant, variant)
    outs = {}
    for flag in ("1", "0"):
        path = tmp_path / f"heads_{flag}.npz"
        # every graph-level fusion on / off
        # (EIOKU_CHAIN_CAT: 2 = every C2f whose closing 1x1 can join its last Bottleneck, the default; 0 = none)
        env = dict(os.environ, EIOKU_CONV_POST=flag, EIOKU_UP_FUSE=flag, EIOKU_CONV_CHAIN=flag,
                   EIOKU_CHAIN_CAT="2" if flag == "1" else "0")
        subprocess.run([sys.executable, "-c", code, str(path)], check=True, env=env, timeout=300)
        with np.load(path) as z:
            outs[flag] = [z[k] for k in z.files]
    assert len(outs["1"]) == 6
    for a, b in zip(outs["1"], outs["0"]):
        assert a.dtype == np.float32 and np.array_equal(a, b)


def test_full_size_batch_split_invariance_and_pipelined_order(gpu):
    """BASELINE cfg2 size (64 x 640 x 640), size-independent properties: a frame's detections do not depend on
    which frames share its batch (the flattened-pixel kernels cut tiles ACROSS frame borders, the persistent ones
    walk tiles grid-stride over the whole batch), and PipelinedDetector (two handles, two streams) returns, in
    submission order, exactly what the synchronous detector returns -- for device and host inputs."""
    import torch

    frames = _dev(prng.synth_frames_bgr(77, 64, 640, 640), gpu)
    det = D.Yolov8Detector("n", 80, W.random_state("n", 80, seed=5))
    det.calibrate_random_head(frames[:8], frac=0.01)
    dets, counts = det.detect(frames, conf=0.25)
    assert counts.sum() > 64
    for lo, hi in [(0, 1), (1, 24), (24, 64)]:
        d2, c2 = det.detect(frames[lo:hi].contiguous(), conf=0.25)
        assert np.array_equal(c2, counts[lo:hi])
        for i in range(hi - lo):
            assert np.array_equal(d2[i, :c2[i]], dets[lo + i, :counts[lo + i]])
    pipe = D.PipelinedDetector(det, depth=2)
    chunks = [(0, 16), (16, 48), (48, 64), (0, 64)]
    for k, (lo, hi) in enumerate(chunks):
        batch = frames[lo:hi].contiguous()
        pipe.submit(batch.cpu().numpy() if k % 2 else batch, conf=0.25)
    for lo, hi in chunks:
        d2, c2 = pipe.result()
        assert np.array_equal(c2, counts[lo:hi])
        for i in range(hi - lo):
            assert np.array_equal(d2[i, :c2[i]], dets[lo + i, :counts[lo + i]])
    assert pipe.in_flight() == 0
    pipe.close()


def test_detect_empty_batch_and_missing_weights(gpu):
    import torch
    from eioku_amd._lib import EiokuHipError

    det = D.Yolov8Detector("n", 80, W.random_state("n", 80, seed=1))
    d, c = det.detect(torch.empty((0, 64, 64, 3), dtype=torch.uint8, device=gpu))
    assert d.shape == (0, 300) and c.shape == (0,)
    det.close()
    bare = D.Yolov8Detector("n", 80, None)
    with pytest.raises(EiokuHipError):
        bare.detect(np.zeros((1, 64, 64, 3), dtype=np.uint8))
    bare.close()


def test_batch_beyond_32bit_offsets_is_refused_not_corrupted(gpu):
    """The conv kernels index activations with 32-bit element offsets: a batch whose tensors would overflow them must
    fail loudly (split the batch) instead of reading out of bounds."""
    import torch

    from eioku_amd._lib import EiokuHipError

    det = D.Yolov8Detector("n", 80, W.random_state("n", 80, seed=1))
    frames = torch.zeros((140000, 64, 64, 3), dtype=torch.uint8, device="cuda")  # 140000 x 64 x 64 x 8 ch >= 2^31 elements
    with pytest.raises(EiokuHipError, match="split the batch"):
        det.detect(frames, conf=0.25)
    dets, counts = det.detect(frames[:3], conf=0.25)  # the handle is still usable
    assert counts.shape == (3,)
    det.close()
