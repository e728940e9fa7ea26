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
  end to end              every MARGIN-STABLE detection (tests/wellcond.py: decided by more than the measured
                          head drift) is kept by both sides with the same anchor index, class and order -
                          exact, both directions, >= 20 of them per frame; objects (nc = 80) and faces (nc = 1)
  fp32 validation         the HIP heads also stay within FP32_MAX / FP32_MEAN of the reference's own fp32
                          arithmetic (Ultralytics half=False) run by the oracle's fp32 mode
"""
import numpy as np
import pytest

import wellcond
from eioku_amd import detect as D, weights as W
from oracle import prng, yolo as oy

pytestmark = pytest.mark.gpu

HEAD_MAX, HEAD_MEAN = 0.025, 0.005
FP32_MAX, FP32_MEAN = 0.05, 0.01  # vs the reference's fp32 arithmetic (tests/test_oracle_pins.py bounds the oracle's own)
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
                                               ("m", 80, 1, 64, 96), ("n", 80, 1, 384, 640),
                                               # BASELINE cfg4 / the live config at the network size every 16:9 source
                                               # letterboxes to: YOLOv8m objects, yolov8n-face, yolov8s (content_creator.json)
                                               ("m", 80, 1, 384, 640), ("n", 1, 1, 384, 640), ("s", 80, 1, 384, 640),
                                               ("m", 80, 2, 640, 640)])
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
    if (h, w) == (384, 640) and variant in ("n", "m"):
        # fp32 validation mode: the reference predicts with half=False; same weights, fp32 everywhere
        f32 = oy.Net(state, *W.YOLO_VARIANTS[variant], nc, fp16=False)
        fbox, fcls = f32.forward(torch.from_numpy(x[..., :3].astype(np.float32)).permute(0, 3, 1, 2))
        for g, r in zip(box + cls, fbox + fcls):
            err = np.abs(g.cpu().numpy() - r)
            rms = float(np.sqrt((r.astype(np.float64) ** 2).mean()))
            assert err.max() <= FP32_MAX * rms and err.mean() <= FP32_MEAN * rms, (variant, float(err.max() / rms))
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
    plan = D.letterbox_plan(480, 854)
    # well-conditioned by construction: a draw whose oracle answer would flip under an ulp-scale nudge of either
    # threshold (a confidence within 1e-6 of `conf`, an IoU within 1e-5 of 0.7) says nothing about the kernels;
    # walk a deterministic seed sequence to the first draw that has no such pair (almost always the first)
    for attempt in range(8):
        box, cls = _random_heads(seed + 1000 * attempt, n, hl, wl, nc, cls_mu, cls_sigma)
        boxes, scores = oy.decode(box, cls)
        ref = oy.non_max_suppression(boxes, scores, conf, 0.7, max_det)
        if _stable(boxes, scores, conf, 0.7, max_det, ref):
            break
    else:
        pytest.fail("no well-conditioned draw in 8 seeds: the conditioning check itself is broken")
    dets, counts = D.postprocess([_dev(b, gpu) for b in box], [_dev(c, gpu) for c in cls], plan, conf, 0.7, max_det)
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
@pytest.mark.parametrize("variant,nc,h,w,seed,min_stable", [
    ("n", 80, 270, 480, 5, 20),     # objects (detect_objects' default model), bilinear letterbox
    ("n", 1, 270, 480, 6, 20),      # faces: yolov8n-face, one class -> every candidate competes in NMS
    ("n", 80, 1080, 1920, 5, 20),   # the reference's 1080p sources (scale 3 letterbox)
    ("n", 1, 480, 854, 5, 20),      # BASELINE cfg1's 480p clip
    ("m", 80, 1080, 1920, 6, 3),    # cfg4's object model: a deeper random net leaves fewer margin-decided detections
])
@pytest.mark.parametrize("oracle_fp16", [True, False], ids=["vs_fp16_oracle", "vs_fp32_reference_precision"])
def test_detect_end_to_end_exact_on_margin_stable_detections(gpu, variant, nc, h, w, seed, min_stable, oracle_fp16):
    """detect() end to end (letterbox -> network -> decode -> NMS -> scale_boxes) against the oracle, BASELINE's bar:
    post-NMS indices exact.  Scope: the margin-stable detections of tests/wellcond.py, with the drift bounds MEASURED
    in this run (oracle decode of the HIP head maps vs of the oracle head maps) and doubled.  Both directions: what
    the oracle keeps stably the HIP path keeps, and what the HIP path keeps stably (judged on ITS maps) the oracle
    keeps; same class; and the stable anchors appear in the same order on both sides.

    oracle_fp16 = False: the oracle is the fp32 network - the reference's own precision
    (model_manager.py:270-291: Ultralytics predicts with half=False) - so the drift is fp16-build vs reference
    arithmetic; it is measured the same way and doubled, the assertions are the same, the floor on the number of
    margin-stable detections is lower (>= 10 per frame for v8n / face) because the margins are wider."""
    conf = 0.25
    frames = wellcond.blob_frames(seed, 1, h, w)
    state = wellcond.calibrated_state(frames, variant, nc, seed=7, frac=0.08, conf=conf)
    det = D.Yolov8Detector(variant, nc, state)
    dets, counts = det.detect(_dev(frames, gpu), conf=conf)
    dets_h, counts_h = det.detect(frames, conf=conf)  # host staging path gives the same bytes
    assert np.array_equal(counts, counts_h) and np.array_equal(dets, dets_h)
    net = oy.Net(state, *W.YOLO_VARIANTS[variant], nc, fp16=oracle_fp16)
    ref, (_, _, boxes_o, scores_o) = oy.detect(net, frames, conf)
    if not oracle_fp16:
        min_stable = 10 if variant == "n" else 3  # measured: 18-41 (v8n / face), 14 (v8m)
    # the HIP side's own head maps, decoded by the oracle: its view of every anchor
    x, plan = D.letterbox_f16(_dev(frames, gpu))
    hb, hc = det.forward_raw(x)
    boxes_g, scores_g = oy.decode([t.cpu().numpy() for t in hb], [t.cpu().numpy() for t in hc])
    dc, du = wellcond.drift(boxes_o[0], scores_o[0], boxes_g[0], scores_g[0], conf)
    # the network tolerance (HEAD_MAX / FP32_MAX) seen through sigmoid / IoU
    assert (dc < 0.03 and du < 0.08) if oracle_fp16 else (dc < 0.04 and du < 0.08), (dc, du)
    dc, du = 2 * dc + 1e-6, 2 * du + 1e-6
    k = int(counts[0])
    got = [int(a) for a in dets["anchor"][0, :k]]
    got_cls = {int(a): int(c) for a, c in zip(dets["anchor"][0, :k], dets["cls"][0, :k])}
    want = [d["anchor"] for d in ref[0]]
    want_cls = {d["anchor"]: d["cls"] for d in ref[0]}
    # the HIP path's post-processing of ITS maps is exactly the oracle's post-processing of those maps
    ref_g = oy.non_max_suppression(boxes_g, scores_g, conf)
    if wellcond.stably_kept(boxes_g[0], scores_g[0], [a for a, *_ in ref_g[0]], conf, 1e-6, 1e-5) == [a for a, *_ in ref_g[0]]:
        assert got == [a for a, *_ in ref_g[0]]
    stable_o = wellcond.stably_kept(boxes_o[0], scores_o[0], want, conf, dc, du)
    stable_g = wellcond.stably_kept(boxes_g[0], scores_g[0], got, conf, dc, du)
    report = (f"margin-stable: {len(stable_o)} of the oracle's {len(want)} detections ({len(stable_o) / max(len(want), 1):.0%}), "
              f"{len(stable_g)} of the HIP path's {len(got)}; drift bounds used: conf {dc:.4f}, IoU {du:.4f}; "
              f"oracle {'fp16' if oracle_fp16 else 'fp32'}")
    print(report)
    assert len(stable_o) >= min_stable and len(stable_g) >= min_stable, report
    assert [a for a in stable_o if a not in got] == [], "a margin-stable oracle detection is missing from the HIP result"
    assert [a for a in stable_g if a not in want] == [], "a margin-stable HIP detection is missing from the oracle result"
    both = [a for a in stable_o if a in set(stable_g)]
    assert all(got_cls[a] == want_cls[a] for a in both)
    # order: anchors whose confidences differ by more than the drift keep their relative order
    co = scores_o[0].max(1)
    pos_g = {a: i for i, a in enumerate(got)}
    pos_o = {a: i for i, a in enumerate(want)}
    for i, a in enumerate(both):
        for b in both[i + 1:]:
            if abs(float(co[a]) - float(co[b])) > 2 * dc:
                assert (pos_g[a] < pos_g[b]) == (pos_o[a] < pos_o[b]), (a, b)
    # and their boxes / confidences agree to the drift (original-frame pixels)
    for a in both:
        d = next(x for x in ref[0] if x["anchor"] == a)
        j = pos_g[a]
        gb = np.array([dets[f][0, j] for f in ("x1", "y1", "x2", "y2")], np.float64)
        assert abs(float(dets["conf"][0, j]) - d["conf"]) <= dc
        assert np.abs(gb - d["xyxy"]).max() <= 0.02 * max(h, w)
    det.close()


def test_lazy_and_dense_box_branch_give_identical_detections(gpu, tmp_path):
    """detect() evaluates the box branch's 3x3 layers lazily (only around passing anchors) when the previous
    call's pass rate was low, densely otherwise; the choice depends on history that arrives asynchronously.  Both
    routes must produce the same bytes: force each in its own process (the switch is read once) on two consecutive
    calls (the second call is the one that can go lazy) and compare."""
    import os
    import subprocess
    import sys

    code = (
        "import sys, numpy as np, torch\n"
        "sys.path.insert(0, %r)\n"
        "from eioku_amd import detect as D, weights as W\n"
        "from oracle import prng\n"
        "f = torch.from_numpy(prng.synth_frames_bgr(41, 4, 240, 427)).cuda()\n"
        "det = D.Yolov8Detector('n', 80, W.random_state('n', 80, seed=9))\n"
        "det.calibrate_random_head(f, frac=0.01)\n"
        "out = []\n"
        "for _ in range(3):\n"
        "    d, c = det.detect(f, conf=0.25)\n"
        "    torch.cuda.synchronize()\n"
        "    out += [d.view(np.uint8).reshape(len(c), -1)[i, :32 * c[i]] for i in range(len(c))] + [c]\n"
        "np.savez(sys.argv[1], *out)\n"
    ) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))),)
    outs = {}
    for frac in ("0", "0.5"):  # never lazy-deep / lazy-deep whenever <= 50 % of the anchors passed last time
        path = tmp_path / f"dets_{frac}.npz"
        subprocess.run([sys.executable, "-c", code, str(path)], check=True, env=dict(os.environ, EIOKU_LAZY_DEEP_FRAC=frac),
                       timeout=300)
        with np.load(path) as z:
            outs[frac] = [z[k] for k in z.files]
    assert len(outs["0"]) == len(outs["0.5"]) and sum(int(a.size) for a in outs["0"]) > 15 * 32
    for a, b in zip(outs["0"], outs["0.5"]):
        assert np.array_equal(a, b)


def test_detect_fusion_switches_leave_the_detections_byte_identical(gpu, tmp_path):
    """Every detect()-level fusion the library keeps a switch for (each read once per process): the one-launch
    YOLOv8n front end (EIOKU_STEM_CHAIN), the stem reading the BGR frames itself (EIOKU_STEM_FUSE), the class head
    writing {max logit, argmax} words instead of 80 logits (EIOKU_CLSMAX) and the lazily evaluated box branch
    (EIOKU_LAZY_BOX), plus the XCD-contiguous tile order of the persistent conv kernels (EIOKU_XCD_TILES, r3).  640-wide
    copy-mode sources so that all of them are eligible; the detections of every variant must be the all-on bytes."""
    import os
    import subprocess
    import sys

    code = (
        "import sys, numpy as np, torch\n"
        "sys.path.insert(0, %r)\n"
        "from eioku_amd import detect as D, weights as W\n"
        "from oracle import prng\n"
        "f = torch.from_numpy(prng.synth_frames_bgr(43, 4, 480, 640)).cuda()\n"
        "det = D.Yolov8Detector('n', 80, W.random_state('n', 80, seed=9))\n"
        "det.calibrate_random_head(f, frac=0.01)\n"
        "out = []\n"
        "for _ in range(2):\n"
        "    d, c = det.detect(f, conf=0.25)\n"
        "    torch.cuda.synchronize()\n"
        "    out += [d.view(np.uint8).reshape(len(c), -1)[i, :32 * c[i]] for i in range(len(c))] + [c]\n"
        "np.savez(sys.argv[1], *out)\n"
    ) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))),)
    outs = {}
    for name in ("all_on", "EIOKU_STEM_CHAIN", "EIOKU_STEM_FUSE", "EIOKU_CLSMAX", "EIOKU_LAZY_BOX", "EIOKU_XCD_TILES"):
        path = tmp_path / f"dets_{name}.npz"
        env = dict(os.environ) if name == "all_on" else dict(os.environ, **{name: "0"})
        subprocess.run([sys.executable, "-c", code, str(path)], check=True, env=env, timeout=300)
        with np.load(path) as z:
            outs[name] = [z[k] for k in z.files]
    assert sum(int(a.size) for a in outs["all_on"]) > 15 * 32
    for name, got in outs.items():
        assert len(got) == len(outs["all_on"]), name
        for a, b in zip(got, outs["all_on"]):
            assert np.array_equal(a, b), name


@pytest.mark.parametrize("h,w", [(470, 640), (640, 470), (640, 472), (940, 1280), (1280, 940), (96, 160),
                                 (1080, 1920), (1920, 1080), (960, 1920), (1605, 1920)])  # exact 1/3: a decimating copy
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
    # the opt-in register-stationary Bottleneck pairs (k_conv3x3_pair_rs: 64-channel pairs at 40 x 40): same bytes too
    path = tmp_path / "heads_rs.npz"
    env = dict(os.environ, EIOKU_CONV_PAIR_RS="1")
    subprocess.run([sys.executable, "-c", code, str(path)], check=True, env=env, timeout=300)
    with np.load(path) as z:
        for k, b in zip(z.files, outs["0"]):
            assert np.array_equal(z[k], b)


@pytest.mark.parametrize("variant,h,w", [("n", 640, 640), ("m", 1080, 1920), ("n", 1080, 1920)])
def test_full_size_batch_split_invariance_and_pipelined_order(gpu, variant, h, w):
    """BASELINE cfg2 size (64 x 640 x 640, YOLOv8n) and cfg4's shape (64 x 1080p sources -> bilinear letterbox ->
    YOLOv8m / YOLOv8n at 384 x 640), size-independent properties: a frame's detections do not depend on which frames
    share its batch (the flattened-pixel kernels cut tiles ACROSS frame borders, the persistent ones walk tiles
    grid-stride over the whole batch), detect() equals the unfused route K3 -> network -> K6/K7 byte for byte, and
    PipelinedDetector (two handles, two streams) returns, in submission order, exactly what the synchronous detector
    returns -- for device and host inputs."""
    import torch

    frames = _dev(prng.synth_frames_bgr(77, 64, h, w), gpu)
    det = D.Yolov8Detector(variant, 80, W.random_state(variant, 80, seed=5))
    det.calibrate_random_head(frames[:8], frac=0.01)
    dets, counts = det.detect(frames, conf=0.25)
    assert counts.sum() > 64
    for lo, hi in [(0, 1), (1, 24), (24, 64)]:
        d2, c2 = det.detect(frames[lo:hi].contiguous(), conf=0.25)
        assert np.array_equal(c2, counts[lo:hi])
        for i in range(hi - lo):
            assert np.array_equal(d2[i, :c2[i]], dets[lo + i, :counts[lo + i]])
    x, plan = D.letterbox_f16(frames[:8].contiguous())
    assert (plan.out_h, plan.out_w) == ((640, 640) if h == 640 else (384, 640))
    d3, c3 = D.postprocess(*det.forward_raw(x), plan, 0.25, 0.7, 300)
    assert np.array_equal(c3, counts[:8])
    for i in range(8):
        assert np.array_equal(d3[i, :c3[i]], dets[i, :counts[i]])
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
