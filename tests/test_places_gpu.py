"""GPU parity of K11 (Places365 ResNet18, csrc/resnet.hip) against oracle/places.py: Pillow's resize bit for bit, the
ReLU epilogues of the 3x3 kernels on ResNet's shapes, the network's logits inside the fp16-network tolerance of both
the fp16 twin and the fp32 (reference-precision) oracle, softmax / sort, and classify() end to end."""
import ctypes as C

import numpy as np
import pytest

from oracle import places as op
from eioku_amd import _lib, places
from eioku_amd._buffers import ptr

pytestmark = pytest.mark.gpu

HEAD_MEAN, HEAD_MAX = 5e-3, 2.5e-2        # of the logits' RMS, vs the fp16 twin (the detector's bars)
FP32_MEAN, FP32_MAX = 1e-2, 5e-2          # vs the fp32 network: the documented fp16-vs-reference bound


def textured(seed, n, h, w):
    rng = np.random.default_rng(seed)
    blobs = rng.integers(0, 256, (n, h // 8 + 1, w // 8 + 1, 3))
    f = np.repeat(np.repeat(blobs, 8, 1), 8, 2)[:, :h, :w] + rng.integers(-20, 21, (n, h, w, 3))
    return np.clip(f, 0, 255).astype(np.uint8)


@pytest.fixture(scope="module")
def model(gpu):
    st = op.random_state(3)
    clf = places.Places365Classifier(st)
    yield st, clf
    clf.close()


@pytest.mark.parametrize("h,w", [(1080, 1920), (480, 854), (224, 224), (300, 200), (97, 640)])
def test_resize_is_pillows_bit_for_bit_and_normalise_is_torchs(gpu, model, h, w):
    import torch

    _, clf = model
    frames = textured(h + w, 2, h, w)
    for src in (frames, torch.from_numpy(frames).to(gpu)):
        x, u8 = clf.preprocess(src)
        want_u8 = np.stack([op.pil_resize_bilinear(np.ascontiguousarray(f[..., ::-1])) for f in frames])
        assert np.array_equal(u8.cpu().numpy(), want_u8)
        want = op.preprocess(frames).permute(0, 2, 3, 1).numpy().astype(np.float16)
        got = x.cpu().numpy()
        assert np.array_equal(got[..., :3], want) and not got[..., 3].any()


@pytest.mark.parametrize("cin,cout,hw,stride,act,res", [
    (64, 64, 56, 1, 2, False), (64, 64, 56, 1, 3, True), (64, 128, 56, 2, 2, False), (128, 128, 28, 1, 3, True),
    (128, 256, 28, 2, 2, False), (256, 256, 14, 1, 3, True), (256, 512, 14, 2, 2, False), (512, 512, 7, 1, 3, True),
    (64, 128, 56, 2, 0, False),   # a downsample branch: 1x1 / s2 run as a centre-tap 3x3 / s2
])
def test_resnet_block_convolutions_with_relu_epilogues(gpu, built_lib, cin, cout, hw, stride, act, res):
    """K4's kernels on ResNet18's shapes with the two new epilogues: act 2 = relu(conv + b), act 3 = relu(fp16(conv + b)
    + residual) (torchvision's BasicBlock order), against torch-CPU on the same fp16-rounded operands."""
    import torch
    import torch.nn.functional as F

    rng = np.random.default_rng(cin * 7 + cout + hw)
    n = 3
    x = rng.standard_normal((n, hw, hw, cin)).astype(np.float16)
    centre_only = act == 0
    w = (rng.standard_normal((cout, cin, 3, 3)) / np.sqrt(cin * (1 if centre_only else 9))).astype(np.float32)
    if centre_only:
        w[:, :, [0, 0, 0, 1, 1, 2, 2, 2], [0, 1, 2, 0, 2, 0, 1, 2]] = 0
    b = (0.1 * rng.standard_normal(cout)).astype(np.float32)
    ho = (hw + 2 - 3) // stride + 1
    r = rng.standard_normal((n, ho, ho, cout)).astype(np.float16) if res else None
    xd = torch.from_numpy(x).to(gpu)
    rd = torch.from_numpy(r).to(gpu) if res else None
    out = torch.empty((n, ho, ho, cout), dtype=torch.float16, device=gpu)
    _lib.check(built_lib.eioku_conv2d_f16(ptr(xd), n, hw, hw, cin, 0, cin, ptr(w), ptr(b), cout, 3, stride, act,
                                          ptr(rd), cout if res else 0, 0, ptr(out), cout, 0, None, None), "eioku_conv2d_f16")
    y = F.conv2d(torch.from_numpy(x).float().permute(0, 3, 1, 2), torch.from_numpy(w).half().float(), torch.from_numpy(b),
                 stride=stride, padding=1)
    if centre_only:  # the same numbers as the 1x1 / stride-2 convolution it stands for
        y1 = F.conv2d(torch.from_numpy(x).float().permute(0, 3, 1, 2), torch.from_numpy(w[:, :, 1:2, 1:2]).half().float(),
                      torch.from_numpy(b), stride=stride, padding=0)
        assert torch.allclose(y, y1, atol=1e-5)
    if act == 2:
        y = F.relu(y)
    y = y.half().float()
    if res:
        y = F.relu((y + torch.from_numpy(r).float().permute(0, 3, 1, 2)).half().float())
    want = y.permute(0, 2, 3, 1).numpy()
    got = out.cpu().numpy().astype(np.float32)
    rms = float(np.sqrt((want.astype(np.float64) ** 2).mean()))
    assert np.abs(got - want).max() <= 4e-3 * max(rms, 1e-3) + 2e-3 * np.abs(want).max()  # fp16 ulps of the largest values
    if act in (2, 3):
        assert got.min() >= 0 and (got == 0).mean() > 0.2


def test_network_logits_against_the_fp16_twin_and_the_fp32_reference_precision(gpu, model):
    import torch

    st, clf = model
    frames = textured(11, 4, 270, 480)
    x = op.preprocess(frames)
    xin, _ = clf.preprocess(torch.from_numpy(frames).to(gpu))
    got = clf.forward_raw(xin).cpu().numpy()
    for fp16, mean_bar, max_bar in ((True, HEAD_MEAN, HEAD_MAX), (False, FP32_MEAN, FP32_MAX)):
        want = op.ResNet18(st, fp16=fp16).logits(x)
        rms = float(np.sqrt((want.astype(np.float64) ** 2).mean()))
        err = np.abs(got - want)
        assert rms > 0.3 and err.mean() <= mean_bar * rms and err.max() <= max_bar * rms, (fp16, err.mean() / rms, err.max() / rms)
    assert clf.last_flops() == pytest.approx(4 * 3.63e9, rel=0.02)  # torchvision's resnet18: 1.8 GMACs per 224^2 frame


def test_classify_end_to_end_top_k_exact_where_the_logits_separate(gpu, model):
    """classify() = resize + network + softmax + sort.  Probabilities against torch's softmax of the device's own logits
    (1e-6), the order against the oracle's on the same logits (exact), and against the fp32 reference-precision network
    on every rank whose logit margin exceeds twice the measured drift (the margin-stable ranks)."""
    import torch

    st, clf = model
    frames = textured(12, 6, 480, 854)
    for src in (frames, torch.from_numpy(frames).to(gpu)):
        probs, cls, logits = clf.classify(src, top_k=10, with_logits=True)
        ref = op.top_predictions(logits, 10)
        for i in range(len(frames)):
            assert [c for c, _ in ref[i]] == list(cls[i])
            assert np.allclose([p for _, p in ref[i]], probs[i], rtol=2e-6, atol=1e-9)
            assert np.all(np.diff(probs[i]) <= 0)
    l32 = op.ResNet18(st, fp16=False).logits(op.preprocess(frames))
    drift = float(np.abs(logits - l32).max())
    stable = total = 0
    for i in range(len(frames)):
        order = np.argsort(-l32[i], kind="stable")
        srt = l32[i][order]
        for r in range(10):
            total += 1
            lo = srt[r] - srt[r + 1]
            hi = srt[r - 1] - srt[r] if r else np.inf
            if min(lo, hi) > 2 * drift:
                stable += 1
                assert cls[i][r] == order[r], (i, r)
    assert stable >= total // 4, f"{stable} of {total} ranks are margin-stable (drift {drift:.4f})"
    # the full sort: top_k = 365 returns a permutation whose probabilities sum to 1
    probs, cls = clf.classify(frames[:1], top_k=365)
    assert sorted(cls[0]) == list(range(365)) and abs(float(probs[0].sum()) - 1.0) < 1e-5


def test_model_manager_classify_places_on_the_device(gpu, tmp_path):
    """The drop-in end to end on the HIP path: frames from a FrameSource, random-init weights (no checkpoint offline),
    the reference's dict shape; equals classify() on the sampled frames."""
    import asyncio

    from eioku_amd.frames import FrameSource
    from eioku_amd.model_manager import ModelManager

    frames = textured(13, 9, 120, 160)

    class Src(FrameSource):
        def __init__(self):
            self.fps, self.total_frames, self.pos = 2.0, len(frames), 0

        def read(self):
            if self.pos >= len(frames):
                return False, None
            self.pos += 1
            return True, frames[self.pos - 1]

        def grab(self):
            if self.pos >= len(frames):
                return False
            self.pos += 1
            return True

    mm = ModelManager(cache_dir=str(tmp_path), frame_source=lambda p: Src(), random_init_seed=3, batch_size=2)
    out = asyncio.run(mm.classify_places("/videos/x.mp4", {"frame_interval": 2, "top_k": 3}))
    cl = out["classifications"]
    assert [c["frame_index"] for c in cl] == [0, 4, 8] and [c["timestamp_ms"] for c in cl] == [0, 2000, 4000]
    clf = places.Places365Classifier(places.random_state(3))
    probs, cls = clf.classify(frames[[0, 4, 8]], 3)
    for c, p, k in zip(cl, probs, cls):
        assert [d["label"] for d in c["predictions"]] == [f"place_{int(j)}" for j in k]
        assert [d["confidence"] for d in c["predictions"]] == [float(v) for v in p]
    clf.close()
    with pytest.raises(FileNotFoundError):
        asyncio.run(ModelManager(cache_dir=str(tmp_path), frame_source=lambda p: Src()).classify_places("/videos/x.mp4", {}))
