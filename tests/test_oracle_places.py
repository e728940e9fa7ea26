"""CPU: the Places365 oracle's pieces against what is installed here - Pillow itself for the antialiased resize - and
its own invariants (BatchNorm fold, fp16 twin)."""
import numpy as np
import pytest

from oracle import places as op


@pytest.mark.parametrize("h,w", [(1080, 1920), (480, 854), (224, 224), (300, 200), (100, 640), (7, 9)])
def test_resize_restatement_equals_pillow(h, w):
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(h * 7 + w)
    rgb = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    rgb[: h // 3] = np.repeat(np.repeat(rng.integers(0, 256, (h // 3 // 8 + 1, w // 8 + 1, 3), dtype=np.uint8), 8, 0), 8, 1)[: h // 3, :w]
    want = np.asarray(Image.fromarray(rgb).resize((224, 224), Image.BILINEAR))
    got = op.pil_resize_bilinear(rgb, 224)
    assert got.dtype == np.uint8 and np.array_equal(got, want)


def test_resize_coefficients_are_normalised_and_cover_the_axis():
    for n_in in (1920, 1080, 224, 100):
        b, k = op.resize_coeffs(n_in, 224)
        assert b[0, 0] == 0 and b[-1, 0] + b[-1, 1] == n_in
        s = k.sum(1)
        assert np.all(np.abs(s - (1 << op.PRECISION_BITS)) <= k.shape[1])  # rounded taps sum to 1.0 +- a few ulps


def test_preprocess_is_totensor_normalize():
    rng = np.random.default_rng(3)
    f = rng.integers(0, 256, (2, 224, 224, 3), dtype=np.uint8)  # no resize at 224
    x = op.preprocess(f).numpy()
    want = (f[..., ::-1].transpose(0, 3, 1, 2).astype(np.float32) / np.float32(255) - np.array(op.MEAN, np.float32)[None, :, None, None]) \
        / np.array(op.STD, np.float32)[None, :, None, None]
    assert np.array_equal(x, want.astype(np.float32))


def test_bn_fold_equals_conv_then_batchnorm():
    import torch

    rng = np.random.default_rng(5)
    w = rng.standard_normal((8, 4, 3, 3)).astype(np.float32)
    g, b, m = (rng.standard_normal(8).astype(np.float32) for _ in range(3))
    v = rng.uniform(0.5, 2.0, 8).astype(np.float32)
    x = torch.from_numpy(rng.standard_normal((1, 4, 9, 9)).astype(np.float32))
    bn = torch.nn.BatchNorm2d(8).eval()
    with torch.no_grad():
        bn.weight.copy_(torch.from_numpy(g)); bn.bias.copy_(torch.from_numpy(b))
        bn.running_mean.copy_(torch.from_numpy(m)); bn.running_var.copy_(torch.from_numpy(v))
        want = bn(torch.nn.functional.conv2d(x, torch.from_numpy(w), padding=1))
        fw, fb = op.fold_bn(w, g, b, m, v)
        got = torch.nn.functional.conv2d(x, torch.from_numpy(fw), torch.from_numpy(fb), padding=1)
    assert torch.allclose(got, want, atol=2e-5)


def test_network_shapes_fp16_twin_and_topk():
    st = op.random_state(3)
    assert [n for n, *_ in op.LAYERS if "downsample" in n] == [f"layer{i}.0.downsample.0" for i in (2, 3, 4)] and len(op.LAYERS) == 20
    rng = np.random.default_rng(1)
    frames = rng.integers(0, 256, (2, 96, 128, 3), dtype=np.uint8)
    x = op.preprocess(frames)
    l32 = op.ResNet18(st, fp16=False).logits(x)
    l16 = op.ResNet18(st, fp16=True).logits(x)
    assert l32.shape == (2, 365) and np.isfinite(l32).all()
    rms = np.sqrt((l32.astype(np.float64) ** 2).mean())
    assert rms > 0.3 and np.abs(l16 - l32).max() < 0.05 * rms  # the fp16 build's drift from the reference's fp32
    top = op.top_predictions(l32, 5)
    assert len(top) == 2 and len(top[0]) == 5 and top[0][0][1] >= top[0][1][1] >= top[0][4][1] > 0
    p = np.exp(l32[0] - l32[0].max()); p /= p.sum()
    assert top[0][0][0] == int(p.argmax()) and abs(top[0][0][1] - p.max()) < 1e-6
