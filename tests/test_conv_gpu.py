"""GPU parity: K4 implicit-GEMM conv (fp16 operands, fp32 accumulate) vs torch-CPU conv2d on the
same fp16-rounded operands.  Tolerance: the fp32 accumulation order differs (MFMA vs CPU), and the
result is rounded to fp16 once, so outputs agree to 1 fp16 ulp (rel 2**-10) + tiny absolute slack."""
import numpy as np
import pytest

from eioku_amd import ops

pytestmark = pytest.mark.gpu

RTOL = 2.0 ** -9   # 2 fp16 ulps
ATOL = 2e-3


def ref_conv(x_nhwc16, w, b, stride, silu, residual16=None, out_f32=False):
    import torch
    import torch.nn.functional as F

    x = torch.from_numpy(x_nhwc16.astype(np.float32)).permute(0, 3, 1, 2)
    w16 = torch.from_numpy(w.astype(np.float16).astype(np.float32))
    k = w.shape[-1]
    y = F.conv2d(x, w16, None if b is None else torch.from_numpy(b.astype(np.float32)), stride=stride, padding=k // 2)
    if silu:
        y = y * torch.sigmoid(y)
    y = y.permute(0, 2, 3, 1).contiguous()
    if out_f32:
        return y.numpy()
    y = y.half()
    if residual16 is not None:
        y = (y.float() + torch.from_numpy(residual16.astype(np.float32))).half()
    return y.numpy()


def close(a, b):
    a = a.astype(np.float32)
    b = b.astype(np.float32)
    return np.all(np.abs(a - b) <= ATOL + RTOL * np.abs(b))


CASES = [
    # n, h, w, cin, cout, k, stride
    (1, 8, 16, 8, 16, 3, 1),      # exactly one tile, one chunk (cin padded 8->32)
    (2, 20, 20, 32, 32, 3, 1),    # ragged tiles (20 = 16+4, 8+8+4)
    (1, 33, 47, 16, 32, 3, 2),    # stride 2, odd sizes
    (2, 40, 40, 64, 64, 3, 1),
    (1, 24, 40, 48, 80, 3, 1),    # cin not multiple of 32, cout 80 (cls branch)
    (3, 12, 20, 128, 64, 1, 1),   # 1x1
    (1, 12, 20, 384, 128, 1, 1),  # 1x1 over a concat width
    (1, 16, 16, 64, 1, 1, 1),     # face head: cout 1
    (1, 16, 16, 80, 80, 1, 1),
    (2, 64, 64, 8, 16, 3, 2),     # stem (3 channels padded to 8)
    (1, 20, 20, 256, 256, 3, 2),
    (1, 9, 9, 144, 144, 3, 1),    # v8m-style widths
    # flattened deep-K kernel: tiles cross rows and frame borders
    (3, 20, 20, 128, 128, 3, 1),
    (2, 40, 40, 128, 80, 3, 1),
    (3, 40, 40, 128, 256, 3, 2),
    (2, 21, 19, 136, 64, 3, 2),   # odd sizes, cin not a multiple of 32
    (5, 5, 7, 128, 32, 3, 1),     # several frames inside one 128-pixel tile
    (3, 7, 5, 160, 48, 3, 2),
    # deep-K 1x1 on few pixels, cin tail
    (2, 20, 20, 512, 256, 1, 1),
    (1, 10, 10, 256, 128, 1, 1),
    (1, 9, 7, 136, 80, 1, 1),
    (2, 5, 3, 160, 64, 1, 1),
]


@pytest.mark.parametrize("n,h,w,cin,cout,k,stride", CASES)
def test_conv_matches_torch_cpu(gpu, n, h, w, cin, cout, k, stride):
    import torch

    rng = np.random.default_rng(cin * 1000 + cout + k + stride)
    x = rng.standard_normal((n, h, w, cin)).astype(np.float16)
    wgt = (rng.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32)
    b = (0.1 * rng.standard_normal(cout)).astype(np.float32)
    got = ops.conv2d_f16(torch.from_numpy(x).to(gpu), wgt, b, stride=stride, silu=True).cpu().numpy()
    want = ref_conv(x, wgt, b, stride, True)
    assert got.shape == want.shape
    assert close(got, want), float(np.abs(got.astype(np.float32) - want.astype(np.float32)).max())


def test_conv_slices_residual_and_fp32_out(gpu):
    """Concat-slice addressing: read channels [16,48) of a 64-wide buffer, write at offset 8 of a
    48-wide buffer with a residual taken from channels [32,64) of a third buffer."""
    import torch

    rng = np.random.default_rng(11)
    n, h, w = 2, 17, 23
    xbuf = rng.standard_normal((n, h, w, 64)).astype(np.float16)
    rbuf = rng.standard_normal((n, h, w, 64)).astype(np.float16)
    wgt = (rng.standard_normal((32, 32, 3, 3)) / 17.0).astype(np.float32)
    b = (0.1 * rng.standard_normal(32)).astype(np.float32)
    out = torch.full((n, h, w, 48), 7.0, dtype=torch.float16, device=gpu)
    ops.conv2d_f16(torch.from_numpy(xbuf).to(gpu), wgt, b, in_coff=16, cin=32, residual=torch.from_numpy(rbuf).to(gpu),
                   res_coff=32, out=out, out_coff=8)
    got = out.cpu().numpy()
    want = ref_conv(xbuf[..., 16:48], wgt, b, 1, True, residual16=rbuf[..., 32:64])
    assert close(got[..., 8:40], want)
    assert np.all(got[..., :8] == 7.0) and np.all(got[..., 40:] == 7.0)  # neighbours untouched
    # fp32 output, no activation (Detect's last 1x1)
    w1 = (rng.standard_normal((80, 64, 1, 1)) / 8.0).astype(np.float32)
    got32 = ops.conv2d_f16(torch.from_numpy(xbuf).to(gpu), w1, None, silu=False, out_f32=True).cpu().numpy()
    want32 = ref_conv(xbuf, w1, None, 1, False, out_f32=True)
    assert np.allclose(got32, want32, rtol=1e-4, atol=1e-4)


def test_conv_ignores_nan_outside_slice(gpu):
    """Channels outside the input slice (and the zero-padded chunk tail) must never leak in."""
    import torch

    rng = np.random.default_rng(12)
    x = rng.standard_normal((1, 8, 16, 24)).astype(np.float16)
    x[..., 16:] = np.nan
    wgt = (rng.standard_normal((16, 16, 3, 3)) / 12.0).astype(np.float32)
    got = ops.conv2d_f16(torch.from_numpy(x).to(gpu), wgt, None, cin=16).cpu().numpy()
    assert np.isfinite(got).all()
    assert close(got, ref_conv(x[..., :16], wgt, None, 1, True))


def test_conv_rejects_bad_shapes(gpu):
    import torch
    from eioku_amd._lib import EiokuHipError

    x = torch.zeros((1, 8, 8, 12), dtype=torch.float16, device=gpu)
    with pytest.raises(EiokuHipError):
        ops.conv2d_f16(x, np.zeros((16, 12, 3, 3), np.float32), None)  # cin % 8 != 0
    x = torch.zeros((1, 8, 8, 16), dtype=torch.float16, device=gpu)
    with pytest.raises(EiokuHipError):
        ops.conv2d_f16(x, np.zeros((16, 16, 5, 5), np.float32), None)  # k=5
