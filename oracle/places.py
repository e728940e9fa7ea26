"""Places365 oracle: what ``ModelManager.classify_places`` computes per sampled frame
(``/root/reference/ml-service/src/services/model_manager.py:560-713``).

Test infrastructure (see ``oracle/__init__.py``).  The reference delegates the arithmetic to pip dependencies that are
not vendored [PUBLIC-LIB]: Pillow's antialiased bilinear resize (``transforms.Resize((224, 224))`` on a PIL image =
``Image.resize(..., BILINEAR)``), torchvision's ``ToTensor`` / ``Normalize`` and its ResNet18 graph, torch's softmax and
descending sort.  Restated here: the resize in integer numpy following Pillow's ``Resample.c`` (pinned against Pillow
itself, which IS installed in this image: tests/test_oracle_places.py), the network in torch-CPU with BatchNorm folded
into the convolutions (the product receives folded weights).  **Parity otherwise unpinned**: torchvision is not
installed and no Places365 checkpoint is reachable offline; the orchestration (frame sampling, timestamps, label
parsing, result dict) IS pinned by a fixture captured from the reference's own loop
(tests/golden/make_reference_fixtures.py -> ref_places_loop.json).

``fp16=True`` emulates the product's arithmetic (fp16 storage of weights and of every activation tensor, fp32
accumulation), ``fp16=False`` is the reference's fp32.
"""
from __future__ import annotations

import numpy as np

MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)
PRECISION_BITS = 32 - 8 - 2  # Pillow Resample.c: 8-bit pixels, 22-bit fixed-point coefficients

# torchvision.models.resnet18: (name, cout, cin, k, stride) in forward order; "dN" = the block's downsample branch
LAYERS = [("conv1", 64, 3, 7, 2)]
for li, (c, s) in enumerate([(64, 1), (128, 2), (256, 2), (512, 2)], start=1):
    cin = 64 if li == 1 else c // 2
    for b in range(2):
        st = s if b == 0 else 1
        ci = cin if b == 0 else c
        LAYERS.append((f"layer{li}.{b}.conv1", c, ci, 3, st))
        LAYERS.append((f"layer{li}.{b}.conv2", c, c, 3, 1))
        if b == 0 and (st != 1 or ci != c):
            LAYERS.append((f"layer{li}.{b}.downsample.0", c, ci, 1, st))


def resize_coeffs(in_size: int, out_size: int):
    """Pillow ``precompute_coeffs`` + ``normalize_coeffs_8bpc`` for the bilinear (triangle) filter over the whole axis:
    per output position the first input index, the tap count, and int32 coefficients scaled by 2**22."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = np.zeros(ksize, np.float64)
        for x in range(xmax):
            v = abs((x + xmin - center + 0.5) * ss)
            w[x] = 1.0 - v if v < 1.0 else 0.0
        ww = w[:xmax].sum()
        # Pillow accumulates ww in a double in x order; the sum of <= ksize doubles here is that same sequence
        ww = 0.0
        for x in range(xmax):
            ww += w[x]
        if ww != 0.0:
            w[:xmax] /= ww
        for x in range(xmax):
            kk[xx, x] = int(0.5 + w[x] * (1 << PRECISION_BITS)) if w[x] >= 0 else int(-0.5 + w[x] * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _pass(img: np.ndarray, bounds, kk, axis: int) -> np.ndarray:
    """one 8-bit resample pass along ``axis`` (0 = rows / vertical, 1 = columns / horizontal) of an (h, w, c) image"""
    out_size = len(bounds)
    shape = list(img.shape)
    shape[axis] = out_size
    out = np.empty(shape, np.uint8)
    src = img.astype(np.int64)
    for xx in range(out_size):
        lo, n = int(bounds[xx, 0]), int(bounds[xx, 1])
        k = kk[xx, :n].astype(np.int64)
        if axis == 1:
            acc = (src[:, lo:lo + n, :] * k[None, :, None]).sum(1)
        else:
            acc = (src[lo:lo + n, :, :] * k[:, None, None]).sum(0)
        acc = (acc + (1 << (PRECISION_BITS - 1))) >> PRECISION_BITS
        if axis == 1:
            out[:, xx, :] = np.clip(acc, 0, 255)
        else:
            out[xx, :, :] = np.clip(acc, 0, 255)
    return out


def pil_resize_bilinear(rgb: np.ndarray, size: int = 224) -> np.ndarray:
    """``Image.fromarray(rgb).resize((size, size), Image.BILINEAR)``: horizontal pass (uint8 result), then vertical."""
    h, w, _ = rgb.shape
    out = rgb
    if w != size:
        out = _pass(out, *resize_coeffs(w, size), axis=1)
    if h != size:
        out = _pass(out, *resize_coeffs(h, size), axis=0)
    return out


def preprocess(frames_bgr: np.ndarray, size: int = 224):
    """BGR uint8 (n,h,w,3) -> float32 NCHW: cv2.cvtColor(BGR2RGB), Resize((224,224)) on the PIL image, ToTensor,
    Normalize(mean, std) - ``model_manager.py:630-640,666-667``."""
    import torch

    out = []
    for f in frames_bgr:
        r = pil_resize_bilinear(np.ascontiguousarray(f[..., ::-1]), size)
        t = torch.from_numpy(r).permute(2, 0, 1).to(torch.float32).div(255)  # ToTensor
        m = torch.tensor(MEAN, dtype=torch.float32)[:, None, None]
        s = torch.tensor(STD, dtype=torch.float32)[:, None, None]
        out.append((t - m) / s)  # Normalize: tensor.sub_(mean).div_(std)
    return torch.stack(out)


def random_state(seed: int = 3) -> dict:
    """Folded (weight OIHW, bias) per convolution + ("fc": (365, 512), (365,)).  He-style scales keep activations O(1)
    through the residual stack so that the softmax is neither flat nor one-hot."""
    rng = np.random.default_rng(seed)
    st = {}
    for name, cout, cin, k, _ in LAYERS:
        fan = cin * k * k
        gain = 0.7 if name.endswith("conv2") else 1.4  # conv2 feeds the residual sum
        st[name] = ((gain * rng.standard_normal((cout, cin, k, k)) / np.sqrt(fan)).astype(np.float32),
                    (0.05 * rng.standard_normal(cout)).astype(np.float32))
    st["fc"] = ((4.0 * rng.standard_normal((365, 512)) / np.sqrt(512)).astype(np.float32),
                (0.1 * rng.standard_normal(365)).astype(np.float32))
    return st


def fold_bn(w, gamma, beta, mean, var, eps: float = 1e-5):
    """conv (no bias) followed by BatchNorm2d in eval mode -> (weight, bias) of one convolution (float64 fold)."""
    s = gamma.astype(np.float64) / np.sqrt(var.astype(np.float64) + eps)
    return (w.astype(np.float64) * s[:, None, None, None]).astype(np.float32), (beta.astype(np.float64) - mean.astype(np.float64) * s).astype(np.float32)


class ResNet18:
    def __init__(self, state: dict, fp16: bool = True):
        import torch

        self.fp16 = fp16
        rw = (lambda t: t.half().float()) if fp16 else (lambda t: t)
        self.p = {k: (rw(torch.from_numpy(np.asarray(w, np.float32))), torch.from_numpy(np.asarray(b, np.float32)))
                  for k, (w, b) in state.items()}

    def _h(self, t):
        return t.half().float() if self.fp16 else t

    def conv(self, name, x, stride, relu):
        import torch.nn.functional as F

        w, b = self.p[name]
        y = F.conv2d(x, w, b, stride=stride, padding=w.shape[-1] // 2)
        if relu:
            y = F.relu(y)
        return self._h(y)

    def logits(self, x):
        """x: float32 NCHW (preprocess output) -> float32 (n, 365)"""
        import torch
        import torch.nn.functional as F

        with torch.no_grad():
            x = self._h(x)
            x = self.conv("conv1", x, 2, True)
            x = F.max_pool2d(x, 3, 2, 1)
            for li, s in ((1, 1), (2, 2), (3, 2), (4, 2)):
                for b in range(2):
                    st = s if b == 0 else 1
                    idn = x
                    y = self.conv(f"layer{li}.{b}.conv1", x, st, True)
                    y = self.conv(f"layer{li}.{b}.conv2", y, 1, False)
                    if f"layer{li}.{b}.downsample.0" in self.p:
                        idn = self.conv(f"layer{li}.{b}.downsample.0", x, st, False)
                    x = self._h(F.relu(self._h(y + idn)))
            pooled = x.mean((2, 3))  # AdaptiveAvgPool2d(1): fp32 mean of the 7 x 7 map
            w, b = self.p["fc"]
            return (pooled @ w.t() + b).numpy()


def top_predictions(logits: np.ndarray, top_k: int):
    """softmax(logit, 1) -> sort descending -> [(class index, probability)] * top_k per frame
    (``model_manager.py:672-687``; ties: torch.sort is not stable there, the tests use separated logits)."""
    import torch

    out = []
    for row in torch.from_numpy(np.asarray(logits, np.float32)):
        h = torch.nn.functional.softmax(row[None], 1).squeeze()
        probs, idx = h.sort(0, True)
        out.append([(int(idx[j]), float(probs[j])) for j in range(min(top_k, len(idx)))])
    return out
