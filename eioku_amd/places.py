"""Place classification (Places365 ResNet18) on the HIP kernels of ``csrc/resnet.hip``.

Drop-in arithmetic for ``ModelManager.classify_places``
(``/root/reference/ml-service/src/services/model_manager.py:560-713``): per sampled frame the reference converts
BGR -> RGB, resizes the PIL image to 224 x 224 (``transforms.Resize``: Pillow's antialiased bilinear resample), applies
``ToTensor`` / ``Normalize``, runs torchvision's ``resnet18`` with a 365-way ``fc``, takes ``softmax``, sorts descending
and keeps ``top_k``.  Here the host computes Pillow's coefficient tables (float64, as ``precompute_coeffs`` does) and
everything else runs on the device behind ``eioku_resnet18_classify``.  No CPU fallback.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import numpy as np

from . import _lib
from ._buffers import current_stream, on_device, ptr

SIZE = 224
NUM_CLASSES = 365
PRECISION_BITS = 22  # Pillow Resample.c: 32 - 8 - 2 for 8-bit pixels
_BN_EPS = 1e-5


def resize_tables(in_size: int, out_size: int = SIZE):
    """Pillow ``precompute_coeffs`` + ``normalize_coeffs_8bpc`` for the bilinear filter, whole axis ``in_size`` ->
    ``out_size``: ``(bounds int32 (out,2) = first input index | taps, k int32 (out,ksize) taps x 2**22, ksize)``."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    xx = np.arange(out_size, dtype=np.float64)
    center = (xx + 0.5) * scale
    xmin = (center - support + 0.5).astype(np.int64)       # C (int) cast: truncation toward zero; values are >= -0.5
    xmin = np.maximum(xmin, 0)
    xmax = np.minimum((center + support + 0.5).astype(np.int64), in_size) - xmin
    x = np.arange(ksize, dtype=np.float64)[None, :]
    arg = np.abs((x + xmin[:, None] - center[:, None] + 0.5) * (1.0 / filterscale))
    w = np.where(arg < 1.0, 1.0 - arg, 0.0)
    w[x >= xmax[:, None]] = 0.0
    ww = np.zeros(out_size, np.float64)
    for t in range(ksize):  # Pillow sums the taps in index order
        ww = ww + w[:, t]
    w = np.where(ww[:, None] != 0.0, w / np.where(ww == 0.0, 1.0, ww)[:, None], w)
    k = (0.5 + w * (1 << PRECISION_BITS)).astype(np.int64)  # taps are >= 0 for the triangle filter: (int)(0.5 + v)
    k[x.repeat(out_size, 0) >= xmax[:, None]] = 0
    bounds = np.stack([xmin, xmax], 1).astype(np.int32)
    return np.ascontiguousarray(bounds), np.ascontiguousarray(k.astype(np.int32)), ksize


def load_labels(cache_dir) -> list[str]:
    """The reference's label lookup (``model_manager.py:579-606``): ``<cache>/places365/categories_places365.txt`` first,
    then the list the reference ships beside ``ml-service/src`` (found here through ``EIOKU_PLACES365_LABELS``, next to
    or above this package when it is vendored into ml-service, or in the working directory), else generic
    ``place_<i>``.  Lines look like ``/a/airfield 0`` -> ``airfield``."""
    import os

    pkg = Path(__file__).resolve().parent
    candidates = [Path(cache_dir) / "places365" / "categories_places365.txt"]
    if os.environ.get("EIOKU_PLACES365_LABELS"):
        candidates.append(Path(os.environ["EIOKU_PLACES365_LABELS"]))
    candidates += [pkg.parent / "categories_places365.txt", pkg.parent.parent / "categories_places365.txt",
                   Path.cwd() / "categories_places365.txt", Path.cwd() / "ml-service" / "categories_places365.txt"]
    for path in candidates:
        if path.exists():
            with open(path) as f:
                return [line.strip().split(" ")[0][3:] for line in f.readlines()]
    return [f"place_{i}" for i in range(NUM_CLASSES)]


def fold_state(state_dict: dict) -> dict:
    """torchvision ``resnet18`` state dict (optionally under ``"state_dict"`` and with ``module.`` prefixes, as the
    Places365 release ships it: ``model_manager.py:612-621``) -> ``{conv name: (weight, bias)}`` with each BatchNorm
    folded into its convolution, plus ``"fc": (weight, bias)``."""
    sd = state_dict.get("state_dict", state_dict)
    sd = {k.replace("module.", ""): np.asarray(v.detach().cpu().numpy() if hasattr(v, "detach") else v) for k, v in sd.items()}
    out = {}

    def fold(conv, bn):
        w = sd[conv + ".weight"].astype(np.float64)
        s = sd[bn + ".weight"].astype(np.float64) / np.sqrt(sd[bn + ".running_var"].astype(np.float64) + _BN_EPS)
        b = sd[bn + ".bias"].astype(np.float64) - sd[bn + ".running_mean"].astype(np.float64) * s
        return (w * s[:, None, None, None]).astype(np.float32), b.astype(np.float32)

    out["conv1"] = fold("conv1", "bn1")
    for li in range(1, 5):
        for b in range(2):
            p = f"layer{li}.{b}"
            out[p + ".conv1"] = fold(p + ".conv1", p + ".bn1")
            out[p + ".conv2"] = fold(p + ".conv2", p + ".bn2")
            if p + ".downsample.0.weight" in sd:
                out[p + ".downsample.0"] = fold(p + ".downsample.0", p + ".downsample.1")
    out["fc"] = (sd["fc.weight"].astype(np.float32), sd["fc.bias"].astype(np.float32))
    return out


def load_checkpoint(path) -> dict:
    """``resnet18_places365.pth.tar`` -> folded state (torch's restricted unpickler: tensors only)."""
    import torch

    return fold_state(torch.load(str(path), map_location="cpu", weights_only=True))


def random_state(seed: int = 3) -> dict:
    """Seeded random folded weights of the exact architecture (benchmarks, tests: no checkpoint is reachable offline)."""
    rng = np.random.default_rng(seed)
    st = {}
    lib = _lib.load()
    _lib.init()
    h = C.c_void_p()
    _lib.check(lib.eioku_resnet18_create(NUM_CLASSES, C.byref(h)), "eioku_resnet18_create")
    try:
        for i in range(lib.eioku_resnet18_num_convs(h)):
            name = C.create_string_buffer(64)
            co, ci, k, s = C.c_int(), C.c_int(), C.c_int(), C.c_int()
            _lib.check(lib.eioku_resnet18_conv_info(h, i, name, 64, C.byref(co), C.byref(ci), C.byref(k), C.byref(s)), "conv_info")
            n = name.value.decode()
            gain = 0.7 if n.endswith("conv2") else 1.4
            st[n] = ((gain * rng.standard_normal((co.value, ci.value, k.value, k.value)) / np.sqrt(ci.value * k.value * k.value)).astype(np.float32),
                     (0.05 * rng.standard_normal(co.value)).astype(np.float32))
    finally:
        lib.eioku_resnet18_destroy(h)
    st["fc"] = ((4.0 * rng.standard_normal((NUM_CLASSES, 512)) / np.sqrt(512)).astype(np.float32),
                (0.1 * rng.standard_normal(NUM_CLASSES)).astype(np.float32))
    return st


class Places365Classifier:
    """``model = resnet18(); model.fc = Linear(512, 365); model.load_state_dict(...)`` + the per-frame transform and
    softmax / sort of ``classify_places`` (``model_manager.py:609-640,666-687``)."""

    def __init__(self, state: dict, labels: list[str] | None = None):
        self._lib = _lib.load()
        _lib.init()
        self.labels = list(labels) if labels is not None else [f"place_{i}" for i in range(NUM_CLASSES)]
        h = C.c_void_p()
        _lib.check(self._lib.eioku_resnet18_create(NUM_CLASSES, C.byref(h)), "eioku_resnet18_create")
        self._h = h
        self._tables = {}
        for i in range(self._lib.eioku_resnet18_num_convs(h)):
            name = C.create_string_buffer(64)
            co, ci, k, s = C.c_int(), C.c_int(), C.c_int(), C.c_int()
            _lib.check(self._lib.eioku_resnet18_conv_info(h, i, name, 64, C.byref(co), C.byref(ci), C.byref(k), C.byref(s)),
                       "eioku_resnet18_conv_info")
            n = name.value.decode()
            if n not in state:
                raise KeyError(f"state has no weights for {n}")
            w, b = (np.ascontiguousarray(a, dtype=np.float32) for a in state[n])
            if w.shape != (co.value, ci.value, k.value, k.value) or b.shape != (co.value,):
                raise ValueError(f"{n}: expected weight {(co.value, ci.value, k.value, k.value)}, got {w.shape}")
            _lib.check(self._lib.eioku_resnet18_set_conv(h, i, ptr(w), ptr(b)), f"eioku_resnet18_set_conv({n})")
        w, b = (np.ascontiguousarray(a, dtype=np.float32) for a in state["fc"])
        if w.shape != (NUM_CLASSES, 512):
            raise ValueError(f"fc: expected weight {(NUM_CLASSES, 512)}, got {w.shape}")
        _lib.check(self._lib.eioku_resnet18_set_fc(h, ptr(w), ptr(b)), "eioku_resnet18_set_fc")

    @classmethod
    def from_cache(cls, cache_dir, seed: int | None = None):
        """Weights from ``<cache>/places365/resnet18_places365.pth.tar`` (``model_manager.py:613``).  The reference falls
        back to torchvision's unseeded random initialisation when the file is missing; that cannot be reproduced, so a
        missing file is an error here unless a ``seed`` asks for this package's seeded random weights."""
        path = Path(cache_dir) / "places365" / "resnet18_places365.pth.tar"
        if path.exists():
            return cls(load_checkpoint(path), load_labels(cache_dir))
        if seed is None:
            raise FileNotFoundError(f"{path} not found (no weights to classify places with)")
        return cls(random_state(seed), load_labels(cache_dir))

    def _tab(self, h: int, w: int):
        key = (h, w)
        if key not in self._tables:
            self._tables[key] = (*resize_tables(w), *resize_tables(h))
        return self._tables[key]

    def preprocess(self, frames_bgr):
        """-> ``(fp16 (n,224,224,4) network input, uint8 (n,224,224,3) resized RGB)`` CUDA tensors (parity helper)."""
        import torch

        n, h, w, _ = (int(s) for s in frames_bgr.shape)
        xb, xk, kx, yb, yk, ky = self._tab(h, w)
        dev = frames_bgr.device if on_device(frames_bgr) else torch.device("cuda", torch.cuda.current_device())
        x = torch.empty((n, SIZE, SIZE, 4), dtype=torch.float16, device=dev)
        u8 = torch.empty((n, SIZE, SIZE, 3), dtype=torch.uint8, device=dev)
        _lib.check(self._lib.eioku_places_preprocess(self._h, ptr(frames_bgr), n, h, w, ptr(xb), ptr(xk), kx, ptr(yb), ptr(yk), ky,
                                                     ptr(x), ptr(u8), _lib.MEM_DEVICE if on_device(frames_bgr) else _lib.MEM_HOST,
                                                     current_stream(x)), "eioku_places_preprocess")
        return x, u8

    def forward_raw(self, x):
        """fp16 (n,224,224,4) CUDA tensor -> float32 (n,365) logits (CUDA)."""
        import torch

        n = int(x.shape[0])
        out = torch.empty((n, NUM_CLASSES), dtype=torch.float32, device=x.device)
        _lib.check(self._lib.eioku_resnet18_forward(self._h, ptr(x.contiguous()), n, ptr(out), current_stream(x)), "eioku_resnet18_forward")
        return out

    def classify(self, frames_bgr, top_k: int = 5, with_logits: bool = False):
        """BGR uint8 ``(n,h,w,3)`` (numpy: staged; CUDA tensor: zero copy) -> ``(probs float32 (n,top_k), classes int32
        (n,top_k))`` numpy arrays in descending probability order (+ logits (n,365) when asked)."""
        n, h, w, c = (int(s) for s in frames_bgr.shape)
        if c != 3:
            raise ValueError("expected (n,h,w,3) BGR frames")
        top_k = max(1, min(int(top_k), NUM_CLASSES))
        xb, xk, kx, yb, yk, ky = self._tab(h, w)
        dev = on_device(frames_bgr)
        if dev:
            import torch

            probs = torch.empty((n, top_k), dtype=torch.float32, device=frames_bgr.device)
            cls = torch.empty((n, top_k), dtype=torch.int32, device=frames_bgr.device)
            logits = torch.empty((n, NUM_CLASSES), dtype=torch.float32, device=frames_bgr.device) if with_logits else None
        else:
            frames_bgr = np.ascontiguousarray(frames_bgr, dtype=np.uint8)
            probs = np.empty((n, top_k), np.float32)
            cls = np.empty((n, top_k), np.int32)
            logits = np.empty((n, NUM_CLASSES), np.float32) if with_logits else None
        _lib.check(self._lib.eioku_resnet18_classify(self._h, ptr(frames_bgr), n, h, w, ptr(xb), ptr(xk), kx, ptr(yb), ptr(yk), ky,
                                                     top_k, ptr(probs), ptr(cls), ptr(logits),
                                                     _lib.MEM_DEVICE if dev else _lib.MEM_HOST, current_stream(frames_bgr)),
                   "eioku_resnet18_classify")
        if dev:
            probs, cls = probs.cpu().numpy(), cls.cpu().numpy()
            logits = logits.cpu().numpy() if with_logits else None
        return (probs, cls, logits) if with_logits else (probs, cls)

    def last_flops(self) -> float:
        f = C.c_double(0)
        _lib.check(self._lib.eioku_resnet18_last_flops(self._h, C.byref(f)), "eioku_resnet18_last_flops")
        return f.value

    def close(self):
        if getattr(self, "_h", None):
            self._lib.eioku_resnet18_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
