"""YOLOv8 weights for the HIP detector: variants, random-init states, and loaders.

PyTorch is used here only to read checkpoints (``torch.load`` / safetensors); the arrays handed to
``libeioku_hip`` are plain fp32 numpy (``[cout][cin][k][k]`` + bias), BatchNorm already folded.
Reference call sites: ``YOLO(model_path); model.to(device)``
(``/root/reference/ml-service/src/services/model_manager.py:252-254, 346-348``).
"""

from __future__ import annotations

import io
import pickle
import re
import zipfile
from pathlib import Path

import numpy as np

# backbone widths c1..c5 and C2f repeats (layers 2,4,6,8; the neck uses the first) per scale:
# yolov8.yaml `scales` [depth, width, max_channels] -> make_divisible(min(c, max_ch) * width, 8)
YOLO_VARIANTS = {
    "n": ((16, 32, 64, 128, 256), (1, 2, 2, 1)),
    "s": ((32, 64, 128, 256, 512), (1, 2, 2, 1)),
    "m": ((48, 96, 192, 384, 576), (2, 4, 4, 2)),
    "l": ((64, 128, 256, 512, 512), (3, 6, 6, 3)),
    "x": ((80, 160, 320, 640, 640), (3, 6, 6, 3)),
}

COCO_NAMES = [
    "person", "bicycle", "car", "motorcycle", "airplane", "bus", "train", "truck", "boat", "traffic light",
    "fire hydrant", "stop sign", "parking meter", "bench", "bird", "cat", "dog", "horse", "sheep", "cow",
    "elephant", "bear", "zebra", "giraffe", "backpack", "umbrella", "handbag", "tie", "suitcase", "frisbee",
    "skis", "snowboard", "sports ball", "kite", "baseball bat", "baseball glove", "skateboard", "surfboard",
    "tennis racket", "bottle", "wine glass", "cup", "fork", "knife", "spoon", "bowl", "banana", "apple",
    "sandwich", "orange", "broccoli", "carrot", "hot dog", "pizza", "donut", "cake", "chair", "couch",
    "potted plant", "bed", "dining table", "toilet", "tv", "laptop", "mouse", "remote", "keyboard", "cell phone",
    "microwave", "oven", "toaster", "sink", "refrigerator", "book", "clock", "vase", "scissors", "teddy bear",
    "hair drier", "toothbrush",
]


def variant_from_model_name(model_name: str) -> tuple[str, int, dict[int, str]]:
    """``"yolov8s.pt"`` -> ("s", 80, names); ``"yolov8n-face.pt"`` -> ("n", 1, {0: "face"})."""
    m = re.match(r"yolov8([nsmlx])(-face)?(\.\w+)?$", Path(model_name).name)
    if not m:
        raise ValueError(f"unsupported model name {model_name!r} (expected yolov8{{n,s,m,l,x}}[-face].pt)")
    if m.group(2):
        return m.group(1), 1, {0: "face"}
    return m.group(1), 80, dict(enumerate(COCO_NAMES))


def conv_table(variant: str, nc: int) -> list[tuple[str, int, int, int, int]]:
    """(name, cout, cin, k, stride) of every convolution in module order (mirrors csrc/yolo.hip)."""
    ch, depth = YOLO_VARIANTS[variant]
    c1, c2, c3, c4, c5 = ch
    d0, d1, d2, d3 = depth
    t: list[tuple[str, int, int, int, int]] = []

    def c2f(p, cin, cout, n):
        c = cout // 2
        t.append((f"{p}.cv1.conv", 2 * c, cin, 1, 1))
        for i in range(n):
            t.append((f"{p}.m.{i}.cv1.conv", c, c, 3, 1))
            t.append((f"{p}.m.{i}.cv2.conv", c, c, 3, 1))
        t.append((f"{p}.cv2.conv", cout, (2 + n) * c, 1, 1))

    t.append(("model.0.conv", c1, 3, 3, 2))
    t.append(("model.1.conv", c2, c1, 3, 2))
    c2f("model.2", c2, c2, d0)
    t.append(("model.3.conv", c3, c2, 3, 2))
    c2f("model.4", c3, c3, d1)
    t.append(("model.5.conv", c4, c3, 3, 2))
    c2f("model.6", c4, c4, d2)
    t.append(("model.7.conv", c5, c4, 3, 2))
    c2f("model.8", c5, c5, d3)
    t.append(("model.9.cv1.conv", c5 // 2, c5, 1, 1))
    t.append(("model.9.cv2.conv", c5, 2 * c5, 1, 1))
    c2f("model.12", c5 + c4, c4, d0)
    c2f("model.15", c4 + c3, c3, d0)
    t.append(("model.16.conv", c3, c3, 3, 2))
    c2f("model.18", c3 + c4, c4, d0)
    t.append(("model.19.conv", c4, c4, 3, 2))
    c2f("model.21", c4 + c5, c5, d0)
    cb = max(16, c3 // 4, 64)
    cc = max(c3, min(nc, 100))
    for l, cx in enumerate((c3, c4, c5)):
        t.append((f"model.22.cv2.{l}.0.conv", cb, cx, 3, 1))
        t.append((f"model.22.cv2.{l}.1.conv", cb, cb, 3, 1))
        t.append((f"model.22.cv2.{l}.2", 64, cb, 1, 1))
        t.append((f"model.22.cv3.{l}.0.conv", cc, cx, 3, 1))
        t.append((f"model.22.cv3.{l}.1.conv", cc, cc, 3, 1))
        t.append((f"model.22.cv3.{l}.2", nc, cc, 1, 1))
    return t


# variance-preserving init gains, found by bisection on the head-logit spread of each graph
_INIT_GAIN = {"n": 3.2, "s": 3.27, "m": 2.75, "l": 2.55, "x": 2.55}


def random_state(variant: str, nc: int, seed: int, gain: float | None = None, head_gain: float = 12.0,
                 cls_bias: float = -6.0) -> dict[str, tuple[np.ndarray, np.ndarray]]:
    """Random-init fused weights ``{name: (w[cout,cin,k,k], b[cout])}`` (fp32) of the exact shapes.

    ``std = sqrt(gain / fan_in)`` keeps activations O(1) through the SiLU stack (per-variant gain,
    ``_INIT_GAIN``); the three Detect output convs get ``head_gain`` so logits have O(1) spread, and
    the class-logit bias is pushed negative so that, as with a trained model, only a few percent of
    the anchors pass ``conf``.
    """
    if gain is None:
        gain = _INIT_GAIN[variant]
    rng = np.random.default_rng(seed)
    out = {}
    for name, cout, cin, k, _ in conv_table(variant, nc):
        fan_in = cin * k * k
        head =re.match(r"model\.22\.cv[23]\.\d\.2$", name) is not None
        w = (rng.standard_normal((cout, cin, k, k)) * np.sqrt((head_gain if head else gain) / fan_in)).astype(np.float32)
        b = (0.05 * rng.standard_normal(cout)).astype(np.float32)
        if re.match(r"model\.22\.cv3\.\d\.2$", name):
            b = (b + cls_bias).astype(np.float32)
        out[name] = (w, b)
    return out


def fold_batchnorm(w, gamma, beta, mean, var, eps=1e-3):
    """Conv2d(no bias) + BatchNorm2d -> conv weight + bias (what ``model.fuse()`` does; Ultralytics eps 1e-3)."""
    scale = gamma / np.sqrt(var + eps)
    return (w * scale[:, None, None, None]).astype(np.float32), (beta - mean * scale).astype(np.float32)


def state_from_tensors(tensors: dict[str, np.ndarray], variant: str, nc: int) -> dict[str, tuple[np.ndarray, np.ndarray]]:
    """Ultralytics ``state_dict`` (fused or not) -> fused ``{prefix: (w, b)}`` for :func:`conv_table`."""
    out = {}
    for name, cout, cin, k, _ in conv_table(variant, nc):
        wkey = f"{name}.weight"
        if wkey not in tensors:
            raise KeyError(f"checkpoint has no {wkey}")
        w = np.asarray(tensors[wkey], dtype=np.float32)
        if w.shape != (cout, cin, k, k):
            raise ValueError(f"{wkey}: shape {w.shape} != {(cout, cin, k, k)}")
        bn = name[:-len(".conv")] + ".bn" if name.endswith(".conv") else None
        if bn and f"{bn}.weight" in tensors:
            w, b = fold_batchnorm(w, *(np.asarray(tensors[f"{bn}.{s}"], dtype=np.float32)
                                       for s in ("weight", "bias", "running_mean", "running_var")))
        elif f"{name}.bias" in tensors:
            b = np.asarray(tensors[f"{name}.bias"], dtype=np.float32)
        else:
            b = np.zeros(cout, dtype=np.float32)
        out[name] = (w, b)
    return out


class _Stub:
    """Placeholder for classes of packages that are not installed (``ultralytics.*``)."""

    def __init__(self, *a, **k):
        pass

    def __setstate__(self, state):
        self.__dict__.update(state if isinstance(state, dict) else {"_state": state})


# globals a checkpoint may legitimately reference: tensor reconstruction, plain containers and numpy scalars/arrays.
# Everything else (``ultralytics.*`` model classes, ``torch.nn`` modules, anything importable) becomes an inert
# ``_Stub`` that only stores its state - a checkpoint from MODEL_CACHE_DIR cannot run code through ``__reduce__``.
_ALLOWED_GLOBALS = {
    ("collections", "OrderedDict"), ("builtins", "set"), ("builtins", "frozenset"), ("builtins", "dict"),
    ("builtins", "list"), ("builtins", "tuple"), ("builtins", "int"), ("builtins", "float"), ("builtins", "bool"),
    ("builtins", "str"), ("builtins", "bytes"), ("builtins", "complex"), ("builtins", "slice"), ("builtins", "range"),
    ("torch._utils", "_rebuild_tensor_v2"), ("torch._utils", "_rebuild_tensor"), ("torch._utils", "_rebuild_parameter"),
    ("torch._utils", "_rebuild_parameter_with_state"), ("torch", "Size"), ("torch", "device"), ("torch", "dtype"),
    ("torch.serialization", "_get_layout"), ("torch._tensor", "_rebuild_from_type_v2"), ("torch", "Tensor"),
    ("torch.nn.parameter", "Parameter"),
    ("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
    ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"), ("numpy", "ndarray"), ("numpy", "dtype"),
}
_ALLOWED_PREFIXES = (("torch", "Storage"),)  # torch.FloatStorage, HalfStorage, ... (typed storage tags)


class _TolerantUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        ok = (module, name) in _ALLOWED_GLOBALS or any(module == m and name.endswith(sfx) for m, sfx in _ALLOWED_PREFIXES)
        if not ok and module == "torch":  # dtypes (torch.float16 ...) and legacy tensor types pickle as torch globals
            import torch

            obj = getattr(torch, name, None)
            ok = isinstance(obj, torch.dtype) or (isinstance(obj, type) and issubclass(obj, torch.Tensor))
        if ok:
            try:
                return super().find_class(module, name)
            except (ImportError, AttributeError):
                pass
        return type(name, (_Stub,), {"__module__": module})


class _TolerantPickle:
    """``pickle_module`` for ``torch.load`` that stubs out missing classes instead of failing."""

    __name__ = "tolerant_pickle"
    Unpickler = _TolerantUnpickler
    load = staticmethod(lambda f, **kw: _TolerantUnpickler(f, **kw).load())


def _walk_modules(obj, prefix, out, seen):
    if id(obj) in seen:
        return
    seen.add(id(obj))
    d = getattr(obj, "__dict__", None)
    if not isinstance(d, dict):
        return
    for group in ("_parameters", "_buffers"):
        for k, v in (d.get(group) or {}).items():
            if v is not None and hasattr(v, "detach"):
                out[f"{prefix}{k}"] = v.detach().float().cpu().numpy()
    for k, m in (d.get("_modules") or {}).items():
        _walk_modules(m, f"{prefix}{k}.", out, seen)


def _names_of(obj) -> dict[int, str] | None:
    """``model.names`` of an Ultralytics checkpoint (what the reference reads as ``result.names``,
    ``model_manager.py:281``): a dict or list on the model object."""
    seen = set()
    stack = [obj]
    while stack:
        o = stack.pop()
        if id(o) in seen or o is None:
            continue
        seen.add(id(o))
        d = o if isinstance(o, dict) else getattr(o, "__dict__", None)
        if not isinstance(d, dict):
            continue
        names = d.get("names")
        if isinstance(names, (list, tuple)) and names and all(isinstance(v, str) for v in names):
            return dict(enumerate(names))
        if isinstance(names, dict) and names and all(isinstance(v, str) for v in names.values()):
            return {int(k): v for k, v in names.items()}
        for key in ("ema", "model"):
            if isinstance(d.get(key), (dict, _Stub)):
                stack.append(d[key])
    return None


def load_checkpoint(path: str | Path, variant: str, nc: int):
    """``(state, names)``: :func:`load_state` plus the checkpoint's own class names (``None`` when it has none,
    e.g. a bare state_dict / .npz / .safetensors): a custom-trained ``yolov8*.pt`` labels its detections with
    ITS names, as the reference's ``result.names[class_id]`` does."""
    holder: dict = {}
    state = load_state(path, variant, nc, _names_out=holder)
    return state, holder.get("names")


def load_state(path: str | Path, variant: str, nc: int, _names_out: dict | None = None) -> dict[str, tuple[np.ndarray, np.ndarray]]:
    """Read ``.pt`` (Ultralytics checkpoint or plain state_dict), ``.safetensors`` or ``.npz``."""
    path = Path(path)
    if not path.exists():
        raise FileNotFoundError(f"model weights not found: {path}")
    if path.suffix == ".npz":
        tensors = dict(np.load(path))
    elif path.suffix == ".safetensors":
        from safetensors.numpy import load_file

        tensors = load_file(str(path))
    else:
        import torch

        ckpt = torch.load(str(path), map_location="cpu", weights_only=False, pickle_module=_TolerantPickle)
        tensors = {}
        if isinstance(ckpt, dict) and all(hasattr(v, "detach") for v in ckpt.values()):
            tensors = {k: v.detach().float().numpy() for k, v in ckpt.items()}
        else:
            root = ckpt
            if isinstance(ckpt, dict):
                root = ckpt.get("ema") or ckpt.get("model")
            _walk_modules(root, "", tensors, set())
            if _names_out is not None:
                names = _names_of(ckpt)
                if names is not None:
                    _names_out["names"] = names
        if not tensors:
            raise ValueError(f"no tensors found in {path}")
    tensors = {re.sub(r"^(module\.|model\.model\.)", lambda m: "" if m.group(1) == "module." else "model.", k): v
               for k, v in tensors.items()}
    return state_from_tensors(tensors, variant, nc)
