"""GPU box, run with EIOKU_HIP_LIB=eioku_amd/libeioku_hip_bc.so (tests/test_bounds_gpu.py starts it as a child process):
the conv family's test cases, whole forwards of three model sizes and detect() on four source geometries through the
bounds-check build; prints one JSON line {"selftest": ..., "violations": ..., "line": ..., "launches": ...}."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from eioku_amd import _lib, detect, ops, places  # noqa: E402
from oracle import prng  # noqa: E402


def bounds(reset=False, selftest=False):
    v, ln = C.c_int(0), C.c_int(0)
    _lib.check(_lib.load().eioku_debug_bounds(C.byref(v), C.byref(ln), int(reset), int(selftest)), "eioku_debug_bounds")
    return v.value, ln.value


def main():
    _lib.init(0)
    gpu = torch.device("cuda:0")
    out = {"library": str(_lib.LIB_PATH)}
    v, _ = bounds(reset=True, selftest=True)
    out["selftest"] = v  # 3 violations made on purpose
    assert bounds(reset=True)[0] == 0
    launches = 0
    # 1. the parity cases of tests/test_conv_gpu.py: every kernel family, ragged shapes, slices, residuals
    from test_conv_gpu import CASES

    rng = np.random.default_rng(0)
    for (n, h, w, cin, cout, k, stride) in CASES:
        x = rng.standard_normal((n, h, w, cin)).astype(np.float16)
        wt = (rng.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32)
        b = rng.standard_normal(cout).astype(np.float32)
        for res in (False, True):
            if res and (stride != 1 or cout % 4):
                continue
            ho, wo = (h + 2 * (k // 2) - k) // stride + 1, (w + 2 * (k // 2) - k) // stride + 1
            r = rng.standard_normal((n, ho, wo, cout)).astype(np.float16) if res else None
            ops.conv2d_f16(torch.from_numpy(x).to(gpu), wt, b, stride=stride, silu=True,
                           residual=torch.from_numpy(r).to(gpu) if res else None)
            launches += 1
    out["after_cases"] = bounds()[0]
    # 2. whole forwards at 96 x 160 (every layer of v8n / v8s / v8m incl. the fused pairs, chains, flat and generic kernels)
    from eioku_amd import weights as W

    for variant in ("n", "s", "m"):
        det = detect.Yolov8Detector(variant, 80, W.random_state(variant, 80, 7))
        x = torch.from_numpy(rng.standard_normal((2, 96, 160, 8)).astype(np.float16)).to(gpu)
        x[..., 3:] = 0
        det.forward_raw(x)
        launches += 1
        # 3. detect() on the source geometries that take the fused front ends: copy (640-wide), decimating (1080p, 1/3),
        #    2x2 area (720p, 1/2), general bilinear (480p) - the last frame of the batch ends the buffer
        if variant == "n":
            for (h, w) in ((640, 640), (1080, 1920), (720, 1280), (480, 854), (270, 480)):
                f = torch.from_numpy(prng.synth_frames_bgr(5, 2, h, w)).to(gpu)
                det.detect(f, conf=0.25)
                det.detect(f, conf=0.02)
                launches += 2
                out.setdefault("per_geometry", {})[f"{h}x{w}"] = bounds()
        det.close()
    # 4. the ResNet18 of classify_places (K4's kernels with the ReLU epilogues on its shapes)
    clf = places.Places365Classifier(places.random_state(3))
    clf.classify(prng.synth_frames_bgr(6, 3, 120, 160), 5)
    clf.close()
    launches += 1
    torch.cuda.synchronize()
    out["violations"], out["line"] = bounds()
    out["launches"] = launches
    print(json.dumps(out))


main()
