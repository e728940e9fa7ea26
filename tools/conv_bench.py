"""Time single convolution shapes through the C ABI (hipEvent profile hooks): python tools/conv_bench.py [n]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

from eioku_amd import _lib, ops

SHAPES = [  # name, cin, cout, k, s, hw(in)
    ("model.7", 128, 256, 3, 2, 40), ("model.8.m", 128, 128, 3, 1, 20), ("model.19", 128, 128, 3, 2, 40),
    ("cv2.1.0", 128, 64, 3, 1, 40), ("cv3.1.0", 128, 80, 3, 1, 40), ("cv2.2.0", 256, 64, 3, 1, 20),
    ("cv3.2.0", 256, 80, 3, 1, 20), ("m6.m", 64, 64, 3, 1, 40), ("cv3.2.1", 80, 80, 3, 1, 20),
    ("m6.cv2", 256, 128, 1, 1, 40), ("m15.cv1", 192, 64, 1, 1, 80), ("m2.cv1", 32, 32, 1, 1, 160), ("m8.cv1", 256, 256, 1, 1, 20),
    ("cv3.1.1", 80, 80, 3, 1, 40), ("cv2.2.1", 64, 64, 3, 1, 20),
    ("m12.cv1", 384, 128, 1, 1, 40), ("m9.cv2", 512, 256, 1, 1, 20), ("m21.cv1", 384, 256, 1, 1, 20),
    ("cv2.0.0", 64, 64, 3, 1, 80), ("cv3.0.0", 64, 80, 3, 1, 80), ("cv3.0.1", 80, 80, 3, 1, 80), ("2.m", 16, 16, 3, 1, 160),
    ("4.m", 32, 32, 3, 1, 80), ("model.3", 32, 64, 3, 2, 160), ("model.16", 64, 64, 3, 2, 80),
    ("stem", 8, 16, 3, 2, 640), ("model.1", 16, 32, 3, 2, 320), ("model.5", 64, 128, 3, 2, 80),
]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    only = sys.argv[2:] or None
    _lib.init()
    _lib.prof_enable(True)
    rng = np.random.default_rng(0)
    for name, cin, cout, k, s, hw in SHAPES:
        if only and name not in only:
            continue
        x = torch.randn((n, hw, hw, cin), device="cuda", dtype=torch.float16)
        w = (rng.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32)
        b = np.zeros(cout, np.float32)
        for _ in range(2):
            y = ops.conv2d_f16(x, w, b, stride=s)
        torch.cuda.synchronize()
        _lib.prof_reset()
        reps = 10
        for _ in range(reps):
            y = ops.conv2d_f16(x, w, b, stride=s)
        torch.cuda.synchronize()
        ms, cnt = _lib.prof_read(2)
        us = ms * 1e3 / reps
        fl = 2.0 * y.shape[0] * y.shape[1] * y.shape[2] * cout * cin * k * k
        mb = (x.numel() + y.numel()) * 2 / 1e6
        print(f"{name:10s} {cin:4d}->{cout:4d} k{k}s{s} in{hw:4d}  {us:8.1f} us  {fl / us / 1e6:7.1f} TF/s  {mb / us * 1e3:7.0f} GB/s", flush=True)


main()
