"""GPU: the regression net for round 2's fault class (an over-read past a tensor's end that only faults on a page
boundary).  libeioku_hip_bc.so is the same library with conv.hip compiled -DEIOKU_BOUNDS_CHECK: every
global access of an activation / residual / image / output tensor is compared with the tensor's extent and a violation
is counted (not performed).  A child process runs tests/test_conv_gpu.py's CASES, whole v8n / v8s / v8m forwards at
96 x 160, detect() on five source geometries and the Places365 ResNet18 through it; the count must stay 0."""
import ctypes as C
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
pytestmark = pytest.mark.gpu


def test_regular_library_reports_that_it_is_not_the_bounds_check_build(gpu, built_lib):
    v, ln = C.c_int(7), C.c_int(7)
    assert built_lib.eioku_debug_bounds(C.byref(v), C.byref(ln), 0, 0) == 0
    assert v.value == -1


def test_no_access_of_the_conv_family_leaves_its_tensor(gpu):
    lib = ROOT / "eioku_amd" / "libeioku_hip_bc.so"
    assert lib.exists(), "build it: make -C eioku_amd/csrc (target all builds both libraries)"
    env = dict(os.environ, EIOKU_HIP_LIB=str(lib))
    r = subprocess.run([sys.executable, str(ROOT / "tools" / "bounds_probe.py")], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["library"].endswith("libeioku_hip_bc.so")
    assert out["selftest"] == 3, out            # the net catches an out-of-extent load before, after and a store
    assert out["after_cases"] == 0, out
    assert out["violations"] == 0, f"{out['violations']} out-of-extent accesses, last at conv.hip:{out['line']}"
    assert out["launches"] > 40
