"""The C-ABI library loads and exports every symbol include/eioku_hip.h declares (no GPU)."""
import ctypes
import re
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def header_functions():
    text = (ROOT / "include" / "eioku_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(eioku_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_something():
    names = header_functions()
    assert "eioku_init" in names and "eioku_scene_sad_luma" in names


def test_every_declared_symbol_is_exported(built_lib):
    raw = ctypes.CDLL(str((ROOT / "eioku_amd" / "libeioku_hip.so")))
    missing = [n for n in header_functions() if not hasattr(raw, n)]
    assert not missing, f"declared in eioku_hip.h but not exported: {missing}"


def test_python_binding_table_matches_header(built_lib):
    from eioku_amd import _lib

    assert sorted(_lib.SIGNATURES) == header_functions()


def test_abi_version_and_error_string(built_lib):
    assert built_lib.eioku_abi_version() == 1
    assert isinstance(built_lib.eioku_last_error(), bytes)


def test_compute_without_init_fails_loudly(built_lib):
    """No silent fallback: a compute entry point refuses to run before eioku_init."""
    import numpy as np
    from eioku_amd import _lib

    if _lib._initialised_device is not None:
        return  # a GPU test already initialised the library in this process
    out = np.zeros(2, dtype=np.uint64)
    y = np.zeros((2, 4, 4), dtype=np.uint8)
    rc = built_lib.eioku_scene_sad_luma(y.ctypes.data, 2, 4, 4, 4, 16, None, out.ctypes.data, 0, None)
    assert rc == -4 and b"eioku_init" in built_lib.eioku_last_error()


def test_product_package_never_imports_oracle():
    """The oracle is test infrastructure: nothing under eioku_amd/ may reference it."""
    offenders = []
    for p in (ROOT / "eioku_amd").rglob("*.py"):
        if re.search(r"^\s*(from|import)\s+oracle\b", p.read_text(), flags=re.M):
            offenders.append(str(p))
    assert not offenders
