"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol that
include/agt_hip.h declares (no compute calls without a GPU); the product refuses to run
without its HIP extension."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "agt_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(agt_[a-z0-9_]+)\s*\(", text)))


def test_build_and_exports():
    import __graft_entry__ as g
    g.build()
    from accurate_aprilgroup_tracking_amd import hiplib
    lib = ctypes.CDLL(hiplib.LIB_PATH)
    syms = declared_symbols()
    assert len(syms) >= 18
    for s in syms:
        assert hasattr(lib, s), "libagt_hip.so does not export %s" % s
    assert sorted(hiplib.SYMBOLS) == syms, "hiplib.SYMBOLS out of sync with include/agt_hip.h"
    lib.agt_version.restype = ctypes.c_int
    assert lib.agt_version() == 505
    lib.agt_error_string.restype = ctypes.c_char_p
    assert lib.agt_error_string(-4) == b"bad point count"
    assert lib.agt_tracker_state_size() > 0
    # argument errors are reported without touching a device
    assert lib.agt_create(None, None, None) == -1
    assert lib.agt_destroy(None) == 0


def test_gfx950_code_object_present():
    from accurate_aprilgroup_tracking_amd import hiplib
    blob = open(hiplib.LIB_PATH, "rb").read()
    assert b"gfx950" in blob and b"lk_kernel" in blob and b"pnp_kernel" in blob and b"pyr_down_kernel" in blob


def test_fails_loudly_without_extension_or_gpu(monkeypatch):
    from accurate_aprilgroup_tracking_amd import hiplib
    monkeypatch.setattr(hiplib, "_lib", None)
    monkeypatch.setattr(hiplib, "LIB_PATH", "/nonexistent/libagt_hip.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        hiplib.lib()
    import torch
    if not torch.cuda.is_available():
        from accurate_aprilgroup_tracking_amd import cv_hip
        import numpy as np
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            cv_hip.solvePnP(np.zeros((6, 3)), np.zeros((6, 2)), np.eye(3), None)
        with pytest.raises(RuntimeError):
            cv_hip.calcOpticalFlowPyrLK(np.zeros((32, 32), np.uint8), np.zeros((32, 32), np.uint8), np.zeros((1, 2)))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "accurate_aprilgroup_tracking_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text and "cvoracle" not in text, f
                assert "libcvoracle" not in text, f
