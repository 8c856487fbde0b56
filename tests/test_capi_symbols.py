"""The C-ABI library loads on a machine without a GPU, exports every function include/orbamd.h
declares, and refuses to do work without a GPU (no CPU fallback).  No compute calls here."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "orbamd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(orbamd_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_expected_surface():
    names = declared_functions()
    for must in ("orbamd_ctx_create", "orbamd_detect", "orbamd_orient", "orbamd_describe", "orbamd_extract_bgr",
                 "orbamd_fetch", "orbamd_knn2", "orbamd_ratio_filter", "orbamd_extract_batch_dev",
                 "orbamd_knn2_batch_dev", "orbamd_knn2_cross_dev"):
        assert must in names


def test_library_exports_every_declared_symbol():
    from visual_slam_amd import _lib
    L = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared_functions():
        assert hasattr(L, name), "liborbamd.so does not export %s" % name


def test_python_binding_covers_the_header():
    from visual_slam_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared_functions()
    _lib.lib()   # binds argtypes for all of them
    assert _lib.lib().orbamd_version() >= 100
    assert _lib.status_string(0) == "ok" and _lib.status_string(-2) == "output capacity too small"


def test_pybind_module_imports_with_reference_name():
    import importlib
    m = importlib.import_module("build.orb_project")
    for fn in ("extract", "extract_gray", "extract_indexed", "knn_match2", "ratio_filter"):
        assert callable(getattr(m, fn))
    # input validation happens before any GPU work and mirrors the reference's messages
    with pytest.raises(RuntimeError, match="Expect HxW"):
        m.extract(np.zeros((4, 4, 2), np.uint8))
    with pytest.raises(RuntimeError, match="uint8 or float32"):
        m.extract(np.zeros((8, 8, 3), np.int16))


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.mark.skipif(_have_gpu(), reason="checks the no-GPU failure mode")
def test_no_gpu_means_loud_failure_not_fallback():
    from visual_slam_amd import _lib
    with pytest.raises(_lib.OrbAmdError):
        _lib.Context(0)
    import importlib
    m = importlib.import_module("build.orb_project")
    with pytest.raises(RuntimeError, match="no CPU path"):
        m.extract(np.zeros((32, 32, 3), np.uint8))
    from visual_slam_amd.engine import OrbEngine
    with pytest.raises(RuntimeError, match="no CPU path"):
        OrbEngine(0)
