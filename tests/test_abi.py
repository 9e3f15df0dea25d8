"""The C-ABI library loads without a GPU and exports every symbol include/rt_native.h declares; the host library
exports every symbol of host/rth_api.h; without a GPU the context constructor fails loudly (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

import harness as h

rt = h.rt
ROOT = h.ROOT


def declared(header, prefix):
    text = open(os.path.join(ROOT, header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(" + prefix + r"[a-z0-9_]+)\s*\(", text)))


def test_native_exports_every_declared_symbol():
    lib = rt.native_lib()
    names = declared("include/rt_native.h", "rt_")
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f"librtnative.so does not export {n}"


def test_host_exports_every_declared_symbol():
    lib = rt.host_lib()
    names = declared("ray-tracer_amd/host/rth_api.h", "rth_")
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"librthost.so does not export {n}"


def test_null_handles_are_rejected_without_touching_a_gpu():
    lib = rt.native_lib()
    assert lib.rt_sync(None) < 0
    assert lib.rt_dispatch(None, 0, 1, 1, 1) < 0
    assert lib.rt_set_int(None, b"FrameRand", 1) < 0
    assert lib.rt_buffer_count(None) == 0
    lib.rt_buffer_release(None)   # null-safe like ComputeHelper.Release
    lib.rt_texture_release(None)
    lib.rt_destroy(None)


def test_no_cpu_fallback():
    """On a machine without a gfx950 GPU rt_create must fail (the product path never computes on the CPU)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    h_ = C.c_void_p()
    assert rt.native_lib().rt_create(0, C.byref(h_)) < 0
    with pytest.raises(rt.NativeError):
        rt.RtContext(0)


def test_product_does_not_reference_the_oracle():
    """Nothing under the product package includes, links or loads anything from oracle/."""
    pkg = os.path.join(ROOT, "ray-tracer_amd")
    for base, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hip")):
                text = open(os.path.join(base, f), errors="ignore").read()
                assert "oracle/" not in text and "rt_oracle" not in text and "librtoracle" not in text, f
    for f in ("bench.py",):
        text = open(os.path.join(ROOT, f)).read()
        assert text.count("import harness") == 1  # only inside cpu_baseline()
