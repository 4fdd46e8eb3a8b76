"""CPU: the C-ABI library loads and exports every symbol include/pgr_hip.h declares; no compute calls."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "pgr_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(pgr_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_exported():
    from pgrtk_amd import _ffi
    assert os.path.exists(_ffi.LIB_PATH), "build first: python __graft_entry__.py build"
    L = ctypes.CDLL(_ffi.LIB_PATH)
    declared = _declared_symbols()
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(L, name), "libpgrhip.so does not export " + name
    # the Python binding covers the same set
    assert sorted(_ffi.SYMBOLS) == declared


def test_struct_layouts():
    from pgrtk_amd import _ffi
    assert ctypes.sizeof(_ffi.Spec) == 20
    assert _ffi.MM128.itemsize == 16
    assert _ffi.FRAG_REC.itemsize == 40
    assert _ffi.HITPAIR.itemsize == 24


def test_no_cpu_fallback():
    """without a GPU the product must fail loudly, not compute on the host"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible here; the no-device path is checked in the build container")
    import pgrtk_amd
    with pytest.raises(pgrtk_amd.PgrError) as e:
        pgrtk_amd.Context(0)
    assert "no CPU fallback" in str(e.value) or "device" in str(e.value)


def test_product_does_not_import_oracle():
    """oracle/ is test infrastructure: nothing under pgr-tk_amd/ may reference it"""
    pkg = os.path.join(ROOT, "pgr-tk_amd")
    for d, _, files in os.walk(pkg):
        if "build" in d.split(os.sep):
            continue
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")) or f == "Makefile":
                txt = open(os.path.join(d, f), errors="ignore").read()
                assert "oracle" not in txt.lower() or f == "_never_", os.path.join(d, f)


def test_host_programs_built_and_fail_loudly_without_gpu(tmp_path):
    """the C++ host programs above the C ABI exist (built by __graft_entry__.build / make), print their usage, and --
    in a container without a GPU -- report the missing device instead of computing anything on the host"""
    import subprocess
    import torch
    bindir = os.path.join(ROOT, "pgr-tk_amd", "bin")
    for exe in ("pgr-mdb", "pgr-query", "pgr-pbundle-decomp"):
        path = os.path.join(bindir, exe)
        assert os.path.exists(path), "build first: python __graft_entry__.py build"
        r = subprocess.run([path], capture_output=True, text=True)
        assert r.returncode == 2 and "usage" in r.stderr
    if torch.cuda.is_available():
        return
    lst = tmp_path / "l.txt"
    lst.write_text("")
    r = subprocess.run([os.path.join(bindir, "pgr-mdb"), str(lst), str(tmp_path / "x")], capture_output=True, text=True)
    assert r.returncode == 1 and "pgr_ctx_create failed" in r.stderr
    assert not os.path.exists(str(tmp_path / "x.mdb"))
