"""The C-ABI library builds, loads and exports every symbol include/*.h declares.
No compute calls here (no GPU needed)."""
import ctypes
import glob
import os
import re

from kyber_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    syms = set()
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        src = open(h).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        syms |= set(re.findall(r"\b(kyb_[a-z0-9_]+)\s*\(", src))
    return syms


def test_library_loads_and_exports_all_declared_symbols():
    lib = _lib.load()
    declared = _declared_symbols()
    assert len(declared) >= 8
    for s in declared:
        assert hasattr(lib, s), f"{s} declared in include/ but not exported"
    # and the Python binding table covers exactly the header
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)


def test_version_and_error_string_without_gpu():
    lib = _lib.load()
    assert lib.kyb_version() >= 1
    assert isinstance(lib.kyb_last_error(), bytes)


def test_headers_cite_reference_lines():
    src = open(os.path.join(ROOT, "include", "kyber_hip.h")).read()
    for cite in ("ge.go:373", "ge.go:443", "share/poly.go:143", "kilic/suite.go:57-68", "optate.go:266",
                 "share/poly.go:340-348", "point.go:630-662"):
        assert cite in src


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    try:
        _lib.load()
    except _lib.KyberHipError as e:
        assert "no CPU fallback" in str(e).replace("There is no", "no")
    else:
        raise AssertionError("load() must raise when the HIP library is missing")
