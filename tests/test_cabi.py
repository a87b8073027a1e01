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


def test_go_binding_stays_in_step_with_the_header():
    """go/kyberhip/hip.go cannot be compiled here (no Go toolchain): at least every C function it calls must be
    declared in include/kyber_hip.h with the same number of arguments."""
    import re

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = open(os.path.join(root, "include", "kyber_hip.h")).read()
    g = open(os.path.join(root, "go", "kyberhip", "hip.go")).read()
    decl = {}
    for m in re.finditer(r"\b(?:int|const char \*)\s*(kyb_\w+)\s*\(([^;]*?)\);", h, re.S):
        args = [a for a in m.group(2).split(",") if a.strip() and a.strip() != "void"]
        decl[m.group(1)] = len(args)
    calls = 0
    for m in re.finditer(r"C\.(kyb_\w+)\(", g):
        name, i, depth = m.group(1), m.end(), 1
        j = i
        while depth:
            depth += {"(": 1, ")": -1}.get(g[j], 0)
            j += 1
        body = g[i:j - 1]
        n, d = (1 if body.strip() else 0), 0
        for ch in body:
            d += {"(": 1, ")": -1}.get(ch, 0)
            n += ch == "," and d == 0
        assert name in decl, name
        assert decl[name] == n, (name, decl[name], n)
        calls += 1
    assert calls >= 25
