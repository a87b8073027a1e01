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


def test_empty_batches_and_argument_errors_never_touch_the_device():
    """n = 0 returns KYB_OK and a null / ill-sized argument returns KYB_E_ARG before any device work, so both are
    checkable without a GPU; a real call without a GPU must fail loudly (no CPU fallback behind the ABI)."""
    import torch

    lib = _lib.load()
    buf = (ctypes.c_uint8 * 4096)()
    p = ctypes.addressof(buf)
    zero_ok = [
        ("kyb_ed25519_mul_base", (0, p, p, 0)),
        ("kyb_ed25519_mul", (0, p, p, p, p, 0)),
        ("kyb_ed25519_unmarshal", (0, p, p, p)),
        ("kyb_ed25519_add", (0, p, p, p, p)),
        ("kyb_bls12381_g1_mul", (0, p, p, p, p, 0)),
        ("kyb_bls12381_g2_unmarshal", (0, p, p, p, 0)),
        ("kyb_bls12381_pair", (0, p, p, p, p, 0)),
        ("kyb_bls12381_pair_check", (0, p, p, p, p, p, p, 0)),
        ("kyb_bn256_g1_unmarshal", (0, p, p, p, 0)),
        ("kyb_bn256_pair", (0, p, p, p, p, 0)),
        ("kyb_bn256_gt_mul", (0, p, p, p, p)),
    ]
    for name, args in zero_ok:
        assert getattr(lib, name)(*args) == 0, name
    bad = [
        ("kyb_ed25519_mul", (4, None, p, p, p, 0)),
        ("kyb_ed25519_unmarshal", (4, p, None, p)),
        ("kyb_ed25519_mul_base", (4, p, p, 1 | 8)),  # KYB_F_VARTIME | KYB_F_UNIFORM: exclusive
        ("kyb_ed25519_mul_same_base", (4, p, p, p, p, 16)),  # an unknown flag bit
        ("kyb_bls12381_g1_mul", (4, p, None, p, p, 0)),
        ("kyb_bls12381_g1_unmarshal", (4, None, p, p, 0)),
        ("kyb_bls12381_g1_mul_dev", (4, p, p, 47, p, p, 0, None)),  # stride that is neither 0 nor the wire size
        ("kyb_bn256_g2_unmarshal_dev", (4, p, None, p, 0, None)),
        ("kyb_bn256_pair_dev", (4, p, p, None, p, 0, None)),
    ]
    for name, args in bad:
        rc = getattr(lib, name)(*args)
        assert rc != 0, name
        assert name.encode() in lib.kyb_last_error() or b"bad argument" in lib.kyb_last_error()
    # the Python mirrors shape-check before calling in
    from kyber_amd.pairing import bls12381 as bls

    out, st = bls.g1_batch_unmarshal(b"")
    assert out.shape == (0, 48) and st.shape == (0,)
    out, st = bls.g1_batch_unmarshal(b"", bls.F_UNCOMPRESSED_OUT)
    assert out.shape == (0, 96)
    try:
        bls.g1_batch_mul(bytes(64), bytes(48))  # 2 scalars, 1 point
    except ValueError:
        pass
    else:
        raise AssertionError("length mismatch must raise")
    if not torch.cuda.is_available():
        try:
            bls.g1_batch_mul(bytes(32), bytes(48))
        except _lib.KyberHipError as e:
            assert "rc=" in str(e)
        else:
            raise AssertionError("a compute call without a GPU must fail loudly")


def test_stream_release_without_a_device_reports_an_error_code():
    import torch

    lib = _lib.load()
    rc = lib.kyb_stream_release(None)
    if torch.cuda.is_available():
        assert rc == 0
    else:
        assert rc != 0 and lib.kyb_last_error()


# Method sets of the reference's interfaces, as of /root/reference (dedis/kyber v4): group.go:23-131 (Scalar, Point),
# :175-183 (Group), encoding.go:15-32 (Marshaling), pairing/pairing.go:8-20 (Suite with the embedded Encoding,
# HashFactory, XOFFactory, Random).  When the reference tree is present they are re-extracted and compared.
_POINT = ["Equal", "Null", "Base", "Pick", "Set", "Clone", "EmbedLen", "Embed", "Data", "Add", "Sub", "Neg", "Mul"]
_MARSHALING = ["MarshalBinary", "UnmarshalBinary", "String", "MarshalSize", "MarshalTo", "UnmarshalFrom"]
_GROUP = ["String", "ScalarLen", "Scalar", "PointLen", "Point"]
_PAIRING = ["G1", "G2", "GT", "Pair", "ValidatePairing", "New", "Read", "Write", "Hash", "XOF", "RandomStream"]


def _go_methods(src: str, recv: str):
    import re

    return set(re.findall(r"^func \(\w+ \*" + recv + r"\) (\w+)\(", src, re.M))


def test_go_suite_implements_every_method_of_the_kyber_interfaces():
    """go/kyberhip/suite cannot be compiled here: at least every method kyber.Point (+ Marshaling), kyber.Group and
    pairing.Suite declare must exist on the engine's types, and the hot ones must reach the engine."""
    import re

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = os.path.join(root, "go", "kyberhip", "suite")
    point = open(os.path.join(d, "point.go")).read()
    group = open(os.path.join(d, "group.go")).read()
    pairing = open(os.path.join(d, "pairing.go")).read()
    suites = open(os.path.join(d, "suites.go")).read()
    assert set(_POINT + _MARSHALING) <= _go_methods(point, "Point")
    assert {"AllowVarTime", "Hash", "IsInCorrectGroup"} <= _go_methods(point, "Point")
    assert set(_GROUP) <= _go_methods(group, "Group")
    assert {"BatchMul", "Commit", "MSM", "Validate"} <= _go_methods(group, "Group")
    assert set(_PAIRING) <= _go_methods(pairing, "PairingSuite")
    assert {"BatchPair", "BatchValidatePairing", "BatchVerify", "BatchGTMul"} <= _go_methods(pairing, "PairingSuite")
    # the n = 1 policy: a single Mul / Pair / ValidatePairing is the embedded
    # reference's unless SingleOpOnDevice says otherwise; the device path calls the binding, never the reference;
    # batches below MinDeviceBatch loop over the reference
    mul = point[point.index("func (P *Point) Mul("):point.index("func (P *Point) mulOnDevice(")]
    assert "if !SingleOpOnDevice" in mul and "P.p.Mul(s, un(p))" in mul and "P.mulOnDevice(s, p)" in mul
    dev = point[point.index("func (P *Point) mulOnDevice("):point.index("// AllowVarTime")]
    assert "P.g.mul(" in dev and "P.g.mulBase(" in dev and "P.p.Mul(" not in dev
    assert re.search(r"var SingleOpOnDevice = false", point) and re.search(r"MinDeviceBatch\s+= \d+", point) and re.search(r"MinDevicePairings = \d+", point)
    pair = pairing[pairing.index("func (s *PairingSuite) Pair("):pairing.index("// ---- kyber.Encoding")]
    assert "s.inner.Pair(un(p1), un(p2))" in pair and "s.inner.ValidatePairing(un(p1), un(p2), un(inv1), un(inv2))" in pair
    assert "s.pairOnDevice(" in pair and "s.validateOnDevice(" in pair
    for fn in ("BatchMul", "Commit", "MSM"):
        body = group[group.index("func (g *Group) %s(" % fn):]
        assert "MinDeviceBatch && !SingleOpOnDevice" in body[:1200], fn
    for fn in ("BatchPair", "BatchValidatePairing"):
        body = pairing[pairing.index("func (s *PairingSuite) %s(" % fn):]
        assert "MinDevicePairings && !SingleOpOnDevice" in body[:800], fn
    assert "hip.Bls12381Pair(" in pairing and "hip.Bn256ValidatePairing(" in pairing and "hip.Bls12381VerifyG1(" in pairing
    # the suites are declared variable-time and registered with the variable-time suites only
    patch = open(os.path.join(root, "go", "patches", "suites_all_vartime.patch")).read()
    assert "!constantTime" in patch and "RequireConstantTime" in patch and "register(hipsuite." in patch
    assert "ariable-time" in point
    for ctor in ("NewBlakeSHA256Ed25519HIP", "NewSuiteBLS12381", "NewSuiteBn256", "NewGroupSuiteBLS12381"):
        assert "func " + ctor + "(" in suites
    # every hip.X the suite uses exists in the binding
    binding = open(os.path.join(root, "go", "kyberhip", "hip.go")).read() + open(os.path.join(root, "go", "kyberhip", "batch.go")).read()
    for name in set(re.findall(r"\bhip\.([A-Z]\w+)", point + group + pairing + suites)):
        assert re.search(r"\b(func|type) " + name + r"\b|^\s+" + name + r"\b", binding, re.M), name
    ref = "/root/reference"
    if os.path.isdir(ref):  # the build container: re-extract the interfaces and compare
        g = open(os.path.join(ref, "group.go")).read()

        def iface(src, name):
            body = src[src.index("type %s interface {" % name):]
            body = body[:body.index("\n}")]
            return [m for m in re.findall(r"^\t(\w+)\(", body, re.M)]

        assert iface(g, "Point") == _POINT
        assert iface(g, "Group") == _GROUP
        assert iface(open(os.path.join(ref, "encoding.go")).read(), "Marshaling") == ["String", "MarshalSize", "MarshalTo", "UnmarshalFrom"]
        assert iface(open(os.path.join(ref, "pairing", "pairing.go")).read(), "Suite") == ["G1", "G2", "GT", "Pair", "ValidatePairing"]


def test_go_packages_reference_only_what_package_hip_defines():
    """No Go toolchain here: at least every `hip.X` the suite / benchmark packages use must be a function, constant or
    type of go/kyberhip (package hip), and braces / parentheses must balance outside strings and comments."""
    import glob
    import re

    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "go", "kyberhip")
    defs = set()
    for f in glob.glob(os.path.join(root, "*.go")):
        src = open(f).read()
        defs |= set(re.findall(r"^func (\w+)\(", src, re.M))
        defs |= set(re.findall(r"^(?:const|var|type) (\w+)", src, re.M))
        for blk in re.findall(r"(?:const|var) \((.*?)\n\)", src, re.S):
            defs |= set(re.findall(r"^\t(\w+)", blk, re.M))
    used = set()
    files = glob.glob(os.path.join(root, "**", "*.go"), recursive=True)
    assert len(files) >= 7
    for f in files:
        src = open(f).read()
        code = re.sub(r'"(?:[^"\\]|\\.)*"|`[^`]*`|//[^\n]*', "", src)
        assert code.count("{") == code.count("}") and code.count("(") == code.count(")"), f
        if os.path.dirname(f) != root:
            used |= set(re.findall(r"\bhip\.(\w+)", code))
    assert used and not (used - defs), sorted(used - defs)
