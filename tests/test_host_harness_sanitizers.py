"""One AddressSanitizer + UndefinedBehaviorSanitizer pass over the device headers' host build (VERDICT r3 hygiene item):
tests/sanitize_main.cpp = tests/host_harness.cpp + a script interpreter, compiled with -fsanitize=address,undefined; a
script of calls across the suites (scalar multiplications, fixed-base tables under every policy, the cooperative slot
arithmetic, decoders on valid and malformed input, hashing, the scalar-field Horner) must give the plain library's
answers with a clean sanitizer report.  (shift-base is off: the signed-limb code shifts negative values left, which
C++20 defines as the two's-complement result every compiler this code meets produces; the device build is -std=c++17.)"""
import os
import random
import subprocess

import pytest

from oracle import bls12381 as OB, bn254 as ON4, bn256 as ON, ed25519 as OE
from tests import _host_harness as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "_build", "sanitize_main")


def _be(k):
    return k.to_bytes(32, "big")


def _script():
    rng = random.Random(2024)
    P1, P2 = OB.g1_mul(rng.randrange(1, OB.R), OB.G1_GEN), OB.g2_mul(rng.randrange(1, OB.R), OB.G2_GEN)
    N1, N2 = ON.g1_mul(rng.randrange(1, ON.ORDER), ON.G1_GEN), ON.g2_mul(rng.randrange(1, ON.ORDER), ON.G2_GEN)
    ks = [0, 1, OB.R - 1, OB.R, (1 << 256) - 1, OB.X_ABS**2, rng.randrange(1 << 256)]
    kb = b"".join(_be(k) for k in ks)
    calls = [
        ("hh_bls_g1_fb_mul", OB.g1_compress(P1), len(ks), kb, ("o", 48 * len(ks))),
        ("hh_bls_g2_fb_mul", OB.g2_compress(P2), 3, kb[:96], ("o", 96 * 3)),
        ("hh_bn_g1_fb_mul", ON.g1_marshal(N1), 3, kb[32:128], ("o", 64 * 3)),
        ("hh_bn_g2_fb_mul", ON.g2_marshal(N2), 2, kb[64:128], ("o", 128 * 2)),
        ("hh_bls_g1_mul", _be(ks[6]), OB.g1_compress(P1), ("o", 48)),
        ("hh_bls_g2_mul", _be(ks[6]), OB.g2_compress(P2), ("o", 96)),
        ("hh_bls_g1_mul", _be(5), bytes(48), ("o", 48)),                      # malformed: compression bit clear
        ("hh_bn_g1_mul", _be(ks[2]), ON.g1_marshal(N1), ("o", 64)),
        ("hh_bn_g2_mul", _be(ks[4]), ON.g2_marshal(N2), ("o", 128)),
        ("hh_bn4_g1_mul", _be(77), ON4.g1_marshal(ON4.G1_GEN), ("o", 64)),
        ("hh_ed_mul", bytes(rng.randrange(256) for _ in range(32)), OE.encode(OE.B), 0, ("o", 32)),
        ("hh_bls_g1_decode", b"\xff" * 48, 1),
        ("hh_bls_g2_decode", OB.g2_compress(P2), 1),
        ("hh_bls_g1_unmarshal", OB.g1_compress(P1), 4, ("o", 96)),
        ("hh_bls_g2_unmarshal", b"\x80" + bytes(95), 0, ("o", 96)),
        ("hh_bls_g1_coop", b"adaxn\0", OB.g1_compress(P1), OB.g1_compress(OB.g1_mul(3, P1)), ("o", 96)),
        ("hh_bls_g2_coop", b"aad\0", OB.g2_compress(P2), OB.g2_compress(P2), ("o", 192)),
        ("hh_bls_hash_g1", b"sanitize me", 11, b"BLS_SIG_BLS12381G1_XMD:SHA-256_SSWU_RO_NUL_", 43, ("o", 48)),
        ("hh_ed_hash", b"abc", 3, b"QUUX-V01-CS02-with-edwards25519_XMD:SHA-512_ELL2_RO_", 52, ("o", 32)),
        ("hh_scalar_poly_eval", 1, 3, (0).to_bytes(4, "little") + (7).to_bytes(4, "little") + (0xFFFFFFFF).to_bytes(4, "little"), 2,
         _be((1 << 256) - 1) + _be(OB.R - 1), ("o", 96)),
        ("hh_scalar_poly_eval", 0, 2, bytes(8), 0, b"", ("o", 64)),
    ]
    return calls


def _line(call):
    out = [call[0]]
    for a in call[1:]:
        if isinstance(a, tuple):
            out.append("o%d" % a[1])
        elif isinstance(a, int):
            out.append("i%d" % a)
        else:
            out.append("x" + bytes(a).hex())
    return " ".join(out)


@pytest.mark.timeout(1500)
def test_host_build_is_clean_under_asan_and_ubsan():
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-pthread", "-fsanitize=address,undefined", "-fno-sanitize=shift-base", "-fno-sanitize-recover=undefined",
                           "-fno-omit-frame-pointer", "-o", EXE, os.path.join(ROOT, "tests", "sanitize_main.cpp")])
    calls = _script()
    r = subprocess.run([EXE], input="\n".join(_line(c) for c in calls) + "\n", capture_output=True, text=True, timeout=1200,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1"))
    assert r.returncode == 0 and "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-3000:]
    lines = r.stdout.strip().split("\n")
    assert len(lines) == len(calls)
    for c, line in zip(calls, lines):
        osz = tuple(a[1] for a in c[1:] if isinstance(a, tuple))
        args = [a for a in c[1:] if not isinstance(a, tuple)]
        want = H.call(c[0], *args, out_sizes=osz)
        got = line.split(" ")
        assert int(got[0]) == want[0], (c[0], line[:80])
        assert [bytes.fromhex(x) for x in got[1:]] == list(want[1:]), c[0]
