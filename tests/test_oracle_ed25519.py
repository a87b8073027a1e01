"""Pin the Ed25519 CPU oracle against every golden vector the reference's own
tests hold for the hot path.  CPU only."""
import hashlib
import json
import os

import numpy as np
import pytest

from oracle import ed25519 as O

G = os.path.join(os.path.dirname(__file__), "golden")
MISC = json.load(open(os.path.join(G, "ed25519_misc.json")))
KAT = np.load(os.path.join(G, "ed25519_sign_input.npy"))


def test_constants():
    assert O.on_curve(O.B)
    assert O.mul_int(O.L, O.B) == O.IDENTITY
    assert O.SQRT_M1 * O.SQRT_M1 % O.P == O.P - 1
    # base point encoding 5866..66 (y = 4/5, x even)
    assert O.encode(O.B).hex() == "58" + "66" * 31


def test_rfc8032_fixed_base():
    # sign/eddsa/eddsa_test.go:24-52: pub = clamp(SHA512(seed)) * B
    for v in MISC["rfc8032"]:
        a = O.secret_scalar(bytes.fromhex(v["seed"]))
        assert O.mul_base(a).hex() == v["pub"]


def test_sign_input_fixed_base_all():
    # 1024 KATs: pub = a*B and R = r*B (eddsa.go:45-59, 91-142)
    for row in KAT:
        a, pub, r, R = (bytes(row[i]) for i in range(4))
        assert O.mul_base(a) == pub
        assert O.mul_base(r) == R


def test_sign_input_var_base():
    # verify equation S*B = R + h*A (eddsa.go:219-227) pins variable-base Mul
    for row in KAT[::8]:
        a, pub, r, R, h, S = (bytes(row[i]) for i in range(6))
        hA = O.decode(O.mul(h, pub))
        lhs = O.decode(O.mul_base(S))
        assert O.add(O.decode(R), hA) == lhs
        # vartime path agrees on reduced scalars
        assert O.mul(h, pub, vartime=True) == O.mul(h, pub)


def _verify(pk: bytes, msg: bytes, sig: bytes) -> bool:
    """sign/eddsa/eddsa.go:207-260 Verify semantics (no canonicality checks
    beyond S < L, which Verify enforces through scalar unmarshalling rules of
    the wycheproof set)."""
    if len(sig) != 64:
        return False
    A = O.decode(pk)
    R = O.decode(sig[:32])
    if A is None or R is None:
        return False
    s = int.from_bytes(sig[32:], "little")
    if s >= O.L:
        return False
    h = int.from_bytes(hashlib.sha512(sig[:32] + pk + msg).digest(), "little") % O.L
    lhs = O.mul_int(s, O.B)
    rhs = O.add(R, O.mul_int(h, A))
    return lhs == rhs


def test_wycheproof_verify_flags():
    bad = []
    for c in MISC["wycheproof"]:
        ok = _verify(bytes.fromhex(c["pk"]), bytes.fromhex(c["msg"]), bytes.fromhex(c["sig"]))
        if ok != c["valid"]:
            bad.append(c["id"])
    assert not bad, bad


def test_rfc9380_points_pin_add_and_cofactor_mul():
    for v in MISC["rfc9380"]:
        pt = O.hash_to_curve_from_u(int(v["u0"], 16), int(v["u1"], 16))
        assert pt == (int(v["x"], 16), int(v["y"], 16))


def test_small_order_list_decodes_and_has_small_order():
    for hx in MISC["small_order"]:
        pt = O.decode(bytes.fromhex(hx))
        assert pt is not None
        assert O.is_small_order(pt)


def test_noncanonical_decode_count():
    # point_test.go:66-96: 24 of the 38 encodings p+i (i<19, two sign bits) decode
    cnt = 0
    for i in range(19):
        for sign in (0, 1):
            n = (O.P + i) | (sign << 255)
            if O.decode(n.to_bytes(32, "little")) is not None:
                cnt += 1
    assert cnt == MISC["noncanonical_decodable_count"]


def test_mul_zero_is_identity_encoding():
    # util/test/test.go:364
    assert O.mul(bytes(32), O.encode(O.B)) == b"\x01" + bytes(31)
    assert O.mul_base(bytes(32)) == b"\x01" + bytes(31)


def test_consttime_high_bit_quirk():
    # a[31] > 127 -> top digit may be dropped by the const-time
    # path; vartime path multiplies by the true 256-bit integer.
    a = bytes([0xFF] * 32)
    e = O.recode_radix16(a)
    assert e[63] == 16
    k = O.effective_scalar_consttime(a)
    assert k == int.from_bytes(a, "little") - (16 << 252)
    assert O.mul(a, O.encode(O.B)) == O.encode(O.mul_int(k, O.B))
    assert O.mul(a, O.encode(O.B), vartime=True) == O.encode(O.mul_int(2**256 - 1, O.B))
    # l (unreduced primeOrderScalar, const.go:28) times a prime-order point = identity
    assert O.mul(O.L.to_bytes(32, "little"), O.encode(O.B)) == b"\x01" + bytes(31)


def test_decode_rejects_nonsquare():
    # y = 2 has no x on the curve
    assert O.decode((2).to_bytes(32, "little")) is None


def test_rfc9380_full_pipeline_hash_to_field_and_point():
    """point_test.go:369-445 TestHashToField / TestHashToPoint from the messages themselves."""
    dst = MISC["rfc9380_dst"].encode()
    for v in MISC["rfc9380"]:
        u0, u1 = O.hash_to_field(v["msg"].encode(), dst)
        assert (u0, u1) == (int(v["u0"], 16), int(v["u1"], 16))
        x, y = O.decode(O.hash_to_curve(v["msg"].encode(), dst))
        assert (x, y) == (int(v["x"], 16), int(v["y"], 16))
