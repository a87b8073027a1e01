"""The bn256 device code (bn256.cuh over mont.cuh / tower.cuh / curve.cuh), compiled for the host,
against the oracle restatement of pairing/bn256 and the reference's golden vectors."""
import json
import os
import random

import pytest

from oracle import bn256 as O
from tests import _host_harness as H


def _fp(x):
    return x.to_bytes(32, "big")


@pytest.fixture(scope="module")
def G(golden_dir):
    return json.load(open(os.path.join(golden_dir, "bn256.json")))


def test_fp_ops():
    rng = random.Random(1)
    vals = [0, 1, 2, O.P - 1, O.P - 2, O.P, O.P + 1, (1 << 256) - 1] + [rng.randrange(O.P) for _ in range(100)]
    for i in range(len(vals)):
        a, b = vals[i], vals[(i * 7 + 3) % len(vals)]
        assert H.call("hh_bn_fp_op", 0, _fp(a), _fp(b), out_sizes=(32,))[1] == _fp(a * b % O.P)
        assert H.call("hh_bn_fp_op", 1, _fp(a), _fp(b), out_sizes=(32,))[1] == _fp((a + b) % O.P)
        assert H.call("hh_bn_fp_op", 2, _fp(a), _fp(b), out_sizes=(32,))[1] == _fp((a - b) % O.P)
    for a in vals + [3, 1 << 32, (1 << 64) - 1, O.P - 3, (O.P + 1) // 2]:  # Kaliski inversion: every step count / edge
        exp = pow(a, -1, O.P) if a % O.P else 0
        assert H.call("hh_bn_fp_op", 4, _fp(a), _fp(0), out_sizes=(32,))[1] == _fp(exp)


def test_golden_bdn_fixtures(G):
    Hm = O.g1_marshal(O.hash_to_g1(G["bdn_msg"].encode()))
    base2 = O.g2_marshal(O.G2_GEN)
    for priv, pub, sig in zip(G["bdn_privs"], G["bdn_pubs"], G["bdn_sigs"]):
        k = bytes.fromhex(priv)
        assert H.call("hh_bn_g2_mul", k, base2, out_sizes=(128,)) == (0, bytes.fromhex(pub))
        assert H.call("hh_bn_g1_mul", k, Hm, out_sizes=(64,)) == (0, bytes.fromhex(sig))


def test_wire_edge_cases():
    x, y = O.G1_GEN
    assert H.call("hh_bn_g1_decode", bytes(64))[0] == 0
    assert H.call("hh_bn_g1_decode", _fp(5) + _fp(5))[0] == 1
    one = _fp(1)
    # non-canonical coordinate (x + p) is accepted and reduced
    assert H.call("hh_bn_g1_mul", one, _fp(x + O.P) + _fp(y), out_sizes=(64,)) == (0, O.g1_marshal(O.G1_GEN))
    # infinity in -> 64 zero bytes out; (p, p) is infinity too
    assert H.call("hh_bn_g1_mul", _fp(7), bytes(64), out_sizes=(64,)) == (0, bytes(64))
    assert H.call("hh_bn_g1_mul", _fp(7), _fp(O.P) * 2, out_sizes=(64,)) == (0, bytes(64))
    assert H.call("hh_bn_g2_mul", _fp(7), bytes(128), out_sizes=(128,)) == (0, bytes(128))
    assert H.call("hh_bn_g2_decode", _fp(1) * 4)[0] == 1


def test_scalar_mul_vs_oracle():
    rng = random.Random(4)
    ks = [0, 1, 2, 16, 17, O.ORDER - 1, O.ORDER, O.ORDER + 3, (1 << 256) - 1] + [rng.randrange(O.ORDER) for _ in range(5)]
    h = rng.randrange(1, O.ORDER)
    p1 = O.g1_marshal(O.g1_mul(h, O.G1_GEN))
    p2 = O.g2_marshal(O.g2_mul(h, O.G2_GEN))
    for k in ks:
        assert H.call("hh_bn_g1_mul", _fp(k), p1, out_sizes=(64,)) == (0, O.g1_mul_bytes(_fp(k), p1)), hex(k)
        assert H.call("hh_bn_g2_mul", _fp(k), p2, out_sizes=(128,)) == (0, O.g2_mul_bytes(_fp(k), p2)), hex(k)



def test_fp_sqr_dedicated_path():
    rng = random.Random(12)
    for a in [0, 1, O.P - 1, (1 << 256) - 1] + [rng.randrange(O.P) for _ in range(300)]:
        assert H.call("hh_bn_fp_op", 5, _fp(a), _fp(0), out_sizes=(32,))[1] == _fp(a * a % O.P)


def test_hash_g1_reference_fixtures_and_lengths(G):
    import hashlib

    for h in G["hash_g1"]:
        msg = bytes.fromhex(h["msg_hex"])
        assert H.call("hh_bn_hash_g1", msg, len(msg), out_sizes=(64,)) == (0, bytes.fromhex(h["point"]))
    msg = G["bdn_msg"].encode()
    assert H.call("hh_bn_hash_g1", msg, len(msg), out_sizes=(64,))[1] == O.g1_marshal(O.hash_to_g1(msg))
    # SHA-256 padding boundaries: lengths around 55 / 56 / 64 / 119 / 120 bytes, and the empty message
    for ln in (0, 1, 54, 55, 56, 57, 63, 64, 65, 119, 120, 121, 200):
        msg = bytes((7 * i + ln) & 0xFF for i in range(ln))
        assert H.call("hh_bn_hash_g1", msg or b"\\x00", ln, out_sizes=(64,))[1] == O.g1_marshal(O.hash_to_g1(msg)), ln


def test_g1_glv_split_edge_scalars():
    """G1 multiplication goes through the GLV split k = k1 + k2 lambda (Babai rounding with 2^256 fixed-point
    constants): scalars at the rounding boundaries, >= n, and random ones, against the plain big-int oracle."""
    rng = random.Random(9)
    n = O.ORDER
    u = 1868033 ** 3
    lam = 36 * u ** 3 + 18 * u ** 2 + 6 * u + 1
    a2 = 254952053719217182022156415784332439563
    edge = [0, 1, 2, 15, 16, n - 1, n, n + 1, lam - 1, lam, lam + 1, a2 - 1, a2, a2 + 1, (1 << 128) - 1, 1 << 128,
            (1 << 256) - 1, 1 << 255, (1 << 256) - n, 3 * lam, n // 2, n // 2 + 1, a2 * a2 % (1 << 256)]
    ks = edge + [rng.randrange(1 << 256) for _ in range(60)]
    h = 0xABCDEF123
    P = O.g1_mul(h, O.G1_GEN)
    pb = O.g1_marshal(P)
    for k in ks:
        kb = k.to_bytes(32, "big")
        assert H.call("hh_bn_g1_mul", kb, pb, out_sizes=(64,)) == (0, O.g1_marshal(O.g1_mul(k % n, P))), hex(k)
    inf = O.g1_marshal(None)
    assert H.call("hh_bn_g1_mul", (77).to_bytes(32, "big"), inf, out_sizes=(64,)) == (0, inf)


def test_unmarshal_wire():
    """g*_unmarshal_wire = UnmarshalBinary + MarshalBinary (pairing/bn256/point.go:170-238, 423-499)."""
    rng = random.Random(31)
    x, y = O.G1_GEN
    for _ in range(4):
        p1, p2 = O.g1_mul(rng.randrange(1, O.ORDER), O.G1_GEN), O.g2_mul(rng.randrange(1, O.ORDER), O.G2_GEN)
        assert H.call("hh_bn_g1_unmarshal", O.g1_marshal(p1), out_sizes=(64,)) == (0, O.g1_marshal(p1))
        assert H.call("hh_bn_g2_unmarshal", O.g2_marshal(p2), out_sizes=(128,)) == (0, O.g2_marshal(p2))
    assert H.call("hh_bn_g1_unmarshal", bytes(64), out_sizes=(64,)) == (0, bytes(64))
    assert H.call("hh_bn_g2_unmarshal", bytes(128), out_sizes=(128,)) == (0, bytes(128))
    assert H.call("hh_bn_g1_unmarshal", _fp(5) + _fp(5), out_sizes=(64,)) == (1, bytes(64))
    assert H.call("hh_bn_g2_unmarshal", _fp(1) * 4, out_sizes=(128,)) == (1, bytes(128))
    # a non-canonical coordinate is accepted and comes back reduced
    assert H.call("hh_bn_g1_unmarshal", _fp(x + O.P) + _fp(y), out_sizes=(64,)) == (0, O.g1_marshal(O.G1_GEN))
    # G2: on the twist but outside the order-r subgroup is accepted, as in the reference (no subgroup check)
    xx = 1
    while True:
        X = (xx, 1)
        y2 = O.f2_sqrt(O.f2_add(O.f2_mul(O.f2_sqr(X), X), O.TWIST_B))
        if y2 is not None and O.g2_mul(O.ORDER, (X, y2)) is not None:
            break
        xx += 1
    buf = O.g2_marshal((X, y2))
    assert H.call("hh_bn_g2_unmarshal", buf, out_sizes=(128,)) == (0, buf)


def test_g2_gls_multiplication_for_vouched_points():
    """bn256's G2 is unchecked, so the default is the reference's plain ladder; with KYB_F_TRUSTED the caller vouches for
    the subgroup and the 4-dimensional GLS walk runs: same multiples on subgroup points."""
    rng = random.Random(77)
    lam = 6 * O.U * O.U
    q = O.g2_mul(rng.randrange(1, O.ORDER), O.G2_GEN)
    for k in [0, 1, 16, O.ORDER - 1, O.ORDER, (1 << 256) - 1, lam, lam - 1, lam * lam % O.ORDER] + [rng.getrandbits(256) for _ in range(6)]:
        kb = k.to_bytes(32, "big")
        assert H.call("hh_bn_g2_mul_f", kb, O.g2_marshal(q), 0x100, out_sizes=(128,)) == (0, O.g2_marshal(O.g2_mul(k, q))), k
        assert H.call("hh_bn_g2_mul_f", kb, O.g2_marshal(q), 0, out_sizes=(128,)) == (0, O.g2_marshal(O.g2_mul(k, q))), k


def test_xyzz_piece_accumulator_vs_oracle():
    """The MSM's bucket-piece accumulator (curve.cuh Xyzz): plain runs, doubling, cancellation, restart, infinity inputs."""
    import random
    rng = random.Random(43)
    for grp in (1, 2):
        gen, mul, add, neg, mar, size, fn = (
            (O.G1_GEN, O.g1_mul, O.g1_add, O.g1_neg, O.g1_marshal, 64, "hh_bn_g1_xyzz_sum") if grp == 1 else
            (O.G2_GEN, O.g2_mul, O.g2_add, O.g2_neg, O.g2_marshal, 128, "hh_bn_g2_xyzz_sum"))
        pts = [mul(rng.randrange(1, O.ORDER), gen) for _ in range(5)]
        P, Q = pts[0], pts[1]
        runs = [
            [(p, rng.random() < 0.5) for p in pts],
            [(P, False)],
            [],
            [(P, False), (P, False), (Q, True)],
            [(P, False), (P, True)],
            [(P, False), (P, True), (Q, False), (Q, False), (Q, False)],
            [(None, False), (P, True), (None, True), (neg(P), True), (Q, False)],
        ]
        for run in runs:
            exp = None
            for p, s in run:
                exp = add(exp, neg(p) if s else p)
            wire = b"".join(mar(p) for p, _ in run) or b"\x00"
            signs = bytes(int(s) for _, s in run) or b"\x00"
            assert H.call(fn, len(run), wire, signs, out_sizes=(size,)) == (0, mar(exp)), (grp, run)


def test_fixed_base_table_multiplication_vs_oracle():
    """fixed_base.cuh on bn256: G1, and G2 with a base OUTSIDE the order-n subgroup (the reference accepts it and its
    double-and-add multiplies by the plain integer): a twist point of order 13 n makes table entries hit infinity."""
    import random
    rng = random.Random(78)
    n = O.ORDER
    ks = [0, 1, 2, 13, 26, 127, 128, 129, 255, 256, 257, 511, 512, 513, 1023, 1024, 1025, (1 << 250) + 512, (512 << 10) + 513, n - 1, n, n + 1, (1 << 256) - 1, 1 << 255, 13 * n, 13 * n - 1,
          int.from_bytes(b"\x80" * 32, "big"), int.from_bytes(b"\x81" * 32, "big")] + [rng.randrange(1 << 256) for _ in range(12)]
    ks = [k % (1 << 256) for k in ks]
    kb = b"".join(k.to_bytes(32, "big") for k in ks)
    P = O.g1_mul(rng.randrange(1, n), O.G1_GEN)
    st, out = H.call("hh_bn_g1_fb_mul", O.g1_marshal(P), len(ks), kb, out_sizes=(64 * len(ks),))
    assert st == 0
    for i, k in enumerate(ks):
        assert out[64 * i:64 * i + 64] == O.g1_marshal(O.g1_mul(k, P)), hex(k)
    # a twist point with a component of order 13: cofactor 2p - n = 13 * 7369 * ...
    cof = 2 * O.P - n
    while True:
        x = (rng.randrange(O.P), rng.randrange(O.P))
        y = O.f2_sqrt(O.f2_add(O.f2_mul(O.f2_sqr(x), x), O.TWIST_B)) if hasattr(O, "f2_sqrt") else None
        if y is None:
            continue
        T = (x, y)
        T13 = O.g2_mul(cof // 13 * n, T)  # order 1 or 13
        if T13 is not None:
            break
    for Q in (O.g2_mul(rng.randrange(1, n), O.G2_GEN), T13, O.g2_add(T13, O.g2_mul(5, O.G2_GEN))):
        st, out = H.call("hh_bn_g2_fb_mul", O.g2_marshal(Q), len(ks), kb, out_sizes=(128 * len(ks),))
        assert st == 0
        for i, k in enumerate(ks):
            assert out[128 * i:128 * i + 128] == O.g2_marshal(O.g2_mul(k, Q)), hex(k)


def test_hash_g1_svdw_reference_vectors_and_padding_boundaries(G):
    """HashG1 (pairing/bn256/hash.go:10-110) through the device headers on the CPU: the 11 outputs of
    hash_test.go:45-57 (TestKnownHashes: HashG1([]byte{i}, nil)), then message lengths around the SHA-256 / HMAC block
    boundaries and domain separation tags of every key-length class (empty, short, one block, longer than a block) against
    the oracle's restatement"""
    for v in G["hash_g1_svdw"]:
        msg, dst = bytes.fromhex(v["msg_hex"]), bytes.fromhex(v["dst_hex"])
        assert H.call("hh_bn_hash_g1_svdw", msg, len(msg), dst or b"\x00", len(dst), out_sizes=(64,)) == (0, bytes.fromhex(v["point"]))
    for ln in (0, 1, 31, 32, 54, 55, 56, 63, 64, 65, 119, 120, 200):
        msg = bytes((5 * i + ln) & 0xFF for i in range(ln))
        for dl in (0, 1, 13, 63, 64, 65, 128, 255):
            dst = bytes((3 * i + dl) & 0xFF for i in range(dl))
            want = O.g1_marshal(O.hash_g1_svdw(msg, dst))
            assert H.call("hh_bn_hash_g1_svdw", msg or b"\x00", ln, dst or b"\x00", dl, out_sizes=(64,))[1] == want, (ln, dl)


def test_svdw_map_edge_inputs():
    """mapToCurve at t = 0 (w0 = 0: the inversion's 0 -> 0) and other fixed field elements, oracle-side: the restatement
    lands on the curve for each and on x1 for t = 0 only when (s - 1) / 2 is a valid abscissa"""
    for t in (0, 1, O.P - 1, 2, (O.P - 1) // 2, (O.P + 1) // 2):
        x, y = O.map_to_curve(t)
        assert (y * y - x * x * x - 3) % O.P == 0
