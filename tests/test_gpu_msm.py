"""GPU parity tests for the multi-scalar multiplication pipeline (config 3), through the C ABI:
bit-exact against the sequential Mul + Add sum of the oracles (the reference's own shape,
share/poly.go:340-348), plus size-independent checks at batch scale."""
import hashlib
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _shake(label, n):
    return np.frombuffer(hashlib.shake_256(label).digest(n), dtype=np.uint8)


# ------------------------------------------------------------------ Ed25519
@pytest.fixture(scope="module")
def ed():
    import torch

    assert torch.cuda.is_available()
    from kyber_amd.group import edwards25519 as ed

    return ed


def _ed_inputs(ed, n, label=b"msm"):
    s = _shake(label + b"/s", n * 32).reshape(n, 32).copy()
    h = _shake(label + b"/h", n * 32).reshape(n, 32).copy()
    s[:, 31] &= 0x0F
    h[:, 31] &= 0x0F
    return s, ed.batch_mul_base(h)


@pytest.mark.parametrize("n", [0, 1, 2, 3, 7, 64, 1000, 4096])
def test_ed25519_msm_vs_c_oracle(ed, n):
    from tests import _oracle_c as OC

    s, P = _ed_inputs(ed, max(n, 1))
    s, P = s[:n], P[:n]
    out, st = ed.msm(s, P)
    assert not st.any()
    exp, rc = OC.ed_msm(s, P)
    assert rc == 0 and bytes(out) == bytes(exp)


def test_ed25519_msm_edge_scalars_and_skew(ed):
    from oracle import ed25519 as O

    n = 300
    s, P = _ed_inputs(ed, n, b"edge")
    s[0] = 0
    s[1] = np.frombuffer((1).to_bytes(32, "little"), dtype=np.uint8)
    s[2] = 0xFF  # 2^256 - 1: the reference's recoding drops its top digit -- Mul multiplies by -1 (oracle effective_scalar_consttime)
    s[3] = np.frombuffer((O.L - 1).to_bytes(32, "little"), dtype=np.uint8)
    s[4] = np.frombuffer(((1 << 255) + 12345).to_bytes(32, "little"), dtype=np.uint8)
    s[100:200] = s[5]  # many equal scalars -> same bucket in every window
    P[100:200] = P[6]  # ... and equal points: exercises doubling inside a bucket
    P[7] = np.frombuffer(bytes([1]) + bytes(31), dtype=np.uint8)  # identity as an input
    out, st = ed.msm(s, P)
    assert not st.any()
    acc = O.IDENTITY
    for i in range(n):
        # the sum of the reference's N x Mul: every scalar counts as the integer geScalarMult multiplies by
        acc = O.add(acc, O.mul_int(O.effective_scalar_consttime(bytes(s[i])) % (8 * O.L), O.decode(bytes(P[i]))))
    assert bytes(out) == O.encode(acc)


def test_ed25519_msm_bad_point_zeroes_output(ed):
    s, P = _ed_inputs(ed, 16, b"bad")
    P[5] = np.frombuffer(bytes([2]) + bytes(31), dtype=np.uint8)  # y = 2 is not on the curve
    out, st = ed.msm(s, P)
    assert st[5] == 1 and st.sum() == 1 and not out.any()


def test_ed25519_msm_2p16_matches_single_mul(ed):
    """P_i = h_i B  =>  sum s_i P_i = (sum s_i h_i mod l) B: size-independent check at 2^16."""
    from oracle import ed25519 as O
    import torch

    n = 1 << 16
    s = _shake(b"big/s", n * 32).reshape(n, 32).copy()
    h = _shake(b"big/h", n * 32).reshape(n, 32).copy()
    s[:, 31] &= 0x0F
    h[:, 31] &= 0x0F
    P = ed.batch_mul_base(h)
    out, st = ed.msm(torch.from_numpy(s).cuda(), torch.from_numpy(P).cuda())
    torch.cuda.synchronize()
    assert not st.any().item()
    tot = sum(int.from_bytes(bytes(s[i]), "little") * int.from_bytes(bytes(h[i]), "little") for i in range(n)) % O.L
    assert bytes(out.cpu().numpy()) == bytes(ed.batch_mul_base(tot.to_bytes(32, "little"))[0])


# ------------------------------------------------------------ pairing suites
def _suite(name):
    import importlib

    return importlib.import_module("kyber_amd.pairing." + name), importlib.import_module("oracle." + name)


def _be_scalars(label, n, order):
    raw = hashlib.shake_256(label).digest(n * 64)
    return np.stack([np.frombuffer((int.from_bytes(raw[64 * i:64 * i + 64], "big") % order).to_bytes(32, "big"),
                                   dtype=np.uint8) for i in range(n)]) if n else np.zeros((0, 32), dtype=np.uint8)


@pytest.mark.parametrize("name", ["bls12381", "bn256"])
def test_pairing_suite_msm_vs_oracle_small(name):
    m, O = _suite(name)
    order = m.ORDER
    rng = random.Random(3)
    z2 = 0xD201000000010000 ** 2  # the BLS12-381 G1 split base: scalars around its multiples / half-multiples
    extra = [z2 // 2, z2 // 2 + 1, z2 - 1, z2, z2 + 1, z2 * (z2 // 2), z2 * (z2 // 2 + 1) + z2 // 2 + 1, 3 * z2 * z2 // 2,
             (1 << 256) - z2, order + z2 // 2]
    n = 19 + len(extra)
    ks = [0, 1, order - 1, (1 << 256) - 1, 1 << 255] + extra + [rng.randrange(order) for _ in range(n - 5 - len(extra))]
    ks = [k % (1 << 256) for k in ks]
    hs = [rng.randrange(1, order) for _ in range(n)]
    kb = b"".join(k.to_bytes(32, "big") for k in ks)
    if name == "bls12381":
        enc1, enc2, G1, G2 = O.g1_compress, O.g2_compress, O.G1_GEN, O.G2_GEN
    else:
        enc1, enc2, G1, G2 = O.g1_marshal, O.g2_marshal, O.G1_GEN, O.G2_GEN
    p1 = [O.g1_mul(h, G1) for h in hs]
    p2 = [O.g2_mul(h, G2) for h in hs]
    p1[16], p2[16] = None, None  # infinity as an input
    p1[18], p2[18] = p1[17], p2[17]
    ks[18] = ks[17]  # equal (scalar, point) pairs: doubling inside a bucket
    kb = b"".join(k.to_bytes(32, "big") for k in ks)
    acc1 = acc2 = None
    for k, a, b in zip(ks, p1, p2):
        acc1 = O.g1_add(acc1, O.g1_mul(k, a))
        acc2 = O.g2_add(acc2, O.g2_mul(k, b))
    out, st = m.g1_msm(kb, b"".join(enc1(p) for p in p1))
    assert not st.any() and bytes(out) == enc1(acc1)
    out, st = m.g2_msm(kb, b"".join(enc2(p) for p in p2))
    assert not st.any() and bytes(out) == enc2(acc2)
    # n = 0 -> identity ; a bad point -> status + zero output
    out, st = m.g1_msm(b"", b"")
    assert bytes(out) == m.G1_NULL
    bad = bytearray(b"".join(enc1(p) for p in p1))
    bad[m.G1_LEN * 2:m.G1_LEN * 3] = (b"\x00" * (m.G1_LEN - 1) + b"\x05") if name == "bls12381" else (5).to_bytes(32, "big") * 2
    out, st = m.g1_msm(kb, bytes(bad))
    assert st[2] != 0 and not out.any()


@pytest.mark.parametrize("name,n", [("bls12381", 1 << 14), ("bn256", 1 << 14)])
def test_pairing_suite_msm_at_scale_matches_single_mul(name, n):
    """P_i = h_i G  =>  sum k_i P_i = (sum k_i h_i mod r) G."""
    m, O = _suite(name)
    k, h = _be_scalars(b"sc/k/" + name.encode(), n, m.ORDER), _be_scalars(b"sc/h/" + name.encode(), n, m.ORDER)
    tot = sum(int.from_bytes(bytes(k[i]), "big") * int.from_bytes(bytes(h[i]), "big") for i in range(n)) % m.ORDER
    P, st = m.g1_commit(h)
    out, st2 = m.g1_msm(k, P)
    assert not st.any() and not st2.any()
    exp, _ = m.g1_commit(tot.to_bytes(32, "big"))
    assert bytes(out) == bytes(exp[0])
    Q, st = m.g2_commit(h[:2048])
    out, st2 = m.g2_msm(k[:2048], Q)
    tot2 = sum(int.from_bytes(bytes(k[i]), "big") * int.from_bytes(bytes(h[i]), "big") for i in range(2048)) % m.ORDER
    exp, _ = m.g2_commit(tot2.to_bytes(32, "big"))
    assert bytes(out) == bytes(exp[0])


def test_bdn_aggregate_key_fixture_via_msm(golden_dir):
    """sum (c_i + 1) * P_i, P_i = (i+1) G2: sign/bdn/bdn_vartime_test.go:24-48 through the G2 MSM."""
    import json, os

    m, O = _suite("bn256")
    G = json.load(open(os.path.join(golden_dir, "bn256.json")))
    P, _ = m.g2_commit(b"".join((i + 1).to_bytes(32, "big") for i in range(3)))
    cp1 = b"".join((int(c, 16) + 1).to_bytes(32, "big") for c in G["bdn_coefs"])
    out, st = m.g2_msm(cp1, P)
    assert not st.any() and bytes(out).hex() == G["bdn_agg_key"]


def test_ed25519_msm_all_equal_scalars_skew(ed):
    """Every scalar equal: all points fall into ONE bucket per window (the piece-splitting path)."""
    from oracle import ed25519 as O

    n = 5000
    s, P = _ed_inputs(ed, n, b"skew")
    s[:] = s[0]
    out, st = ed.msm(s, P)
    assert not st.any()
    # sum_i k P_i = k * sum_i P_i ; sum of points via an MSM with unit scalars, then one var-base mul
    ones = np.zeros((n, 32), dtype=np.uint8)
    ones[:, 0] = 1
    tot, _ = ed.msm(ones, P)
    exp, st2 = ed.batch_mul(s[0], tot)
    assert bytes(out) == bytes(exp[0])


def test_short_scalar_msm_and_wave_uniform_vartime():
    """KYB_F_SCALAR_BITS: an MSM over scalars below 2^b run with the flag equals the plain one (and bits at and above
    b are ignored); KYB_F_VARTIME on Ed25519: short scalars, sparse digits and full-length ones all match the oracle."""
    import hashlib

    from kyber_amd.group import edwards25519 as ed
    from kyber_amd.pairing import bls12381 as bls, bn256 as bn
    from oracle import ed25519 as O

    n = 3000
    raw = np.frombuffer(hashlib.shake_256(b"short/k").digest(n * 32), dtype=np.uint8).reshape(n, 32).copy()
    for m in (bn, bls):
        k = raw.copy()
        k[:, :15] = 0           # big-endian: 136 bits left ...
        k[:, 15] &= 1           # ... 129
        for grp, commit, msm in ((1, m.g1_commit, m.g1_msm), (2, m.g2_commit, m.g2_msm)):
            P = np.asarray(commit(raw[:n] & 0x3F)[0])
            full, st = msm(k, P)
            short, st2 = msm(k, P, m.F_SCALAR_BITS(129))
            assert not np.asarray(st).any() and not np.asarray(st2).any()
            assert bytes(np.asarray(full)) == bytes(np.asarray(short)), (m.__name__, grp)
            junk = k.copy()
            junk[:, 0] = 0xA5   # bits above 2^129 are ignored (BLS12-381 G1 too since round 6: b <= 160 takes the plain adapter)
            ign, _ = msm(junk, P, m.F_SCALAR_BITS(129))
            assert bytes(np.asarray(ign)) == bytes(np.asarray(short))
    # Ed25519 (little-endian scalars)
    s = raw.copy()
    s[:, 17:] = 0
    s[:, 16] &= 1
    h = raw.copy()
    h[:, 31] &= 0x0F
    P = ed.batch_mul_base(h)
    full, _ = ed.msm(s, P)
    short, _ = ed.msm(s, P, scalar_bits=129)
    assert bytes(full) == bytes(short)
    # variable-time multiplication: short / sparse / full scalars against the oracle
    m = 256
    sc = np.zeros((m, 32), dtype=np.uint8)
    sc[:64, :16] = raw[:64, :16]                 # 128-bit scalars: the wave starts at digit 32
    sc[64:128, 3] = raw[64:128, 3]               # one non-zero byte: most windows add nothing in any lane
    sc[128:192] = raw[128:192]                   # all 256 bits (geScalarMultVartime honours them)
    sc[192:, 0] = np.arange(64, dtype=np.uint8)  # tiny scalars incl. 0
    out, st = ed.batch_mul(sc, P[:m], vartime=True)
    base = ed.batch_mul_base(sc, vartime=True)
    assert not np.asarray(st).any()
    for i in list(range(0, m, 7)) + [192, 193, 255]:
        assert bytes(out[i]) == O.mul(bytes(sc[i]), bytes(P[i]), vartime=True), i
        # (fixed-base with the flag: the plain 256-bit multiple -- the reference itself always takes geScalarMultBase)
        assert bytes(base[i]) == O.encode(O.mul_int(int.from_bytes(bytes(sc[i]), "little"), O.B)), i
    # mixed lengths inside one wave
    mix = sc[[0, 70, 130, 200] * 16]
    out2, _ = ed.batch_mul(mix, P[:64], vartime=True)
    for i in range(0, 64, 5):
        assert bytes(out2[i]) == O.mul(bytes(mix[i]), bytes(P[i]), vartime=True), i
