"""GPU parity tests for the bn254 suite (pairing/bn254: alt_bn128), through the C ABI: the reference's own hash-to-curve
vectors, bit-exact against the oracle restatement (oracle/bn254.py), the suite's strict decoding rules, and
size-independent properties at batch scale."""
import hashlib
import json
import os
import random

import numpy as np
import pytest

from oracle import bn254 as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def bn():
    import torch

    assert torch.cuda.is_available()
    from kyber_amd.pairing import bn254 as bn

    return bn


@pytest.fixture(scope="module")
def G(golden_dir):
    return json.load(open(os.path.join(golden_dir, "bn254.json")))


def _scalars(label: bytes, n: int) -> np.ndarray:
    raw = hashlib.shake_256(label).digest(n * 64)
    out = np.empty((n, 32), dtype=np.uint8)
    for i in range(n):
        out[i] = np.frombuffer((int.from_bytes(raw[64 * i:64 * i + 64], "big") % O.ORDER).to_bytes(32, "big"), dtype=np.uint8)
    return out


def _fp(x):
    return x.to_bytes(32, "big")


def _off_subgroup_g2(rng):
    while True:
        x = (rng.randrange(O.P), rng.randrange(O.P))
        y = O.f2_sqrt(O.f2_add(O.f2_mul(O.f2_sqr(x), x), O.TWIST_B))
        if y is not None:
            assert not O.g2_in_subgroup((x, y))
            return (x, y)


def test_hash_g1_reference_vectors_and_batch(bn, G):
    dst = G["hash_g1_dst"].encode()
    for h in G["hash_g1"]:                                    # point_test.go:14-48
        out, st = bn.batch_hash_g1([bytes.fromhex(h["msg_hex"])], dst)
        assert st[0] == 0 and bytes(out[0]).hex() == h["point"]
    # hash_to_field vectors (point_test.go:81-101) feed the map: whole hashes against the oracle, grouped by length
    d2 = G["h2f_dst"].encode()
    by_len = {}
    for v in G["hash_to_field"]:
        m = bytes.fromhex(v["msg"])
        by_len.setdefault(len(m), []).append(m)
    checked = 0
    for ln, ms in sorted(by_len.items())[:12]:
        out, st = bn.batch_hash_g1(ms, d2)
        assert not st.any()
        for i, m in enumerate(ms[:3]):
            assert bytes(out[i]) == O.g1_marshal(O.hash_to_g1(m, d2)), ln
            checked += 1
    assert checked >= 12
    msgs = [hashlib.sha256(b"m%d" % i).digest() for i in range(500)]
    out, st = bn.batch_hash_g1(msgs)
    assert not st.any()
    for i in range(0, 500, 37):
        assert bytes(out[i]) == O.g1_marshal(O.hash_to_g1(msgs[i]))
    for ln in (0, 1, 100, 101, 135, 136, 137, 300):          # around the 136-byte Keccak rate (after the 136-byte Z_pad)
        m = bytes((3 * i + ln) & 0xFF for i in range(ln))
        out, st = bn.batch_hash_g1([m, m])
        assert bytes(out[1]) == O.g1_marshal(O.hash_to_g1(m)), ln


def test_mul_vs_oracle_and_strict_decoding(bn):
    rng = random.Random(4)
    ks = [0, 1, 2, 16, 17, O.ORDER - 1, O.ORDER, O.ORDER + 3, (1 << 256) - 1] + [rng.randrange(O.ORDER) for _ in range(15)]
    hs = [rng.randrange(1, O.ORDER) for _ in ks]
    p1 = [O.g1_marshal(O.g1_mul(h, O.G1_GEN)) for h in hs]
    p2 = [O.g2_marshal(O.g2_mul(h, O.G2_GEN)) for h in hs]
    p1[2], p2[2] = bytes(64), bytes(128)
    kb = [_fp(k) for k in ks]
    out, st = bn.g1_batch_mul(b"".join(kb), b"".join(p1))
    assert not st.any()
    for i in range(len(ks)):
        assert bytes(out[i]) == O.g1_mul_bytes(kb[i], p1[i]), i
    out, st = bn.g2_batch_mul(b"".join(kb), b"".join(p2))
    assert not st.any()
    for i in range(len(ks)):
        assert bytes(out[i]) == O.g2_mul_bytes(kb[i], p2[i]), i
    out_t, st = bn.g2_batch_mul(b"".join(kb), b"".join(p2), bn.F_TRUSTED(0))   # validated operands: same bytes
    assert not st.any() and (out_t == out).all()
    # rejected: off the curve; a coordinate >= p (gfp.go:101-118; bn256 would reduce it); G2 outside the subgroup
    x, y = O.G1_GEN
    out, st = bn.g1_batch_mul(kb[1] * 3, _fp(5) * 2 + _fp(x + O.P) + _fp(y) + p1[0])
    assert list(st) == [1, 1, 0] and not out[:2].any()
    off = O.g2_marshal(_off_subgroup_g2(rng))
    out, st = bn.g2_batch_mul(kb[3] * 2, off + p2[0])
    assert list(st) == [2, 0] and not out[0].any()
    out, st = bn.g2_batch_mul(kb[3] * 2, off + p2[0], bn.F_TRUSTED(0))          # the caller's word is taken: no status
    assert list(st) == [0, 0] and bytes(out[1]) == O.g2_mul_bytes(kb[3], p2[0])  # (and a result only for the valid row)


def test_pairing_bytes_vs_oracle(bn):
    rng = random.Random(5)
    n = 6
    g1 = [O.g1_marshal(O.g1_mul(rng.randrange(1, O.ORDER), O.G1_GEN)) for _ in range(n)]
    g2 = [O.g2_marshal(O.g2_mul(rng.randrange(1, O.ORDER), O.G2_GEN)) for _ in range(n)]
    g1[3] = bytes(64)
    g2[4] = bytes(128)
    g1[0], g2[0] = bn.G1_BASE, bn.G2_BASE
    gt, st = bn.batch_pair(b"".join(g1), b"".join(g2))
    assert not st.any()
    for i in range(n):
        assert bytes(gt[i]) == O.pair_bytes(g1[i], g2[i]), i
    gt2, st = bn.batch_pair(b"".join(g1), b"".join(g2), bn.F_TRUSTED(0) | bn.F_TRUSTED(1))
    assert not st.any() and (gt2 == gt).all()
    # a G2 operand outside the subgroup: status, zero output, its neighbours unaffected
    g2[1] = O.g2_marshal(_off_subgroup_g2(rng))
    gt3, st = bn.batch_pair(b"".join(g1), b"".join(g2))
    assert list(st) == [0, 2, 0, 0, 0, 0] and not gt3[1].any() and (gt3[[0, 2, 3, 4, 5]] == gt[[0, 2, 3, 4, 5]]).all()


def test_bilinearity_and_validate_pairing_at_scale(bn):
    """e(aP, bQ) == e(abP, Q) == e(P, abQ) (suite_test.go:240-251) for 2048 independent pairs; ValidatePairing truth table."""
    n = 2048
    a, b = _scalars(b"bn4/a", n), _scalars(b"bn4/b", n)
    ab = np.empty_like(a)
    for i in range(n):
        v = int.from_bytes(bytes(a[i]), "big") * int.from_bytes(bytes(b[i]), "big") % O.ORDER
        ab[i] = np.frombuffer(_fp(v), dtype=np.uint8)
    aP, _ = bn.g1_commit(a)
    abP, _ = bn.g1_commit(ab)
    bQ, _ = bn.g2_commit(b)
    abQ, _ = bn.g2_commit(ab)
    G1 = np.tile(np.frombuffer(bn.G1_BASE, dtype=np.uint8), (n, 1))
    G2 = np.tile(np.frombuffer(bn.G2_BASE, dtype=np.uint8), (n, 1))
    e1, s1 = bn.batch_pair(aP, bQ)
    e2, s2 = bn.batch_pair(abP, G2)
    e3, s3 = bn.batch_pair(G1, abQ)
    assert not (s1.any() or s2.any() or s3.any())
    assert (e1 == e2).all() and (e1 == e3).all()
    assert len({bytes(r) for r in e1[:64]}) == 64
    for i in (0, 777, n - 1):                                 # and the bytes are the oracle's
        assert bytes(e1[i]) == O.pair_bytes(bytes(aP[i]), bytes(bQ[i]))
    forged = abP.copy()
    forged[::97] = aP[::97]
    ok, st = bn.batch_validate_pairing(aP, bQ, forged, G2)
    exp = np.ones(n, dtype=np.uint8)
    exp[::97] = 0
    assert not st.any() and (np.asarray(ok) == exp).all()
    inf1, inf2 = np.zeros_like(aP), np.zeros_like(bQ)
    ok, st = bn.batch_validate_pairing(inf1, bQ, G1, inf2)    # 1 == 1
    assert not st.any() and np.asarray(ok).all()


def test_suite_mirror_and_bls_scheme(bn):
    s = bn.NewSuite()
    a, b = s.G1().Scalar().SetInt64(5), s.G1().Scalar().SetInt64(7)
    P, Q = s.G1().Point().Mul(a, None), s.G2().Point().Mul(b, None)
    ab = s.G1().Scalar().Mul(a, b)
    assert s.Pair(P, Q).Equal(s.Pair(s.G1().Point().Mul(ab, None), s.G2().Point().Base()))
    assert s.ValidatePairing(P, Q, s.G1().Point().Mul(ab, None), s.G2().Point().Base())
    # sign/bls over the suite (pairing/bn254/bls_test.go:12-16): sig = x H(m), e(H(m), X) == e(sig, G2)
    from kyber_amd.sign.bls import NewSchemeOnG1_bn254

    sch = NewSchemeOnG1_bn254()
    rng = random.Random(77)
    privs = [_fp(rng.randrange(1, O.ORDER)) for _ in range(6)]
    pubs, st = bn.g2_commit(b"".join(privs))
    msgs = [b"msg %d" % i * (i + 1) for i in range(6)]
    sigs = [sch.sign(privs[i], msgs[i]) for i in range(6)]
    for i in (0, 5):
        assert sigs[i] == O.g1_marshal(O.g1_mul(int.from_bytes(privs[i], "big"), O.hash_to_g1(msgs[i])))
    ok = sch.batch_verify([bytes(p) for p in pubs], msgs, sigs)
    assert ok.all()
    bad = list(sigs)
    bad[2] = sigs[3]
    assert list(sch.batch_verify([bytes(p) for p in pubs], msgs, bad, keys_validated=True)) == [True, True, False, True, True, True]


def test_gt_mul_vs_oracle_and_homomorphism(bn):
    rng = random.Random(8)
    n = 4
    g1 = [O.g1_marshal(O.g1_mul(rng.randrange(1, O.ORDER), O.G1_GEN)) for _ in range(n)]
    g2 = [O.g2_marshal(O.g2_mul(rng.randrange(1, O.ORDER), O.G2_GEN)) for _ in range(n)]
    gt, _ = bn.batch_pair(b"".join(g1), b"".join(g2))
    ks = [0, 1, O.ORDER - 1, rng.randrange(O.ORDER)]
    kb = b"".join(_fp(k) for k in ks)
    out, st = bn.gt_batch_mul(kb, gt)
    assert not st.any()
    for i in range(n):
        assert bytes(out[i]) == O.gt_mul_bytes(_fp(ks[i]), bytes(gt[i])), i
    bad = bytearray(bytes(gt[0]))
    bad[:32] = _fp(O.P)                                      # a coefficient >= p is rejected (point.go:662-735)
    out, st = bn.gt_batch_mul(_fp(3), bytes(bad))
    assert st[0] == 1 and not out[0].any()
    m = 512
    k = _scalars(b"bn4/gt/k", m)
    P, _ = bn.g1_commit(_scalars(b"bn4/gt/p", m))
    G2 = np.tile(np.frombuffer(bn.G2_BASE, dtype=np.uint8), (m, 1))
    e, _ = bn.batch_pair(P, G2)
    kP, _ = bn.g1_batch_mul(k, P)
    ek, _ = bn.batch_pair(kP, G2)
    out, st = bn.gt_batch_mul(k, e)
    assert not st.any() and (out == ek).all()


def test_msm_add_unmarshal(bn):
    rng = random.Random(41)
    n = 3000
    k = _scalars(b"bn4/msm/k", n)
    P, _ = bn.g1_commit(_scalars(b"bn4/msm/p", n))
    Q, _ = bn.g2_commit(_scalars(b"bn4/msm/q", 600))
    out, st = bn.g1_msm(k, P)
    assert not np.asarray(st).any()
    acc = None
    terms, _ = bn.g1_batch_mul(k, P)
    for t in terms[:64]:
        acc = O.g1_add(acc, O.g1_unmarshal(bytes(t)))
    o64, _ = bn.g1_msm(k[:64], P[:64])
    assert bytes(np.asarray(o64)) == O.g1_marshal(acc)
    ones = np.zeros((n, 32), dtype=np.uint8)
    ones[:, 31] = 1
    rhs, _ = bn.g1_msm(ones, terms)
    assert bytes(np.asarray(out)) == bytes(np.asarray(rhs))
    o2, st = bn.g2_msm(k[:600], Q)
    o2t, _ = bn.g2_msm(k[:600], Q, bn.F_TRUSTED(0))
    assert not np.asarray(st).any() and bytes(np.asarray(o2)) == bytes(np.asarray(o2t))
    acc = None
    t2, _ = bn.g2_batch_mul(k[:40], Q[:40], bn.F_TRUSTED(0))
    for t in t2:
        acc = O.g2_add(acc, O.g2_unmarshal(bytes(t)))
    o40, _ = bn.g2_msm(k[:40], Q[:40])
    assert bytes(np.asarray(o40)) == O.g2_marshal(acc)
    s, st = bn.g1_batch_add(P[:100], P[100:200])
    assert not st.any() and bytes(s[7]) == O.g1_marshal(O.g1_add(O.g1_unmarshal(bytes(P[7])), O.g1_unmarshal(bytes(P[107]))))
    # batch UnmarshalBinary: strict coordinates, subgroup check on G2 unless vouched for
    x, y = O.G1_GEN
    batch = [bytes(P[0]), bytes(64), _fp(5) + _fp(5), _fp(x + O.P) + _fp(y)]
    out, st = bn.g1_batch_unmarshal(b"".join(batch))
    assert list(st) == [0, 0, 1, 1] and bytes(out[0]) == batch[0] and not out[2:].any()
    off = O.g2_marshal(_off_subgroup_g2(rng))
    batch = [bytes(Q[0]), bytes(128), _fp(1) * 4, off]
    out, st = bn.g2_batch_unmarshal(b"".join(batch))
    assert list(st) == [0, 0, 1, 2] and bytes(out[0]) == batch[0] and not out[2:].any()
    out, st = bn.g2_batch_unmarshal(b"".join(batch), bn.F_TRUSTED(0))
    assert list(st) == [0, 0, 1, 0] and bytes(out[3]) == off
