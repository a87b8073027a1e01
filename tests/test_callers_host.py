"""Host logic of the caller-level mirrors (no GPU): BLAKE2Xs coefficients of sign/bdn against the reference's fixtures."""
import json
import os

from kyber_amd.sign import bdn
from oracle import bn256 as O


def test_bdn_coefficients_match_reference_fixture(golden_dir):
    G = json.load(open(os.path.join(golden_dir, "bn256.json")))
    pubs = [O.g2_marshal(O.g2_mul(i, O.G2_GEN)) for i in (1, 2, 3)]
    assert [f"{c:032x}" for c in bdn.hash_point_to_r(pubs)] == G["bdn_coefs"]


def _oracle_backed_ed25519(monkeypatch):
    """share/poly over the Ed25519 mirror types with every engine call answered by the Python oracle (no GPU): the
    host logic -- share selection, Lagrange bases, which scalars meet which points -- is what is under test."""
    import numpy as np

    from kyber_amd.group import edwards25519 as ed
    from kyber_amd.share import poly
    from oracle import ed25519 as E

    def rows(bs):
        return np.frombuffer(b"".join(bs), dtype=np.uint8).reshape(len(bs), 32)

    def mul_same_base(scal, base):
        pt = E.encode(E.B) if base is None else bytes(base)
        return rows([E.mul(scal[i:i + 32], pt, vartime=True) for i in range(0, len(scal), 32)])

    def msm(scal, pts):
        acc = E.IDENTITY
        for i in range(0, len(scal), 32):
            acc = E.add(acc, E.mul_int(int.from_bytes(scal[i:i + 32], "little"), E.decode(pts[i:i + 32])))
        n = len(scal) // 32
        return np.frombuffer(E.encode(acc), dtype=np.uint8), np.zeros(n, dtype=np.uint8)

    def batch_add(a, b):
        out = [E.encode(E.add(E.decode(a[i:i + 32]), E.decode(b[i:i + 32]))) for i in range(0, len(a), 32)]
        return rows(out), np.zeros(len(out), dtype=np.uint8)

    monkeypatch.setattr(poly, "_ops", lambda group: (mul_same_base, msm, 32))
    monkeypatch.setattr(poly, "_add_op", lambda group: batch_add)
    return ed.NewSuite(), poly, E


def test_share_poly_recover_pub_poly_pairing_group_scalars(monkeypatch):
    """The same host logic with the pairing suites' scalar mirror (mod.Int, big-endian) on bn256 G1, engine calls
    answered by the bn256 oracle."""
    import random

    import numpy as np

    from kyber_amd.pairing import bn256
    from kyber_amd.share import poly

    def rows(bs):
        return np.frombuffer(b"".join(bs), dtype=np.uint8).reshape(len(bs), 64)

    def mul_same_base(scal, base):
        pt = O.G1_GEN if base is None else O.g1_unmarshal(bytes(base))
        return rows([O.g1_marshal(O.g1_mul(int.from_bytes(scal[i:i + 32], "big"), pt)) for i in range(0, len(scal), 32)])

    def msm(scal, pts):
        acc = None
        for i in range(len(scal) // 32):
            acc = O.g1_add(acc, O.g1_mul(int.from_bytes(scal[32 * i:32 * i + 32], "big"), O.g1_unmarshal(pts[64 * i:64 * i + 64])))
        return np.frombuffer(O.g1_marshal(acc), dtype=np.uint8), np.zeros(len(scal) // 32, dtype=np.uint8)

    def batch_add(a, b):
        out = [O.g1_marshal(O.g1_add(O.g1_unmarshal(a[i:i + 64]), O.g1_unmarshal(b[i:i + 64]))) for i in range(0, len(a), 64)]
        return rows(out), np.zeros(len(out), dtype=np.uint8)

    monkeypatch.setattr(poly, "_ops", lambda group: (mul_same_base, msm, 64))
    monkeypatch.setattr(poly, "_add_op", lambda group: batch_add)
    g = bn256.NewSuite().G1()
    rng = random.Random(5)
    rand = lambda n: bytes(rng.randrange(256) for _ in range(n))
    t, n = 3, 6
    pri, q = poly.PriPoly.new(g, t, rand=rand), poly.PriPoly.new(g, t, rand=rand)
    pub = pri.Commit(None)
    shares = [pub.Eval(i) for i in range(n)]
    shares[1] = None
    rng.shuffle(shares)
    assert poly.recover_pub_poly(g, shares, t, n).Equal(pub)
    assert poly.recover_commit(g, shares, t, n).Equal(pub.Commit())
    assert pub.Add(q.Commit(None)).Equal(pri.Add(q).Commit(None))
    assert poly.recover_secret(g, pri.Shares(n)[1:], t, n).Equal(pri.coeffs[0])
    assert poly.recover_pri_poly(g, pri.Shares(n)[2:2 + t], t, n).Equal(pri)


def test_share_poly_recover_pub_poly_and_scalar_side(monkeypatch):
    """RecoverPubPoly as t MSMs over the same shares (poly.go:480-508), PubPoly.Add / Equal, and the scalar-field
    routines (RecoverSecret, RecoverPriPoly, PriPoly.Mul / Add, lagrangeBasis) of poly.go:96-283, 513-536."""
    import random

    import pytest

    g, poly, E = _oracle_backed_ed25519(monkeypatch)
    rng = random.Random(12)
    rand = lambda n: bytes(rng.randrange(256) for _ in range(n))
    t, n = 4, 7
    pri = poly.PriPoly.new(g, t, rand=rand)
    pub = pri.Commit(None)
    for c, A in zip(pri.coeffs, pub.commits):  # Commit: coeffs[i] * B
        assert A.MarshalBinary() == E.mul(c.MarshalBinary(), E.encode(E.B), vartime=True)
    # scalar side
    pshares = pri.Shares(n)
    assert [s.I for s in pshares] == list(range(n))
    some = list(pshares)
    some[0] = None
    rng.shuffle(some)
    assert poly.recover_secret(g, some, t, n).Equal(pri.coeffs[0])
    assert poly.recover_pri_poly(g, some, t, n).Equal(pri)
    with pytest.raises(ValueError):
        poly.recover_secret(g, some[:0] + [s for s in some if s is not None][:t - 1], t, n)
    xs = {i: g.Scalar().SetInt64(i + 1) for i in (1, 2, 5, 6)}
    for i in xs:  # L_i(x_m) = [i == m]
        L = poly.lagrange_basis(g, i, xs)
        assert L.Threshold() == len(xs)
        for m in xs:
            assert L.Eval(m).V.Equal(g.Scalar().One() if m == i else g.Scalar().Zero())
    q = poly.PriPoly.new(g, t, rand=rand)
    prod = pri.Mul(q)  # (p q)(x) = p(x) q(x), degree 2t - 2
    assert prod.Threshold() == 2 * t - 1
    assert prod.Eval(5).V.Equal(g.Scalar().Mul(pri.Eval(5).V, q.Eval(5).V))
    assert pri.Add(q).Eval(5).V.Equal(g.Scalar().Add(pri.Eval(5).V, q.Eval(5).V))
    with pytest.raises(ValueError):
        pri.Add(prod)
    # point side: any t public shares give back every commitment
    shares = [pub.Eval(i) for i in range(n)]
    for s, ps in zip(shares, pshares):  # Eval(i) = p(i) * B
        assert s.V.MarshalBinary() == E.mul(ps.V.MarshalBinary(), E.encode(E.B), vartime=True)
    shares[2] = None
    rng.shuffle(shares)
    rec = poly.recover_pub_poly(g, shares, t, n)
    assert rec.Equal(pub) and rec.Threshold() == t
    assert poly.recover_commit(g, shares, t, n).Equal(pub.Commit())
    with pytest.raises(ValueError):
        poly.recover_pub_poly(g, [s for s in shares if s is not None][:t - 1], t, n)
    # PubPoly.Add commutes with PriPoly.Add
    qpub = q.Commit(None)
    assert pub.Add(qpub).Equal(pri.Add(q).Commit(None))
    assert not pub.Equal(qpub)
    with pytest.raises(ValueError):
        pub.Add(poly.PubPoly(g, None, qpub.commits[:2]))


def test_pack_fixed_isolates_wrong_length_elements():
    """Packing of per-element encodings for the batch entry points: a wrong-length element becomes a row of zeros
    and is reported, every other element stays in its own row."""
    import numpy as np

    from kyber_amd.pairing._engine import pack_fixed

    items = [bytes([i + 1]) * 48 for i in range(5)]
    items[1] = b""
    items[2] = items[2][:-1]
    items[3] = items[3] + b"\x00"
    arr, bad = pack_fixed(items, 48)
    assert arr.shape == (5, 48) and bad == [1, 2, 3]
    assert bytes(arr[0]) == items[0] and bytes(arr[4]) == items[4] and not arr[1:4].any()
    arr, bad = pack_fixed([], 96)
    assert arr.shape == (0, 96) and bad == []


def test_single_element_policy_constants_match_the_go_suite_and_the_measurement():
    """SURVEY.md section 8b / VERDICT r2 item 6: one element stays on the CPU.  The thresholds the Python mirror documents are
    the Go suite's, and both are at or above every break-even batch size measured on the GPU
    (profiles/r03_single_call_latency.json, written by tools/latency_probe.py)."""
    import json
    import os
    import re

    from kyber_amd.pairing import _engine as E

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    go = open(os.path.join(root, "go", "kyberhip", "suite", "point.go")).read()
    assert int(re.search(r"MinDeviceBatch\s+= (\d+)", go).group(1)) == E.MIN_DEVICE_BATCH
    assert int(re.search(r"MinDevicePairings = (\d+)", go).group(1)) == E.MIN_DEVICE_PAIRINGS
    lat = json.load(open(os.path.join(root, "profiles", "r03_single_call_latency.json")))
    for name, be in lat["break_even_batch"].items():
        assert be is not None and be <= (E.MIN_DEVICE_PAIRINGS if "pair" in name else E.MIN_DEVICE_BATCH), (name, be)
    # a single call on the device is slower than the reference's one core for every operation measured: the policy is needed
    for name, v in lat["latency_us"].items():
        assert v[0] > lat["reference_single_core_us_per_op"][name], name
