"""Randomised GPU soak against the oracles, seed after seed (opt-in: KYB_SOAK_SEEDS=<count>, default 1 seed so that the
regular `-m gpu` run stays short): what the fixed-seed parity tests check, on fresh inputs every seed -- Ed25519 fixed /
variable base (both scalar semantics) and MSM element for element against the C oracle with edge scalars mixed in; G1 / G2
scalar multiplication, Pair bytes, ValidatePairing and the fused verification of the three pairing suites against the
Python oracles on a few elements per seed (an oracle pairing takes about a second)."""
import hashlib
import importlib
import os
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SEEDS = list(range(int(os.environ.get("KYB_SOAK_SEEDS", "1"))))


def _shake(label, n):
    return np.frombuffer(hashlib.shake_256(label).digest(n), dtype=np.uint8)


@pytest.mark.parametrize("seed", SEEDS)
def test_ed25519_soak(seed):
    import torch

    from kyber_amd.group import edwards25519 as ed
    from oracle import ed25519 as O
    from tests import _oracle_c as OC

    n = 4096
    rng = random.Random(1000 + seed)
    s = _shake(b"soak/ed/s/%d" % seed, n * 32).reshape(n, 32).copy()
    h = _shake(b"soak/ed/h/%d" % seed, n * 32).reshape(n, 32).copy()
    h[:, 31] &= 0x0F
    # edge scalars: 0, 1, l - 1, l, l + 1, 2^252, 2^255 - 1, 2^255, 2^256 - 1, and a few with long zero / one runs
    edge = [0, 1, O.L - 1, O.L, O.L + 1, 1 << 252, (1 << 255) - 1, 1 << 255, (1 << 256) - 1,
            ((1 << 256) - 1) ^ ((1 << 128) - 1), (1 << 128) - 1, 0x8888888888888888888888888888888888888888888888888888888888888888 >> 1]
    for j, e in enumerate(edge):
        s[rng.randrange(n) if j else 0] = np.frombuffer(e.to_bytes(32, "little"), dtype=np.uint8)
    d_s, d_h = torch.from_numpy(s).cuda(), torch.from_numpy(h).cuda()
    P = ed.batch_mul_base(d_h)
    assert (OC.ed_mul_base(h) == P.cpu().numpy()).all()
    # points outside the prime-order subgroup: the small-order points themselves and prime-order points shifted by
    # them, some sitting on the edge scalars' rows
    import json

    small = [bytes.fromhex(x) for x in json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ed25519_misc.json")))["small_order"]]
    Pn = P.cpu().numpy().copy()
    for j, t in enumerate(small):
        Pn[rng.randrange(n)] = np.frombuffer(t, dtype=np.uint8)
        r = rng.randrange(n) if j else 0
        Pn[r] = np.frombuffer(O.encode(O.add(O.decode(bytes(Pn[r])), O.decode(t))), dtype=np.uint8)
    P = torch.from_numpy(Pn).cuda()
    Pb = ed.batch_mul_base(d_s)  # the edge scalars through the fixed-base kernel too
    assert (OC.ed_mul_base(s) == Pb.cpu().numpy()).all()
    for vt in (False, True):
        A, st = ed.batch_mul(d_s, P, vartime=vt)
        exp, est = OC.ed_mul(s, P.cpu().numpy(), vartime=vt)
        assert (st.cpu().numpy() == est).all() and (A.cpu().numpy() == exp).all(), ("vartime", vt)
    m, _ = ed.msm(d_s, P)
    exp, rc = OC.ed_msm(s, P.cpu().numpy())
    assert rc == 0 and bytes(m.cpu().numpy()) == bytes(exp)


@pytest.mark.parametrize("seed", SEEDS)
@pytest.mark.parametrize("name", ["bls12381", "bn256", "bn254"])
def test_pairing_suite_soak(name, seed):
    m = importlib.import_module("kyber_amd.pairing." + name)
    O = importlib.import_module("oracle." + name)
    rng = random.Random(2000 + seed)
    order = O.R if name == "bls12381" else O.ORDER
    n = 3
    ks = [rng.randrange(1, order) for _ in range(n)]
    hs = [rng.randrange(1, order) for _ in range(n)]
    g1, g2 = O.G1_GEN, O.G2_GEN
    if name == "bls12381":
        e1, e2 = O.g1_compress, O.g2_compress
        gtb = lambda a: O.gt_to_bytes(a)
    else:
        e1, e2 = O.g1_marshal, O.g2_marshal
        gtb = lambda a: O.gt_marshal(a)
    Ps = [O.g1_mul(h, g1) for h in hs]
    Qs = [O.g2_mul(k, g2) for k in ks]
    P = np.stack([np.frombuffer(e1(p), dtype=np.uint8) for p in Ps])
    Q = np.stack([np.frombuffer(e2(q), dtype=np.uint8) for q in Qs])
    kb = np.stack([np.frombuffer(k.to_bytes(32, "big"), dtype=np.uint8) for k in ks])
    # scalar multiplications
    kP, st = m.g1_batch_mul(kb, P)
    assert not np.asarray(st).any()
    assert [bytes(r) for r in kP] == [e1(O.g1_mul(k, p)) for k, p in zip(ks, Ps)]
    kQ, st = m.g2_batch_mul(kb, Q)
    assert not np.asarray(st).any()
    assert [bytes(r) for r in kQ] == [e2(O.g2_mul(k, q)) for k, q in zip(ks, Qs)]
    # pairings, byte for byte
    gt, st = m.batch_pair(P, Q)
    assert not np.asarray(st).any()
    assert [bytes(r) for r in gt] == [gtb(O.pair(p, q)) for p, q in zip(Ps, Qs)]
    # e(kP, Q) == e(P, kQ); a forged entry fails
    ok, st = m.batch_validate_pairing(kP, Q, P, kQ)
    assert not np.asarray(st).any() and np.asarray(ok).astype(bool).all()
    forged = np.array(kP).copy()
    forged[1] = P[1]
    ok, st = m.batch_validate_pairing(forged, Q, P, kQ)
    assert list(np.asarray(ok).astype(bool)) == [True, False, True]


@pytest.mark.parametrize("seed", SEEDS)
def test_bls12381_fused_verification_soak(seed):
    """sign/bls Verify on BLS12-381: hash to G1 by the oracle's RFC 9380 restatement, signature = [x] H(m) by the oracle;
    the engine's fused verification accepts them and rejects a swapped pair"""
    from kyber_amd.pairing import bls12381 as m
    from oracle import bls12381 as O

    rng = random.Random(3000 + seed)
    n = 4
    xs = [rng.randrange(1, O.R) for _ in range(n)]
    msgs = np.stack([_shake(b"soak/fv/%d/%d" % (seed, i), 32) for i in range(n)])
    Hm, st = m.batch_hash_g1(msgs)
    assert not np.asarray(st).any()
    hpts = [O.hash_to_g1(bytes(mm), m.DOMAIN_G1) for mm in msgs]
    assert [bytes(r) for r in Hm] == [O.g1_compress(h) for h in hpts]
    X = np.stack([np.frombuffer(O.g2_compress(O.g2_mul(x, O.G2_GEN)), dtype=np.uint8) for x in xs])
    xb = np.stack([np.frombuffer(x.to_bytes(32, "big"), dtype=np.uint8) for x in xs])
    sig, _ = m.g1_batch_mul(xb, Hm)
    assert [bytes(r) for r in sig] == [O.g1_compress(O.g1_mul(x, h)) for x, h in zip(xs, hpts)]
    ok, st = m.batch_verify_g1(X, msgs, sig)
    assert not np.asarray(st).any() and np.asarray(ok).astype(bool).all()
    swapped = np.array(sig).copy()
    swapped[[0, 1]] = swapped[[1, 0]]
    ok, st = m.batch_verify_g1(X, msgs, swapped)
    assert list(np.asarray(ok).astype(bool)) == [False, False, True, True]


@pytest.mark.parametrize("seed", SEEDS)
@pytest.mark.parametrize("name", ["bls12381", "bn256", "bn254"])
def test_pairing_suite_msm_and_hash_soak(name, seed):
    """G1 / G2 MSM with scalars on both sides of the group order (mod.Int encodings are plain integers: 0, 1, n - 1, n,
    n + 1, 2^255, 2^256 - 1 mixed into random ones), short-scalar MSM, and hash-to-G1 of random messages, against the
    oracle's sums"""
    m = importlib.import_module("kyber_amd.pairing." + name)
    O = importlib.import_module("oracle." + name)
    rng = random.Random(4000 + seed)
    order = O.R if name == "bls12381" else O.ORDER
    e1, e2 = (O.g1_compress, O.g2_compress) if name == "bls12381" else (O.g1_marshal, O.g2_marshal)
    n = 24
    ks = [rng.getrandbits(256) for _ in range(n)]
    for j, e in enumerate([0, 1, order - 1, order, order + 1, 1 << 255, (1 << 256) - 1, (1 << 128) - 1]):
        ks[2 * j] = e
    kb = np.stack([np.frombuffer(k.to_bytes(32, "big"), dtype=np.uint8) for k in ks])
    Ps = [O.g1_mul(rng.randrange(1, order), O.G1_GEN) for _ in range(n)]
    Qs = [O.g2_mul(rng.randrange(1, order), O.G2_GEN) for _ in range(8)]
    P = np.stack([np.frombuffer(e1(p), dtype=np.uint8) for p in Ps])
    Q = np.stack([np.frombuffer(e2(q), dtype=np.uint8) for q in Qs])
    acc = None
    for k, p in zip(ks, Ps):
        acc = O.g1_add(acc, O.g1_mul(k % order, p))
    out, st = m.g1_msm(kb, P)
    assert not np.asarray(st).any() and bytes(out) == e1(acc)
    acc = None
    for k, q in zip(ks[:8], Qs):
        acc = O.g2_add(acc, O.g2_mul(k % order, q))
    out, st = m.g2_msm(kb[:8], Q)
    assert not np.asarray(st).any() and bytes(out) == e2(acc)
    # element-wise multiplication by the same edge scalars
    kP, st = m.g1_batch_mul(kb, P)
    assert not np.asarray(st).any()
    assert [bytes(r) for r in kP] == [e1(O.g1_mul(k % order, p)) for k, p in zip(ks, Ps)]
    # hash to G1 (sign/bls Hash): the suite's own map, random message bytes
    msgs = np.stack([_shake(b"soak/hash/%s/%d/%d" % (name.encode(), seed, i), 32) for i in range(6)])
    Hm, st = m.batch_hash_g1(msgs)
    assert not np.asarray(st).any()
    if name == "bls12381":
        exp = [e1(O.hash_to_g1(bytes(x), m.DOMAIN_G1)) for x in msgs]
    else:
        exp = [e1(O.hash_to_g1(bytes(x))) for x in msgs]
    assert [bytes(r) for r in Hm] == exp


@pytest.mark.parametrize("seed", SEEDS)
def test_ed25519_unmarshal_and_hash_soak(seed):
    """random 32-byte strings through the batch UnmarshalBinary (about half decode) and random messages through Hash,
    against the oracle"""
    from kyber_amd.group import edwards25519 as ed
    from oracle import ed25519 as O

    n = 512
    raw = _shake(b"soak/ed/unmarshal/%d" % seed, n * 32).reshape(n, 32).copy()
    out, st = ed.batch_unmarshal(raw)
    out, st = np.asarray(out), np.asarray(st)
    for i in range(n):
        pt = O.decode(bytes(raw[i]))
        assert (st[i] == 0) == (pt is not None), i
        if pt is not None:
            assert bytes(out[i]) == O.encode(pt), i
    assert 100 < int((st == 0).sum()) < 400
    dst = b"soak-dst-%d" % seed
    for ln in (1, 31, 64, 129):  # equal-length batches (the call's contract), lengths around the SHA-512 block
        msgs = [bytes(_shake(b"soak/ed/hash/%d/%d/%d" % (seed, ln, i), ln)) for i in range(4)]
        Hm = ed.batch_hash(msgs, dst)
        assert [bytes(r) for r in np.asarray(Hm)] == [O.hash_to_curve(x, dst) for x in msgs], ln


@pytest.mark.parametrize("seed", SEEDS)
@pytest.mark.parametrize("name", ["bls12381", "bn256", "bn254"])
def test_pairing_suite_unmarshal_mutation_soak(name, seed):
    """valid G1 / G2 encodings with one random bit flipped (most land off the curve, some on the curve outside the
    subgroup, a few on another valid point), plus random strings: the batch UnmarshalBinary accepts exactly what the
    oracle's decoder accepts and re-encodes what it accepts canonically"""
    m = importlib.import_module("kyber_amd.pairing." + name)
    O = importlib.import_module("oracle." + name)
    rng = random.Random(5000 + seed)
    order = O.R if name == "bls12381" else O.ORDER
    if name == "bls12381":
        enc = {1: O.g1_compress, 2: O.g2_compress}
        dec = {1: O.g1_decompress, 2: O.g2_decompress}
    else:
        enc = {1: O.g1_marshal, 2: O.g2_marshal}
        dec = {1: O.g1_unmarshal, 2: O.g2_unmarshal}
    for group, gen, mul, cnt in ((1, O.G1_GEN, O.g1_mul, 48), (2, O.G2_GEN, O.g2_mul, 16)):
        rows = []
        for i in range(cnt):
            b = bytearray(enc[group](mul(rng.randrange(1, order), gen)))
            if i % 8 != 7:  # every eighth stays valid
                bit = rng.randrange(8 * len(b))
                b[bit >> 3] ^= 1 << (bit & 7)
            rows.append(bytes(b))
        rows += [bytes(rng.getrandbits(8) for _ in range(len(rows[0]))) for _ in range(8)]
        out, st = getattr(m, "g%d_batch_unmarshal" % group)(np.stack([np.frombuffer(r, dtype=np.uint8) for r in rows]))
        out, st = np.asarray(out), np.asarray(st)
        accepted = 0
        for i, r in enumerate(rows):
            try:
                pt, ok = dec[group](r), True
            except Exception:
                pt, ok = None, False
            assert (st[i] == 0) == ok, (name, group, i, int(st[i]))
            if ok:
                accepted += 1
                assert bytes(out[i]) == enc[group](pt), (name, group, i)
        assert accepted >= cnt // 8


@pytest.mark.parametrize("name", ["bls12381", "bn256", "bn254"])
def test_pairing_batch_shapes(name):
    """the tower machine works on 64 pairings per workgroup: batches of 0, 1, 2, 63, 64, 65 and 129 pairings give, row
    for row, the bytes of the 129-row batch (partial workgroups, a workgroup with one live lane, grid-stride reuse)"""
    m = importlib.import_module("kyber_amd.pairing." + name)
    n = 129
    raw = _shake(b"shapes/" + name.encode(), 2 * n * 32).reshape(2, n, 32).copy()
    raw[:, :, 0] &= 0x3F
    g1b = np.frombuffer(m.G1_BASE, dtype=np.uint8)
    g2b = np.frombuffer(m.G2_BASE, dtype=np.uint8)
    P, _ = m._mul(1, raw[0], g1b, True)
    Q, _ = m._mul(2, raw[1], g2b, True)
    P, Q = np.asarray(P), np.asarray(Q)
    gt, st = m.batch_pair(P, Q)
    gt = np.asarray(gt)
    assert not np.asarray(st).any()
    kP, _ = m.g1_batch_mul(raw[1], P)
    kQ, _ = m.g2_batch_mul(raw[1], Q)
    kP, kQ = np.array(kP), np.asarray(kQ)
    kP[5] = P[5]  # one false row
    ok, st = m.batch_validate_pairing(kP, Q, P, kQ)
    ok = np.asarray(ok).astype(bool)
    assert not ok[5] and ok.sum() == n - 1
    for k in (0, 1, 2, 63, 64, 65):
        g, s = m.batch_pair(P[:k], Q[:k])
        assert np.asarray(g).shape == (k, gt.shape[1]) and (np.asarray(g) == gt[:k]).all(), k
        g, s = m.batch_pair(P[n - k:], Q[n - k:])
        assert (np.asarray(g) == gt[n - k:]).all(), k
        o, s = m.batch_validate_pairing(kP[:k], Q[:k], P[:k], kQ[:k])
        assert (np.asarray(o).astype(bool) == ok[:k]).all(), k


def test_concurrent_host_threads_and_streams():
    """four host threads hammer different entry points at once -- host-buffer calls (staging pool, page-locked slots,
    per-device mutex) and device-tensor calls on their own streams (per-(kind, stream) workspaces) -- and every call
    returns what the same call returned alone"""
    import threading

    import torch

    from kyber_amd.group import edwards25519 as ed
    from kyber_amd.pairing import bls12381 as bls, bn256 as bn

    n = 20000
    s = _shake(b"conc/s", n * 32).reshape(n, 32).copy()
    h = _shake(b"conc/h", n * 32).reshape(n, 32).copy()
    h[:, 31] &= 0x0F
    P = np.asarray(ed.batch_mul_base(h))
    ref_mul = np.asarray(ed.batch_mul(s, P)[0])
    ref_msm = bytes(np.asarray(ed.msm(s[:5000], P[:5000])[0]))
    m = 256
    raw = _shake(b"conc/p", 4 * m * 32).reshape(4, m, 32).copy()
    raw[:, :, 0] &= 0x3F
    bP, _ = bls._mul(1, raw[0], np.frombuffer(bls.G1_BASE, dtype=np.uint8), True)
    bQ, _ = bls._mul(2, raw[1], np.frombuffer(bls.G2_BASE, dtype=np.uint8), True)
    nP, _ = bn._mul(1, raw[2], np.frombuffer(bn.G1_BASE, dtype=np.uint8), True)
    nQ, _ = bn._mul(2, raw[3], np.frombuffer(bn.G2_BASE, dtype=np.uint8), True)
    bP, bQ, nP, nQ = (np.asarray(x) for x in (bP, bQ, nP, nQ))
    ref_bls = np.asarray(bls.batch_pair(bP, bQ)[0])
    ref_bn = np.asarray(bn.batch_pair(nP, nQ)[0])
    errors = []

    def run(fn, check, reps):
        try:
            for _ in range(reps):
                if not check(fn()):
                    errors.append(fn.__name__ if hasattr(fn, "__name__") else "mismatch")
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    def dev_mul():
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            out, _ = ed.batch_mul(torch.from_numpy(s).cuda(), torch.from_numpy(P).cuda())
            st.synchronize()
            return out.cpu().numpy()

    def dev_pair():
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            out, _ = bls.batch_pair(torch.from_numpy(bP).cuda(), torch.from_numpy(bQ).cuda())
            st.synchronize()
            return out.cpu().numpy()

    jobs = [(lambda: np.asarray(ed.batch_mul(s, P)[0]), lambda r: (r == ref_mul).all(), 4),
            (lambda: bytes(np.asarray(ed.msm(s[:5000], P[:5000])[0])), lambda r: r == ref_msm, 6),
            (lambda: np.asarray(bls.batch_pair(bP, bQ)[0]), lambda r: (r == ref_bls).all(), 6),
            (lambda: np.asarray(bn.batch_pair(nP, nQ)[0]), lambda r: (r == ref_bn).all(), 6),
            (dev_mul, lambda r: (r == ref_mul).all(), 4),
            (dev_pair, lambda r: (r == ref_bls).all(), 6)]
    # daemon threads and a bounded join: a deadlock fails the test instead of hanging the run (and the interpreter's exit)
    threads = [threading.Thread(target=run, args=j, daemon=True) for j in jobs]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert not any(t.is_alive() for t in threads), "host threads are stuck: lock order?"
    assert not errors, errors


def test_concurrent_round4_entry_points():
    """the per-stream caches of round 4 under contention: two signers' same-key verifications on two device streams
    (a line table per (stream, key)), same-base commitments over two different bases (a window table and its hint per
    (group, stream)), bn256 ValidatePairing (product program + fallback mask in the call's workspace) and the
    scalar-field Horner kernel from host threads -- every call returns what it returned alone"""
    import threading

    import torch

    from kyber_amd.pairing import bls12381 as bls, bn256 as bn

    n = 4096
    msgs = _shake(b"r4/m", n * 32).reshape(n, 32).copy()
    keys = [_shake(b"r4/k%d" % j, 32).reshape(1, 32).copy() for j in range(2)]
    for k in keys:
        k[0, 0] &= 0x3F
    X = [np.asarray(bls._mul(2, k, np.frombuffer(bls.G2_BASE, dtype=np.uint8), True)[0])[0] for k in keys]
    H, _ = bls.batch_hash_g1(msgs)
    sig = [np.asarray(bls.g1_batch_mul(np.repeat(k, n, axis=0), np.asarray(H), bls.F_TRUSTED(0))[0]) for k in keys]
    sig[0][7] = sig[1][7]  # one forged signature under the first key
    ref_v = [np.asarray(bls.batch_verify_g1_same_key(bytes(X[j]), msgs, sig[j])[0]).copy() for j in range(2)]
    assert ref_v[1].all() and ref_v[0].sum() == n - 1
    nc = 1 << 17
    sc = _shake(b"r4/c", nc * 32).reshape(nc, 32).copy()
    bases = [np.asarray(bls._mul(1, keys[j], np.frombuffer(bls.G1_BASE, dtype=np.uint8), True)[0])[0] for j in range(2)]
    ref_c = [np.asarray(bls.g1_commit(sc, bases[j])[0]).copy() for j in range(2)]
    m = 512
    raw = _shake(b"r4/p", 2 * m * 32).reshape(2, m, 32).copy()
    raw[:, :, 0] &= 0x3F
    nP = np.asarray(bn._mul(1, raw[0], np.frombuffer(bn.G1_BASE, dtype=np.uint8), True)[0])
    nQ = np.asarray(bn._mul(2, raw[1], np.frombuffer(bn.G2_BASE, dtype=np.uint8), True)[0])
    nP2 = nP.copy()
    nP2[5] = nP[6]
    ref_chk = np.asarray(bn.batch_validate_pairing(nP, nQ, nP2, nQ)[0]).copy()
    assert ref_chk.sum() == m - 1
    coeffs = _shake(b"r4/poly", 67 * 32).reshape(67, 32).copy()
    coeffs[:, 0] &= 0x3F
    idx = np.arange(5000, dtype=np.uint32)
    ref_poly = bls.ENGINE.scalar_poly_eval(coeffs, idx).copy()
    errors = []

    def run(fn, check, reps):
        try:
            for _ in range(reps):
                if not check(fn()):
                    errors.append("mismatch")
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    def dev_verify(j):
        def f():
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                ok, _ = bls.batch_verify_g1_same_key(torch.from_numpy(X[j]).cuda(), torch.from_numpy(msgs).cuda(), torch.from_numpy(sig[j]).cuda())
                st.synchronize()
                return ok.cpu().numpy()
        return f

    def dev_commit(j):
        def f():
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                out, _ = bls.g1_commit(torch.from_numpy(sc).cuda(), torch.from_numpy(bases[j]).cuda())
                st.synchronize()
                return out.cpu().numpy()
        return f

    jobs = [(dev_verify(0), lambda r: (r == ref_v[0]).all(), 5), (dev_verify(1), lambda r: (r == ref_v[1]).all(), 5),
            (dev_commit(0), lambda r: (r == ref_c[0]).all(), 4), (dev_commit(1), lambda r: (r == ref_c[1]).all(), 4),
            (lambda: np.asarray(bls.g1_commit(sc, bases[1])[0]), lambda r: (r == ref_c[1]).all(), 3),
            (lambda: np.asarray(bn.batch_validate_pairing(nP, nQ, nP2, nQ)[0]), lambda r: (r == ref_chk).all(), 6),
            (lambda: bls.ENGINE.scalar_poly_eval(coeffs, idx), lambda r: (r == ref_poly).all(), 6)]
    threads = [threading.Thread(target=run, args=j, daemon=True) for j in jobs]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=180)
    assert not any(t.is_alive() for t in threads), "host threads are stuck: lock order?"
    assert not errors, errors


@pytest.mark.parametrize("seed", SEEDS)
@pytest.mark.parametrize("name", ["bls12381", "bn256", "bn254"])
def test_fixed_base_policies_soak(name, seed):
    """Round 4: same-base batches through the fixed-base table under the group's policy (BLS12-381: sub-scalars over
    endomorphism images) -- a fresh random base per seed on both groups, scalars built to stress the splits (multiples of
    z^2 / |z| and their neighbours, zero sub-scalars, digit carries, values at and above the order), sampled lanes against
    the oracle's own multiplication; a base outside the subgroup is rejected where UnmarshalBinary rejects it."""
    import torch

    m = importlib.import_module("kyber_amd.pairing." + name)
    O = importlib.import_module("oracle." + name)
    rng = random.Random(7000 + seed)
    n = 1 << 17
    z = getattr(O, "X_ABS", 0x44E992B44A6909F1)
    edge = [0, 1, z, z - 1, z * z, z * z + 1, z * z - 1, z**3, z**3 - 1, 512 * z * z + 512, 3 * z**3 + 2 * z * z + z, (1 << 256) - 1,
            m.ORDER, m.ORDER - 1, m.ORDER + 1, ((1 << 256) // (z * z)) * z * z, 1023 << 120, (511 << 246) | 512]
    for grp in (1, 2):
        omul, gen = (O.g1_mul, O.G1_GEN) if grp == 1 else (O.g2_mul, O.G2_GEN)
        enc = (getattr(O, "g1_compress", None) or O.g1_marshal) if grp == 1 else (getattr(O, "g2_compress", None) or O.g2_marshal)
        bp = omul(rng.randrange(1, m.ORDER), gen)
        base = torch.from_numpy(np.frombuffer(enc(bp), dtype=np.uint8).copy()).cuda()
        ks = edge + [rng.randrange(1 << 256) for _ in range(24)]
        s = torch.from_numpy(_shake(b"soak/fb/%s/%d/%d" % (name.encode(), grp, seed), n * 32).reshape(n, 32).copy()).cuda()
        where = [rng.randrange(n) for _ in ks]
        where[0], where[1] = 0, n - 1
        for w, k in zip(where, ks):
            s[w] = torch.from_numpy(np.frombuffer(k.to_bytes(32, "big"), dtype=np.uint8).copy()).cuda()
        out, st = (m.g1_commit if grp == 1 else m.g2_commit)(s, base)
        torch.cuda.synchronize()
        assert not st.any().item()
        got, sc = out[where].cpu().numpy(), s[where].cpu().numpy()
        for j in range(len(ks)):
            k = int.from_bytes(bytes(sc[j]), "big")  # (a later edge scalar may have overwritten an earlier one's lane)
            assert bytes(got[j]) == enc(omul(k % m.ORDER, bp)), (name, grp, hex(k))


@pytest.mark.parametrize("seed", SEEDS)
def test_same_key_verification_soak(seed):
    """Round 4: program VERIFYK on a fresh key, fresh messages and oracle-made signatures per seed, forged ones mixed in;
    the general fused verification must agree element for element."""
    from kyber_amd.pairing import bls12381 as bls
    from oracle import bls12381 as O

    rng = random.Random(8000 + seed)
    x = rng.randrange(1, O.R)
    X = O.g2_compress(O.g2_mul(x, O.G2_GEN))
    n = 40
    msgs = [bytes(rng.randrange(256) for _ in range(32)) for _ in range(n)]
    sigs = []
    for i, mm in enumerate(msgs):
        good = i % 3 != 1
        if i < 6:  # a handful straight from the oracle (hash-to-curve + multiplication), the rest through the engine
            sigs.append(O.g1_compress(O.g1_mul(x if good else x + 1, O.hash_to_g1(mm, bls.DOMAIN_G1))))
        else:
            h = bls.batch_hash_g1([mm])[0]
            sigs.append(bytes(np.asarray(bls.g1_batch_mul(((x if good else x + 1) % O.R).to_bytes(32, "big"), h)[0])[0]))
    ok, st = bls.batch_verify_g1_same_key(X, msgs, sigs)
    ok_g, st_g = bls.batch_verify_g1([X] * n, msgs, sigs)
    assert not np.asarray(st).any() and list(ok) == list(ok_g) == [1 if i % 3 != 1 else 0 for i in range(n)]
