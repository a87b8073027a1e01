"""GPU parity tests for the BLS12-381 hot path, through the C ABI, bit-exact against the oracle and
the reference's ZCash fixtures; size-independent properties at larger batch sizes."""
import hashlib
import json
import os
import random

import numpy as np
import pytest

from oracle import bls12381 as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def bls():
    import torch

    assert torch.cuda.is_available()
    from kyber_amd.pairing import bls12381 as bls

    return bls


def _scalars(label: bytes, n: int) -> np.ndarray:
    raw = hashlib.shake_256(label).digest(n * 64)
    out = np.empty((n, 32), dtype=np.uint8)
    for i in range(n):
        out[i] = np.frombuffer((int.from_bytes(raw[64 * i:64 * i + 64], "big") % O.R).to_bytes(32, "big"), dtype=np.uint8)
    return out


def test_zcash_fixtures_through_the_engine(bls, golden_dir):
    d = json.load(open(os.path.join(golden_dir, "bls12381_zcash.json")))
    one = (1).to_bytes(32, "big")
    for grp, fn, size in (("G1", bls.g1_batch_mul, 48), ("G2", bls.g2_batch_mul, 96)):
        for e in d[grp]:
            buf = bytes.fromhex(e["hex"])
            if len(buf) != size:
                with pytest.raises(ValueError):
                    (bls.G1Elt() if grp == "G1" else bls.G2Elt()).UnmarshalBinary(buf)
                continue
            out, st = fn(one, buf)
            assert (st[0] == 0) == e["valid"], (grp, e["name"], st[0])
            if e["valid"]:
                assert bytes(out[0]) == buf
            else:
                assert not out.any()


def test_g1_g2_mul_vs_oracle(bls):
    rng = random.Random(4)
    edge = [0, 1, 2, 15, 16, 17, O.R - 1, O.R, O.R + 5, (1 << 256) - 1, 8 << 252]
    ks = edge + [rng.randrange(O.R) for _ in range(21)]
    hs = [rng.randrange(1, O.R) for _ in ks]
    p1 = [O.g1_compress(O.g1_mul(h, O.G1_GEN)) for h in hs]
    p2 = [O.g2_compress(O.g2_mul(h, O.G2_GEN)) for h in hs]
    p1[3], p2[3] = O.g1_compress(None), O.g2_compress(None)
    kb = [k.to_bytes(32, "big") for k in ks]
    out, st = bls.g1_batch_mul(b"".join(kb), b"".join(p1))
    assert not st.any()
    for i in range(len(ks)):
        assert bytes(out[i]) == O.g1_mul_bytes(kb[i], p1[i]), i
    out, st = bls.g2_batch_mul(b"".join(kb), b"".join(p2))
    assert not st.any()
    for i in range(len(ks)):
        assert bytes(out[i]) == O.g2_mul_bytes(kb[i], p2[i]), i
    # Commit (same base) == per-element mul
    out2, st2 = bls.g1_commit(b"".join(kb), p1[0])
    for i in range(0, len(ks), 5):
        assert bytes(out2[i]) == O.g1_mul_bytes(kb[i], p1[0])


def test_bad_inputs_status_and_zero_output(bls):
    k = (7).to_bytes(32, "big")
    good = O.g1_compress(O.G1_GEN)
    pts = good + bytes(48) + good
    out, st = bls.g1_batch_mul(k * 3, pts)
    assert list(st) == [0, 1, 0] and not out[1].any() and bytes(out[0]) == bytes(out[2])


def test_pairing_vs_oracle(bls):
    rng = random.Random(5)
    n = 6
    g1 = [O.g1_compress(O.g1_mul(rng.randrange(1, O.R), O.G1_GEN)) for _ in range(n)]
    g2 = [O.g2_compress(O.g2_mul(rng.randrange(1, O.R), O.G2_GEN)) for _ in range(n)]
    g1[4] = O.g1_compress(None)
    gt, st = bls.batch_pair(b"".join(g1), b"".join(g2))
    assert not st.any()
    for i in range(n):
        assert bytes(gt[i]) == O.pair_bytes(g1[i], g2[i]), i


def test_pairing_bilinear_at_scale(bls):
    """e(a P, b Q) == e(ab P, Q) == e(P, ab Q) for 2048 independent (a, b): bls12381_test.go:448-474."""
    n = 2048
    a, b = _scalars(b"t/a", n), _scalars(b"t/b", n)
    ab = np.empty_like(a)
    for i in range(n):
        x = int.from_bytes(bytes(a[i]), "big") * int.from_bytes(bytes(b[i]), "big") % O.R
        ab[i] = np.frombuffer(x.to_bytes(32, "big"), dtype=np.uint8)
    aP, st = bls.g1_commit(a)
    abP, _ = bls.g1_commit(ab)
    bQ, _ = bls.g2_commit(b)
    abQ, _ = bls.g2_commit(ab)
    assert not st.any()
    G2 = np.tile(np.frombuffer(bls.G2_BASE, dtype=np.uint8), (n, 1))
    G1 = np.tile(np.frombuffer(bls.G1_BASE, dtype=np.uint8), (n, 1))
    e1, s1 = bls.batch_pair(aP, bQ)
    e2, s2 = bls.batch_pair(abP, G2)
    e3, s3 = bls.batch_pair(G1, abQ)
    assert not (s1.any() or s2.any() or s3.any())
    assert (e1 == e2).all() and (e1 == e3).all()
    assert len({bytes(r) for r in e1[:64]}) == 64  # and they are not all the same value


def test_validate_pairing_truth_table_at_scale(bls):
    """bls.Verify shape (sign/bls/bls.go:36-38): e(H, X) == e(sig, G2), with a sprinkle of forgeries."""
    n = 1024
    x, h = _scalars(b"t/x", n), _scalars(b"t/h", n)
    xh = np.empty_like(x)
    for i in range(n):
        v = int.from_bytes(bytes(x[i]), "big") * int.from_bytes(bytes(h[i]), "big") % O.R
        xh[i] = np.frombuffer(v.to_bytes(32, "big"), dtype=np.uint8)
    Hm, _ = bls.g1_commit(h)
    X, _ = bls.g2_commit(x)
    sig, _ = bls.g1_commit(xh)
    G2 = np.tile(np.frombuffer(bls.G2_BASE, dtype=np.uint8), (n, 1))
    bad = np.zeros(n, dtype=bool)
    bad[::7] = True
    sig = sig.copy()
    sig[bad] = Hm[bad]  # wrong signature (valid point, wrong value)
    ok, st = bls.batch_validate_pairing(Hm, X, sig, G2)
    assert not st.any()
    assert (ok.astype(bool) == ~bad).all()


def test_device_resident_path_matches_host_path(bls):
    import torch

    n = 256
    k = _scalars(b"t/k", n)
    P, _ = bls.g1_commit(_scalars(b"t/p", n))
    out_h, st_h = bls.g1_batch_mul(k, P)
    out_d, st_d = bls.g1_batch_mul(torch.from_numpy(k).cuda(), torch.from_numpy(P).cuda())
    torch.cuda.synchronize()
    assert (out_d.cpu().numpy() == out_h).all() and not st_d.any().item()


def test_suite_mirror(bls):
    s = bls.NewSuite()
    a, b = s.G1().Scalar().SetInt64(5), s.G1().Scalar().SetInt64(7)
    P = s.G1().Point().Mul(a, None)
    Q = s.G2().Point().Mul(b, None)
    ab = s.G1().Scalar().Mul(a, b)
    assert s.Pair(P, Q).Equal(s.Pair(s.G1().Point().Mul(ab, None), s.G2().Point().Base()))
    assert s.ValidatePairing(P, Q, s.G1().Point().Mul(ab, None), s.G2().Point().Base())
    with pytest.raises(TypeError):
        P.Equal(Q)


def test_gt_mul_vs_oracle_and_homomorphism(bls):
    rng = random.Random(8)
    g1 = O.g1_compress(O.g1_mul(rng.randrange(1, O.R), O.G1_GEN))
    g2 = O.g2_compress(O.g2_mul(rng.randrange(1, O.R), O.G2_GEN))
    gt, _ = bls.batch_pair(g1, g2)
    ks = [0, 1, O.R - 1, rng.randrange(O.R)]
    out, st = bls.gt_batch_mul(b"".join(k.to_bytes(32, "big") for k in ks), np.tile(gt, (4, 1)))
    assert not st.any()
    for i, k in enumerate(ks):
        assert bytes(out[i]) == O.gt_mul_bytes(k.to_bytes(32, "big"), bytes(gt[0])), i
    m = 512
    k = _scalars(b"bls/gt/k", m)
    P, _ = bls.g1_commit(_scalars(b"bls/gt/p", m))
    G2 = np.tile(np.frombuffer(bls.G2_BASE, dtype=np.uint8), (m, 1))
    e, _ = bls.batch_pair(P, G2)
    kP, _ = bls.g1_batch_mul(k, P)
    ek, _ = bls.batch_pair(kP, G2)
    out, st = bls.gt_batch_mul(k, e)
    assert not st.any() and (out == ek).all()
    # rejected inputs: coefficient >= p, and an element outside the order-r subgroup
    bad = np.stack([np.full(576, 0xFF, dtype=np.uint8), np.frombuffer(bytes(575) + b"\x02", dtype=np.uint8)])
    out, st = bls.gt_batch_mul(k[:2], bad)
    assert list(st) == [1, 2] and not out.any()


def test_drand_fixtures_through_the_engine(bls, golden_dir):
    """The reference's BLS12-381 signature fixtures replayed on the GPU: hash-to-curve + decompression + pairing
    check (kilic/suite_test.go:17-72,84-106; gnark/suite_test.go:16-40; bls12381_test.go:877-904)."""
    import struct

    from kyber_amd.sign import bls as sbls

    D = json.load(open(os.path.join(golden_dir, "bls12381_drand.json")))
    f = D["sig_on_g1"]
    msg = hashlib.sha256(struct.pack(">Q", f["round"])).digest()
    pk, sig = bytes.fromhex(f["pk_g2"]), bytes.fromhex(f["sig_g1"])
    assert not sbls.NewSchemeOnG1_bls12381().verify(pk, msg, sig)  # default G1 domain: must fail
    assert sbls.NewSchemeOnG1_bls12381(D["dst_g2"].encode()).verify(pk, msg, sig)  # G2 domain on G1: passes
    f = D["sig_on_g2"]
    msg = hashlib.sha256(bytes.fromhex(f["prev_sig"]) + struct.pack(">Q", f["round"])).digest()
    assert sbls.NewSchemeOnG2_bls12381().verify(bytes.fromhex(f["pk_g1"]), msg, bytes.fromhex(f["sig_g2"]))
    f = D["edge_case"]
    assert sbls.NewSchemeOnG1_bls12381().verify(bytes.fromhex(f["pk_g2"]), bytes.fromhex(f["msg"]), bytes.fromhex(f["sig_g1"]))


def test_hash_to_curve_vs_oracle_and_sign_verify(bls):
    from kyber_amd.sign import bls as sbls

    msgs = [hashlib.sha256(b"m%d" % i).digest() for i in range(64)]
    h1, st = bls.batch_hash_g1(msgs)
    assert not st.any()
    for i in range(0, 64, 9):
        assert bytes(h1[i]) == O.g1_compress(O.hash_to_g1(msgs[i], bls.DOMAIN_G1))
    h2, st = bls.batch_hash_g2(msgs[:8])
    assert not st.any()
    for i in (0, 7):
        assert bytes(h2[i]) == O.g2_compress(O.hash_to_g2(msgs[i], bls.DOMAIN_G2))
    for ln in (0, 1, 55, 56, 64, 100):
        m = bytes((5 * i + ln) & 0xFF for i in range(ln))
        out, st = bls.batch_hash_g1([m], b"QUUX-V01-CS02-with-BLS12381G1_XMD:SHA-256_SSWU_RO_")
        assert bytes(out[0]) == O.g1_compress(O.hash_to_g1(m, b"QUUX-V01-CS02-with-BLS12381G1_XMD:SHA-256_SSWU_RO_")), ln
    # sign / verify round trip on both schemes (internal/test/scheme.go shape)
    x = (0x1234567890ABCDEF << 64 | 0xFEDCBA) % bls.ORDER
    xb = x.to_bytes(32, "big")
    for sch, pub in ((sbls.NewSchemeOnG1_bls12381(), bls.g2_commit(xb)[0][0]), (sbls.NewSchemeOnG2_bls12381(), bls.g1_commit(xb)[0][0])):
        sig = sch.sign(xb, b"Hello Boneh-Lynn-Shacham")
        assert sch.verify(bytes(pub), b"Hello Boneh-Lynn-Shacham", sig)
        assert not sch.verify(bytes(pub), b"another message", sig)


def test_fused_verify_matches_hash_plus_pairing_check(bls):
    """kyb_bls12381_verify_g1 == batch_hash_g1 + batch_validate_pairing on 1024 (key, msg, sig) triples with
    forged entries and undecodable inputs sprinkled in."""
    n = 1024
    x = _scalars(b"fv/x", n)
    msgs = np.frombuffer(hashlib.shake_256(b"fv/m").digest(n * 32), dtype=np.uint8).reshape(n, 32).copy()
    X, _ = bls.g2_commit(x)
    Hm, st = bls.batch_hash_g1(msgs)
    assert not st.any()
    sig, _ = bls.g1_batch_mul(x, Hm)
    sig = sig.copy()
    sig[::5] = Hm[::5]  # forged: valid point, wrong value
    sig[7] = 0  # does not unmarshal (compression flag missing)
    X = X.copy()
    X[11] = 0xFF
    # points at infinity (the ZCash encoding 0xC0 00 ..): a dead pair contributes 1 -- signature alone, key alone, both
    inf1, inf2 = np.zeros(48, dtype=np.uint8), np.zeros(96, dtype=np.uint8)
    inf1[0] = inf2[0] = 0xC0
    sig[13] = inf1
    X[17] = inf2
    sig[19], X[19] = inf1, inf2
    ok_f, st_f = bls.batch_verify_g1(X, msgs, sig)
    G2 = np.tile(np.frombuffer(bls.G2_BASE, dtype=np.uint8), (n, 1))
    ok_r, st_r = bls.batch_validate_pairing(Hm, X, sig, G2)
    assert ((st_f != 0) == (st_r != 0)).all() and st_f[7] != 0 and st_f[11] != 0
    assert (ok_f == ok_r).all()
    exp = np.ones(n, dtype=bool)
    exp[::5] = False
    exp[[7, 11, 13, 17]] = False  # e(H, pk) = 1 or e(sig, g2) = 1 alone is false; index 19 (both dead) is 1 = 1
    assert (ok_f.astype(bool) == exp).all() and ok_f[19]


def test_wrong_length_key_or_signature_fails_alone(bls):
    """A signature / key of the wrong length (attacker-supplied bytes) is rejected by itself -- it neither shifts the
    elements packed after it nor makes the native call read past the buffer (sigs = [48 B, b""] used to
    reshape to one row).  The reference's Verify fails only the offending signature (bls.go:82-96)."""
    from kyber_amd.sign import bls as sbls

    n = 9
    xs = _scalars(b"wl/x", n)
    msgs = [b"msg-%02d" % i for i in range(n)]
    for sch, commit in ((sbls.NewSchemeOnG1_bls12381(), bls.g2_commit), (sbls.NewSchemeOnG2_bls12381(), bls.g1_commit)):
        pubs = [bytes(r) for r in commit(xs)[0]]
        sigs = [sch.sign(bytes(xs[i]), msgs[i]) for i in range(n)]
        assert sch.batch_verify(pubs, msgs, sigs).all()
        bad = list(sigs)
        bad[3] = b""                 # empty
        bad[5] = sigs[5][:-1]        # one byte short ...
        bad[6] = sigs[6] + b"\x00"   # ... one byte long: the lengths compensate, nothing may shift
        res = sch.batch_verify(pubs, msgs, bad)
        assert list(res) == [i not in (3, 5, 6) for i in range(n)]
        badk = list(pubs)
        badk[0] = pubs[0][:-2]
        res = sch.batch_verify(badk, msgs, sigs)
        assert list(res) == [i != 0 for i in range(n)]
    # the low-level entry point reports the lane as BAD_POINT and leaves the others alone
    sch = sbls.NewSchemeOnG1_bls12381()
    pubs = [bytes(r) for r in bls.g2_commit(xs)[0]]
    m32 = [hashlib.sha256(m).digest() for m in msgs]
    sigs = [sch.sign(bytes(xs[i]), m32[i]) for i in range(n)]
    sigs[4] = sigs[4][:10]
    ok, st = bls.batch_verify_g1(pubs, m32, sigs)
    assert st[4] == 1 and ok[4] == 0 and ok.sum() == n - 1 and not np.delete(st, 4).any()
    with pytest.raises(ValueError):
        bls.batch_verify_g1(pubs[:-1], m32, sigs)


# ------------------------------------------------------------------ call flags (KYB_F_*)
def _off_subgroup_g1():
    x = 1
    while True:
        y = O.fp_sqrt((x * x * x + 4) % O.P)
        if y is not None and not O.g1_in_subgroup((x, y)):
            return (x, y)
        x += 1


def test_mul_uncompressed_in_out_and_trusted_flags(bls):
    rng = random.Random(21)
    n = 12
    ks = [rng.randrange(O.R) for _ in range(n)]
    hs = [rng.randrange(1, O.R) for _ in range(n)]
    P1 = [O.g1_mul(h, O.G1_GEN) for h in hs]
    P2 = [O.g2_mul(h, O.G2_GEN) for h in hs[:4]]
    P1[2] = None
    kb = b"".join(k.to_bytes(32, "big") for k in ks)
    U, UO, T = bls.F_UNCOMPRESSED, bls.F_UNCOMPRESSED_OUT, bls.F_TRUSTED(0)
    for flags in (0, U, UO, U | UO, U | UO | T, T):
        ser_in = O.g1_serialize_unc if flags & U else O.g1_compress
        ser_out = O.g1_serialize_unc if flags & UO else O.g1_compress
        out, st = bls.g1_batch_mul(kb, b"".join(ser_in(p) for p in P1), flags)
        assert not st.any()
        for i in range(n):
            assert bytes(out[i]) == ser_out(O.g1_mul(ks[i], P1[i]) if P1[i] else None), (flags, i)
        ser_in = O.g2_serialize_unc if flags & U else O.g2_compress
        ser_out = O.g2_serialize_unc if flags & UO else O.g2_compress
        out, st = bls.g2_batch_mul(kb[:4 * 32], b"".join(ser_in(p) for p in P2), flags)
        assert not st.any()
        for i in range(4):
            assert bytes(out[i]) == ser_out(O.g2_mul(ks[i], P2[i])), (flags, i)
    # same-base with an uncompressed base and uncompressed outputs
    out, st = bls.g1_commit(kb, O.g1_serialize_unc(P1[0]), U | UO)
    assert not st.any() and bytes(out[5]) == O.g1_serialize_unc(O.g1_mul(ks[5], P1[0]))
    # malformed uncompressed inputs and a point outside the subgroup
    c = _off_subgroup_g1()
    good = O.g1_serialize_unc(P1[0])
    bad_y = good[:-1] + bytes([good[-1] ^ 1])
    batch = good + bad_y + O.g1_serialize_unc(c) + bytes([good[0] | 0x80]) + good[1:]
    out, st = bls.g1_batch_mul(kb[:4 * 32], batch, U)
    assert list(st) == [0, 1, 2, 1] and not out[1:].any()
    out, st = bls.g1_batch_mul((5).to_bytes(32, "big"), O.g1_compress(c), T)  # the caller vouched for it: unchecked
    assert st[0] == 0


def test_pair_check_verify_and_msm_flags_agree_with_the_checked_path(bls):
    import torch

    n = 256
    k = torch.from_numpy(_scalars(b"flags/k", n)).cuda()
    h = torch.from_numpy(_scalars(b"flags/h", n)).cuda()
    g1b = torch.from_numpy(np.frombuffer(bls.G1_BASE, dtype=np.uint8).copy()).cuda()
    g2b = torch.from_numpy(np.frombuffer(bls.G2_BASE, dtype=np.uint8).copy()).cuda()
    U, UO = bls.F_UNCOMPRESSED, bls.F_UNCOMPRESSED_OUT
    Hc, _ = bls._mul(1, h, g1b, True)
    Hu, _ = bls._mul(1, h, g1b, True, UO)
    Xc, _ = bls._mul(2, k, g2b, True)
    Xu, _ = bls._mul(2, k, g2b, True, UO)
    sig_c, _ = bls.g1_batch_mul(k, Hc)
    sig_u, _ = bls.g1_batch_mul(k, Hc, UO)
    sig_c[7] = Hc[7]
    sig_u[7] = Hu[7]  # one wrong signature
    G2c = g2b.repeat(n, 1)
    G2u = torch.from_numpy(np.frombuffer(O.g2_serialize_unc(O.G2_GEN), dtype=np.uint8).copy()).cuda().repeat(n, 1)
    ok0, st0 = bls.batch_validate_pairing(Hc, Xc, sig_c, G2c)
    exp = np.ones(n, dtype=np.uint8)
    exp[7] = 0
    assert not st0.any().item() and (ok0.cpu().numpy() == exp).all()
    for flags, (a, b, c, d) in ((bls.F_TRUSTED(0) | bls.F_TRUSTED(1) | bls.F_TRUSTED(3), (Hc, Xc, sig_c, G2c)),
                                (bls.F_TRUSTED_ALL, (Hc, Xc, sig_c, G2c)),
                                (U, (Hu, Xu, sig_u, G2u)),
                                (U | bls.F_TRUSTED_ALL, (Hu, Xu, sig_u, G2u))):
        ok, st = bls.batch_validate_pairing(a, b, c, d, flags)
        assert not st.any().item() and (ok.cpu().numpy() == exp).all(), flags
    gt0, _ = bls.batch_pair(Hc, Xc)
    gt1, st = bls.batch_pair(Hu, Xu, U | bls.F_TRUSTED(0) | bls.F_TRUSTED(1))
    assert not st.any().item() and torch.equal(gt0, gt1)
    # an unvalidated operand outside the subgroup is still caught when the others are trusted
    bad = sig_c.clone()
    bad[3] = torch.from_numpy(np.frombuffer(O.g1_compress(_off_subgroup_g1()), dtype=np.uint8).copy()).cuda()
    ok, st = bls.batch_validate_pairing(Hc, Xc, bad, G2c, bls.F_TRUSTED(0) | bls.F_TRUSTED(1) | bls.F_TRUSTED(3))
    assert st[3].item() == 2 and ok[3].item() == 0 and st.sum().item() == 2
    # fused verify: trusted keys / uncompressed keys + signatures
    msgs = torch.from_numpy(np.frombuffer(hashlib.shake_256(b"flags/m").digest(n * 32), dtype=np.uint8).reshape(n, 32).copy()).cuda()
    Hm, _ = bls.batch_hash_g1(msgs)
    s_c, _ = bls.g1_batch_mul(k, Hm)
    s_u, _ = bls.g1_batch_mul(k, Hm, UO)
    s_c[9] = Hm[9]
    s_u[9] = Hu[9]
    v0, st = bls.batch_verify_g1(Xc, msgs, s_c)
    expv = np.ones(n, dtype=np.uint8)
    expv[9] = 0
    assert not st.any().item() and (v0.cpu().numpy() == expv).all()
    for flags, (kk, ss) in ((bls.F_TRUSTED(0), (Xc, s_c)), (U, (Xu, s_u)), (U | bls.F_TRUSTED(0) | bls.F_TRUSTED(1), (Xu, s_u))):
        v, st = bls.batch_verify_g1(kk, msgs, ss, flags=flags)
        assert not st.any().item() and (v.cpu().numpy() == expv).all(), flags
    # MSM: validated / uncompressed inputs give the same point
    for grp, (pc, pu) in ((1, (Hc, Hu)), (2, (Xc, Xu))):
        r0, st = bls.ENGINE.msm(grp, k, pc)
        assert not st.any().item()
        for flags, p in ((bls.F_TRUSTED(0), pc), (U, pu), (U | bls.F_TRUSTED(0), pu)):
            r, st = bls.ENGINE.msm(grp, k, p, flags)
            assert not st.any().item() and torch.equal(r, r0), (grp, flags)


def test_bn256_accepts_the_documented_flags_and_rejects_stray_bits():
    """bn256 accepts (and has nothing to do for) the documented flags of a call; bits that do not apply to the call --
    KYB_F_TRUSTED(i) beyond its point arguments, undefined bits -- are KYB_E_ARG on both suites."""
    from kyber_amd.pairing import bn256 as bn

    k = _scalars(b"bnflags/k", 8)
    P, _ = bn.g1_commit(k)
    a, st = bn.g1_batch_mul(k, P)
    b, st2 = bn.g1_batch_mul(k, P, bn.F_UNCOMPRESSED | bn.F_UNCOMPRESSED_OUT | bn.F_TRUSTED(0))
    assert not st.any() and not st2.any() and (a == b).all()
    r0, _ = bn.g1_msm(k, P)
    r1, _ = bn.g1_msm(k, P, bn.F_TRUSTED(0) | bn.F_UNCOMPRESSED)
    assert (r0 == r1).all()
    from kyber_amd.pairing import bls12381 as bl

    Q, _ = bn.g2_commit(k)
    for call in (lambda: bn.g1_batch_mul(k, P, bn.F_TRUSTED(1)), lambda: bn.g1_batch_mul(k, P, 0x10000),
                 lambda: bn.g1_msm(k, P, bn.F_TRUSTED_ALL), lambda: bn.batch_pair(P, Q, bn.F_TRUSTED(2)),
                 lambda: bn.batch_pair(P, Q, bn.F_UNCOMPRESSED_OUT), lambda: bn.g1_batch_unmarshal(P, 1),
                 lambda: bl.g1_commit(k, None, bl.F_TRUSTED(3)), lambda: bl.g1_msm(k, bl.g1_commit(k)[0], 0x8)):
        with pytest.raises(RuntimeError):
            call()


def test_fused_verify_g2_matches_hash_plus_pairing_check(bls):
    """kyb_bls12381_verify_g2 (signatures on G2, keys on G1) == batch_hash_g2 + batch_validate_pairing with the
    argument order of bls.go:51-53, with forged and undecodable entries."""
    n = 256
    x = _scalars(b"fv2/x", n)
    msgs = np.frombuffer(hashlib.shake_256(b"fv2/m").digest(n * 32), dtype=np.uint8).reshape(n, 32).copy()
    X, _ = bls.g1_commit(x)
    Hm, st = bls.batch_hash_g2(msgs)
    assert not st.any()
    sig, _ = bls.g2_batch_mul(x, Hm)
    sig = sig.copy()
    sig[::7] = Hm[::7]  # forged: valid point, wrong value
    sig[5] = 0
    X = X.copy()
    X[9] = 0xFF
    ok_f, st_f = bls.batch_verify_g2(X, msgs, sig)
    G1 = np.tile(np.frombuffer(bls.G1_BASE, dtype=np.uint8), (n, 1))
    ok_r, st_r = bls.batch_validate_pairing(G1, sig, X, Hm)
    assert ((st_f != 0) == (st_r != 0)).all() and st_f[5] != 0 and st_f[9] != 0
    assert (ok_f == ok_r).all()
    exp = np.ones(n, dtype=bool)
    exp[::7] = False
    exp[[5, 9]] = False
    assert (ok_f.astype(bool) == exp).all()
    ok_t, st_t = bls.batch_verify_g2(X, msgs, sig, flags=bls.F_TRUSTED(0))
    exp_t = exp.copy()
    assert (ok_t.astype(bool)[np.arange(n) != 9] == exp_t[np.arange(n) != 9]).all()


def test_batch_unmarshal_zcash_fixtures_flags_and_device_path(bls, golden_dir):
    """kyb_bls12381_g*_unmarshal = N x UnmarshalBinary (+ MarshalBinary): the reference's 34 deserialization fixtures in
    one batch per group, the uncompressed output feeding a later trusted call, and the device-tensor entry point."""
    import torch

    d = json.load(open(os.path.join(golden_dir, "bls12381_zcash.json")))
    U, UO, T = bls.F_UNCOMPRESSED, bls.F_UNCOMPRESSED_OUT, bls.F_TRUSTED(0)
    for grp, fn, size, dec, unc in (("G1", bls.g1_batch_unmarshal, 48, O.g1_decompress, O.g1_serialize_unc),
                                    ("G2", bls.g2_batch_unmarshal, 96, O.g2_decompress, O.g2_serialize_unc)):
        cases = [e for e in d[grp] if len(bytes.fromhex(e["hex"])) == size]
        assert len(cases) >= 12
        batch = b"".join(bytes.fromhex(e["hex"]) for e in cases)
        out, st = fn(batch)
        out_u, st_u = fn(batch, UO)
        assert out.shape == (len(cases), size) and out_u.shape == (len(cases), 2 * size)
        assert (st == st_u).all()
        for i, e in enumerate(cases):
            buf = bytes.fromhex(e["hex"])
            assert (st[i] == 0) == e["valid"], (grp, e["name"], st[i])
            if e["valid"]:
                assert bytes(out[i]) == buf and bytes(out_u[i]) == unc(dec(buf))
            else:
                assert not out[i].any() and not out_u[i].any()
        # device tensors, same answers
        t = torch.frombuffer(bytearray(batch), dtype=torch.uint8).cuda()
        out_d, st_d = fn(t, UO)
        assert bytes(out_d.cpu().numpy().tobytes()) == out_u.tobytes() and bytes(st_d.cpu().numpy().tobytes()) == st_u.tobytes()
    # validate once, then multiply the uncompressed affine points without re-checking them
    rng = random.Random(77)
    n = 64
    hs = [rng.randrange(1, O.R) for _ in range(n)]
    comp = bls.g1_commit(b"".join(h.to_bytes(32, "big") for h in hs))[0]
    aff, st = bls.g1_batch_unmarshal(comp, UO)
    assert not st.any()
    ks = _scalars(b"unmarshal-then-mul", n)
    a, sa = bls.g1_batch_mul(ks, comp)
    b, sb = bls.g1_batch_mul(ks, aff, U | T)
    assert not sa.any() and not sb.any() and (a == b).all()
    # off-subgroup: rejected with status 2, let through only when the caller vouches for it
    c = O.g1_compress(_off_subgroup_g1())
    out, st = bls.g1_batch_unmarshal(c + comp[0].tobytes())
    assert list(st) == [2, 0] and not out[0].any() and bytes(out[1]) == comp[0].tobytes()
    out, st = bls.g1_batch_unmarshal(c, T)
    assert st[0] == 0 and bytes(out[0]) == c
    out, st = bls.g1_batch_unmarshal(b"")
    assert out.shape == (0, 48) and st.shape == (0,)


def test_ibe_vector_pins_pair_bytes_on_the_engine(bls, golden_dir):
    """encrypt/ibe/ibe_test.go:202-245 through the engine: DecryptCCAonG1 (ibe.go:100-135) hashes the 576 bytes of
    Suite.Pair(U, beacon) -- the one vector in the reference that fixes BLS12-381 GT BYTES.  The GPU's bytes must
    decrypt to deadbeef x 4; the same vector placed in a 256-pair batch must give the same bytes in every slot; and
    the CCA check r*P == U (ibe.go:123-131) is replayed with the engine's G1 multiplication on a ciphertext made by
    the oracle (the vector's own U predates today's h3, see tests/test_oracle_bls12381.py)."""
    from tests.test_oracle_bls12381 import _ibe_decrypt, _ibe_h3
    v = json.load(open(os.path.join(golden_dir, "bls12381_ibe.json")))
    U, beacon = bytes.fromhex(v["U_g1"]), bytes.fromhex(v["beacon_g2"])
    V, W, want = bytes.fromhex(v["V"]), bytes.fromhex(v["W"]), bytes.fromhex(v["expected"])
    for flags in (0, bls.F_TRUSTED(0) | bls.F_TRUSTED(1)):
        gt, st = bls.batch_pair(U, beacon, flags)
        assert st[0] == 0
        assert _ibe_decrypt(bytes(gt[0]), V, W, v["tags"])[1] == want
    gts, st = bls.batch_pair(U * 256, beacon * 256)
    assert not np.asarray(st).any() and all(bytes(g) == bytes(gt[0]) for g in gts)
    assert bytes(gt[0]) == O.pair_bytes(U, beacon)
    # a ciphertext of our own: encrypt on the oracle, decrypt with the engine's Pair and G1 Mul
    s = 0x1CEB00DA % O.R
    qid = O.hash_to_g2(b"passtherand", bls.DOMAIN_G2)
    msg, sigma = b"kyberhip ibe msg", bytes(range(16))
    r = _ibe_h3(sigma, msg, v["tags"])
    Uc = O.g1_compress(O.g1_mul(r, O.G1_GEN))
    gid_r = O.f12_pow(O.pair(O.g1_mul(s, O.G1_GEN), qid), r)
    Vc = bytes(a ^ b for a, b in zip(sigma, hashlib.sha256(b"IBE-H2" + O.gt_to_bytes(gid_r)).digest()[:16]))
    Wc = bytes(a ^ b for a, b in zip(msg, hashlib.sha256(b"IBE-H4" + sigma).digest()[:16]))
    gt, st = bls.batch_pair(Uc, O.g2_compress(O.g2_mul(s, qid)))
    sigma2, msg2 = _ibe_decrypt(bytes(gt[0]), Vc, Wc, v["tags"])
    assert (sigma2, msg2) == (sigma, msg)
    rp, st = bls.g1_batch_mul(_ibe_h3(sigma2, msg2, v["tags"]).to_bytes(32, "big"), bls.G1_BASE)
    assert st[0] == 0 and bytes(rp[0]) == Uc
