"""kyb_bls12381_verify_g1_same_msg: sign/bls Verify for many (public key, signature) pairs over ONE message -- the
verification loop of tbls.Recover (sign/tbls/tbls.go:118-131: one msg, a different public share public.Eval(idx).V per
partial signature).  H(msg) is hashed once per call by one extra workgroup of the operand kernel and dealt to every
pairing.  Held against the oracle, against kyb_bls12381_verify_g1 with the message repeated (byte for byte), and through
the tbls host mirror."""
import hashlib
import random

import numpy as np
import pytest

from oracle import bls12381 as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def bls():
    import torch

    assert torch.cuda.is_available()
    from kyber_amd.pairing import bls12381

    return bls12381


def _scalars(label, n):
    a = np.frombuffer(hashlib.shake_256(label).digest(n * 32), dtype=np.uint8).reshape(n, 32).copy()
    a[:, 0] &= 0x3F
    return a


def test_oracle_signatures_over_one_message(bls):
    msg = b"round 17 of the beacon"
    xs = [(0x1234567 * (i + 1)) % O.R for i in range(5)]
    H = O.hash_to_g1(msg, bls.DOMAIN_G1)
    keys = [O.g2_compress(O.g2_mul(x, O.G2_GEN)) for x in xs]
    sigs = [O.g1_compress(O.g1_mul(x, H)) for x in xs]
    ok, st = bls.batch_verify_g1_same_msg(keys, msg, sigs)
    assert list(ok) == [1] * 5 and not st.any()
    ok, st = bls.batch_verify_g1_same_msg(keys, msg + b"!", sigs)            # another message
    assert not ok.any() and not st.any()
    ok, st = bls.batch_verify_g1_same_msg(keys[1:] + keys[:1], msg, sigs)    # keys rotated against the signatures
    assert not ok.any() and not st.any()
    ok, st = bls.batch_verify_g1_same_msg(keys, msg, sigs, dst=b"another domain tag")
    assert not ok.any()
    # the empty message, a single element, no element
    He = O.hash_to_g1(b"", bls.DOMAIN_G1)
    ok, st = bls.batch_verify_g1_same_msg(keys[:1], b"", [O.g1_compress(O.g1_mul(xs[0], He))])
    assert list(ok) == [1] and not st.any()
    ok, st = bls.batch_verify_g1_same_msg([], msg, [])
    assert len(ok) == 0 and len(st) == 0
    # wrong-length elements fail alone; an undecodable key / signature carries its status
    bad_key = bytes([keys[2][0] & 0x7F]) + keys[2][1:]                       # compression bit cleared
    ok, st = bls.batch_verify_g1_same_msg([keys[0], keys[1][:-1], bad_key, keys[3]], msg, [sigs[0], sigs[1], sigs[2], sigs[3][:-2]])
    assert list(ok) == [1, 0, 0, 0] and st[0] == 0 and st[1] == 1 and st[2] != 0 and st[3] == 1


@pytest.mark.parametrize("n", [1, 63, 4096, 1 << 16])
def test_same_message_equals_the_general_verification(bls, n):
    """device-resident batch: valid signatures with forged, undecodable and infinite ones scattered through it; verdicts
    and statuses byte for byte those of kyb_bls12381_verify_g1 with the message repeated n times; validated keys too"""
    import torch

    msg = hashlib.sha256(b"one message for the whole batch %d" % n).digest()
    x = torch.from_numpy(_scalars(b"same-msg/x/%d" % n, n)).cuda()
    X, stx = bls.g2_commit(x)
    Hm, _ = bls.batch_hash_g1([msg])
    Hd = torch.from_numpy(np.asarray(Hm)).cuda().repeat(n, 1)
    sig, sts = bls.g1_batch_mul(x, Hd)
    assert not stx.any().item() and not sts.any().item()
    sig = sig.clone()
    rng = random.Random(n)
    forged = sorted({rng.randrange(n) for _ in range(max(1, n // 50))})
    for i in forged:
        sig[i] = sig[(i + 1) % n] if n > 1 else torch.from_numpy(np.frombuffer(O.g1_compress(O.G1_GEN), dtype=np.uint8).copy()).cuda()
    if n >= 63:
        sig[7, 0] &= 0x7F                                                    # compression bit cleared: undecodable
        sig[11] = torch.from_numpy(np.frombuffer(bytes([0xC0]) + bytes(47), dtype=np.uint8).copy()).cuda()  # infinity
        X = X.clone()
        X[13, 5] ^= 1                                                        # a key off the curve (or not a square root)
    msgs = torch.from_numpy(np.frombuffer(msg, dtype=np.uint8).copy()).cuda().repeat(n, 1)
    m1 = torch.from_numpy(np.frombuffer(msg, dtype=np.uint8).copy()).cuda()
    for flags in (0, bls.F_TRUSTED(1)):
        ok_g, st_g = bls.batch_verify_g1(X, msgs, sig, flags=flags)
        ok_m, st_m = bls.batch_verify_g1_same_msg(X, m1, sig, flags=flags)
        assert torch.equal(ok_g, ok_m) and torch.equal(st_g, st_m), (n, flags)
    good = torch.ones(n, dtype=torch.bool)
    good[forged] = False
    if n >= 63:
        good[[7, 11, 13]] = False
    assert torch.equal(ok_m.cpu() == 1, good) or n == 1
    # host buffers through the same entry point
    ok_h, st_h = bls.batch_verify_g1_same_msg(X.cpu().numpy(), msg, sig.cpu().numpy())
    ok0, st0 = bls.batch_verify_g1_same_msg(X, m1, sig)
    assert np.array_equal(np.asarray(ok_h), ok0.cpu().numpy()) and np.array_equal(np.asarray(st_h), st0.cpu().numpy())


def test_tbls_recover_on_bls12381_goes_through_the_same_message_entry(bls, monkeypatch):
    """internal/test/threshold.go shape over BLS12-381: t of n partial signatures recover the signature the secret
    would produce; Recover's verification is ONE same-message call (the general entry point is not used)."""
    from kyber_amd.share import poly
    from kyber_amd.sign import bls as sbls, tbls

    rng = random.Random(12)
    rand = lambda k: bytes(rng.randrange(256) for _ in range(k))
    suite = bls.NewSuite()
    t, n = 4, 9
    pri = poly.PriPoly.new(suite.G2(), t, rand=rand)
    pub = pri.Commit(None)
    sch = tbls.NewThresholdSchemeOnG1_bls12381()
    msg = b"Hello threshold Boneh-Lynn-Shacham"
    partials = [sch.sign(pri.Eval(i), msg) for i in range(n)]
    assert all(sch.verify_partial(pub, msg, s) for s in partials[:3])
    partials[2] = partials[2][:12] + bytes([partials[2][12] ^ 1]) + partials[2][13:]  # corrupt one
    calls = {"same_msg": 0, "general": 0}
    real_m, real_g = bls.batch_verify_g1_same_msg, bls.batch_verify_g1
    monkeypatch.setattr(bls, "batch_verify_g1_same_msg", lambda *a, **k: (calls.__setitem__("same_msg", calls["same_msg"] + 1), real_m(*a, **k))[1])
    monkeypatch.setattr(bls, "batch_verify_g1", lambda *a, **k: (calls.__setitem__("general", calls["general"] + 1), real_g(*a, **k))[1])
    sig = sch.recover(pub, msg, partials, t, n)
    assert calls == {"same_msg": 1, "general": 0}
    plain = sbls.NewSchemeOnG1_bls12381()
    assert sig == plain.sign(pri.coeffs[0].MarshalBinary(), msg)
    assert plain.verify(pub.Commit().MarshalBinary(), msg, sig)
    with pytest.raises(ValueError):
        sch.recover(pub, msg, partials[:3], t, n)
