"""kyb_bls12381_verify_g1_same_key: sign/bls Verify (sign/bls/bls.go:82-96) for many (message, signature) pairs under ONE
public key -- a drand chain, one tbls participant's partial signatures (sign/tbls/tbls.go:100-107) -- on program
VERIFYK of the tower machine (both Miller loops from line tables; the key's table built on the device by one lane,
bls12381_keylines.cuh).  Held against the general fused verification, the oracle and the reference's drand fixture."""
import hashlib
import json
import os
import struct

import numpy as np
import pytest

from oracle import bls12381 as O

pytestmark = pytest.mark.gpu
VKEY_SLOTS_PLUS = 10  # more rejected keys than the cache has slots (bls12381_pair.hip VKEY_SLOTS = 8)


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


def test_drand_fixture_and_oracle_signatures(bls, golden_dir):
    """the reference's drand signature (kilic/suite_test.go:17-72: signature on G1 under the G2 domain tag) and three
    oracle-made signatures under one key verify; each fails under another key or for another message"""
    D = json.load(open(os.path.join(golden_dir, "bls12381_drand.json")))
    f = D["sig_on_g1"]
    msg = hashlib.sha256(struct.pack(">Q", f["round"])).digest()
    pk, sig = bytes.fromhex(f["pk_g2"]), bytes.fromhex(f["sig_g1"])
    dst = D["dst_g2"].encode()
    ok, st = bls.batch_verify_g1_same_key(pk, [msg, msg[::-1]], [sig, sig], dst)
    assert list(ok) == [1, 0] and not st.any()
    assert list(bls.batch_verify_g1_same_key(pk, [msg], [sig])[0]) == [0]          # default domain: must fail
    x = 0x5A17C0DE % O.R
    X = O.g2_compress(O.g2_mul(x, O.G2_GEN))
    msgs = [b"beacon-%02d-padding-to-32-bytes!!" % i for i in range(3)]
    sigs = [O.g1_compress(O.g1_mul(x, O.hash_to_g1(m, bls.DOMAIN_G1))) for m in msgs]
    ok, st = bls.batch_verify_g1_same_key(X, msgs, sigs)
    assert list(ok) == [1, 1, 1] and not st.any()
    ok, _ = bls.batch_verify_g1_same_key(O.g2_compress(O.g2_mul(x + 1, O.G2_GEN)), msgs, sigs)
    assert not ok.any()
    ok, _ = bls.batch_verify_g1_same_key(X, msgs, sigs[1:] + sigs[:1])
    assert not ok.any()


def test_same_key_equals_the_general_verification_at_2p16(bls):
    """2^16 triples under one key, device-resident: forged and undecodable signatures and signatures at infinity
    scattered through the batch; byte for byte the verdicts and statuses of kyb_bls12381_verify_g1 with the key repeated;
    then another key (table rebuilt), the first again (rebuilt again), uncompressed + vouched-for forms."""
    import torch

    n = 1 << 16
    msgs = torch.from_numpy(np.frombuffer(hashlib.shake_256(b"sk/m").digest(n * 32), dtype=np.uint8).reshape(n, 32).copy()).cuda()
    Hm, st = bls.batch_hash_g1(msgs)
    assert not st.any().item()
    inf1 = torch.zeros(48, dtype=torch.uint8, device="cuda")
    inf1[0] = 0xC0
    for which, x in enumerate((0x1F2E3D4C5B6A79 % O.R, (O.R - 5), 0x1F2E3D4C5B6A79 % O.R)):
        xb = torch.from_numpy(np.frombuffer(x.to_bytes(32, "big"), dtype=np.uint8).copy()).cuda()
        X = bls.g2_commit(xb.view(1, 32))[0][0].contiguous()
        sig, _ = bls.g1_batch_mul(xb.repeat(n, 1), Hm)
        sig = sig.clone()
        sig[::7] = Hm[::7]          # forged: a valid point, the wrong one
        sig[5] = 0                  # does not unmarshal
        sig[n - 1] = 0xFF
        sig[11] = inf1              # e(H, X) = 1 alone is false
        ok, st = bls.batch_verify_g1_same_key(X, msgs, sig)
        ok_r, st_r = bls.batch_verify_g1(X.repeat(n, 1), msgs, sig)
        torch.cuda.synchronize()
        assert torch.equal(ok, ok_r) and torch.equal(st, st_r), which
        exp = torch.ones(n, dtype=torch.bool, device="cuda")
        exp[::7] = False
        exp[[5, 11, n - 1]] = False
        assert torch.equal(ok.bool(), exp) and st[5].item() != 0 and st[n - 1].item() != 0
    # the same key uncompressed and vouched for (no checks on it): same verdicts
    Xu = bls.g2_batch_unmarshal(X.view(1, 96), bls.F_UNCOMPRESSED_OUT)[0][0].contiguous()
    sigu, stu = bls.g1_batch_unmarshal(sig, bls.F_UNCOMPRESSED_OUT)
    good = (stu == 0)
    ok_u, st_u = bls.batch_verify_g1_same_key(Xu, msgs, sigu, flags=bls.F_UNCOMPRESSED | bls.F_TRUSTED(0))
    torch.cuda.synchronize()
    assert torch.equal(ok_u[good], ok[good])


def test_rejected_and_infinite_keys(bls):
    """a key UnmarshalBinary rejects fails every element with its status; a key outside the subgroup is status 2 (unless
    vouched for); the key at infinity leaves e(-sig, g2) == 1, true only for the signature at infinity"""
    n = 200
    msgs = [hashlib.sha256(b"rk%d" % i).digest() for i in range(n)]
    x = 77
    sigs = [O.g1_compress(O.g1_mul(x, O.hash_to_g1(m, bls.DOMAIN_G1))) for m in msgs[:3]] * (n // 3 + 1)
    sigs = sigs[:n]
    bad = bytes(96)                                   # compression bit clear
    ok, st = bls.batch_verify_g1_same_key(bad, msgs, sigs)
    assert not ok.any() and (st == 1).all()
    xx = 1
    while O.f2_sqrt(O.f2_add(O.f2_mul(O.f2_sqr((xx, 1)), (xx, 1)), (4, 4))) is not None:
        xx += 1                                       # no point above this x: the wave's square root finds none
    nox = bytearray((1).to_bytes(48, "big") + xx.to_bytes(48, "big"))
    nox[0] |= 0x80
    ok, st = bls.batch_verify_g1_same_key(bytes(nox), msgs, sigs)
    assert not ok.any() and (st == 1).all()
    xx = 1
    while True:                                       # a twist point outside G2
        c = (xx, 1)
        y = O.f2_sqrt(O.f2_add(O.f2_mul(O.f2_sqr(c), c), (4, 4)))
        if y is not None and not O.g2_in_subgroup((c, y)):
            off = O.g2_compress((c, y))
            break
        xx += 1
    ok, st = bls.batch_verify_g1_same_key(off, msgs, sigs)
    assert not ok.any() and (st == 2).all()
    # the rule is read off the key's own line walk (g2_walk_end_is_minus_psi): a key of order 13 -- the walk's point runs
    # through +-key and infinity on the way -- and such a point plus a good key are outside too
    o13 = O.g2_mul(O.R * O.H2 // 169, (c, y))
    assert o13 is not None and O.g2_mul(13, o13) is None
    for K in (o13, O.g2_add(o13, O.g2_mul(x, O.G2_GEN))):
        ok, st = bls.batch_verify_g1_same_key(O.g2_compress(K), msgs, sigs)
        assert not ok.any() and (st == 2).all()
    inf2 = b"\xc0" + bytes(95)
    sigs2 = list(sigs)
    sigs2[4] = b"\xc0" + bytes(47)
    ok, st = bls.batch_verify_g1_same_key(inf2, msgs, sigs2)
    assert not st.any() and list(np.nonzero(ok)[0]) == [4]
    assert not bls.batch_verify_g1_same_key(bytes(95), msgs, sigs)[0].any()   # a key of the wrong length
    # ... and a valid key afterwards gets a fresh table
    X = O.g2_compress(O.g2_mul(x, O.G2_GEN))
    sg = [O.g1_compress(O.g1_mul(x, O.hash_to_g1(m, bls.DOMAIN_G1))) for m in msgs[:4]]
    ok, st = bls.batch_verify_g1_same_key(X, msgs[:4], sg)
    assert ok.all() and not st.any()


def test_sign_bls_mirror_routes_one_signer_batches_to_the_same_key_program(bls):
    """kyber_amd/sign/bls.py: a batch whose keys are all the same (>= SAME_KEY_MIN) goes through VERIFYK; same answers as
    the per-key path, messages of two lengths, a forged signature"""
    from kyber_amd.sign import bls as sbls

    sch = sbls.NewSchemeOnG1_bls12381()
    x = (0x77AA55 << 40 | 0x1234) % bls.ORDER
    xb = x.to_bytes(32, "big")
    pub = bytes(bls.g2_commit(xb)[0][0])
    n = sch.SAME_KEY_MIN + 9
    msgs = [(b"round-%05d" % i) + (b"!" if i % 3 == 0 else b"") for i in range(n)]
    Hs = sch.batch_hash(msgs)
    sigs = [bytes(r) for r in np.asarray(bls.g1_batch_mul(np.tile(np.frombuffer(xb, dtype=np.uint8), (n, 1)), Hs)[0])]
    sigs[17] = sigs[18]
    ok = sch.batch_verify_same_key(pub, msgs, sigs)
    assert list(np.nonzero(~ok)[0]) == [17]
    ok2 = sch.batch_verify([pub] * (sch.SAME_KEY_MIN - 1), msgs[:sch.SAME_KEY_MIN - 1], sigs[:sch.SAME_KEY_MIN - 1])  # the per-key path
    assert list(np.nonzero(~ok2)[0]) == [17]


def test_empty_batch_single_element_and_empty_messages(bls):
    """the ragged ends: no elements at all, one element, messages of length zero (hash-to-curve of the empty string,
    RFC 9380's first test vector shape) -- the verdicts of the general program"""
    x = 1234567
    X = O.g2_compress(O.g2_mul(x, O.G2_GEN))
    ok, st = bls.batch_verify_g1_same_key(X, [], [])
    assert len(ok) == 0 and len(st) == 0
    m = b"one"
    s = O.g1_compress(O.g1_mul(x, O.hash_to_g1(m, bls.DOMAIN_G1)))
    ok, st = bls.batch_verify_g1_same_key(X, [m], [s])
    assert list(ok) == [1] and list(st) == [0]
    e = O.g1_compress(O.g1_mul(x, O.hash_to_g1(b"", bls.DOMAIN_G1)))
    ok, st = bls.batch_verify_g1_same_key(X, [b"", b"", b""], [e, s, e])
    assert list(ok) == [1, 0, 1] and not st.any()
    ok2, st2 = bls.batch_verify_g1([X] * 3, [b"", b"", b""], [e, s, e])
    assert list(ok2) == list(ok) and list(st2) == list(st)


def test_alternating_between_a_handful_of_keys_costs_a_copy_not_a_walk(bls):
    """A verifier over a small committee (sign/bls/bls.go:82-96 called with three different keys in turn): since round 5
    the last eight keys' line tables are kept per stream, so after each key has been seen once a switch costs a 17 KB copy
    instead of the 9 ms walk of the key through the Miller loop.  Verdicts must not depend on the order the keys were
    seen in; every later call is held to 1.2 x the steady-state time of a single key."""
    import torch

    n = 1 << 12
    msgs = torch.from_numpy(_scalars(b"committee/msgs", n)).cuda()
    Hm, _ = bls.batch_hash_g1(msgs)
    keys, sigs = [], []
    for j in range(3):
        x = torch.from_numpy(_scalars(b"committee/x/%d" % j, 1)).cuda()
        keys.append(bls.g2_commit(x)[0][0].contiguous())
        sigs.append(bls.g1_batch_mul(x.repeat(n, 1), Hm)[0])
    sigs[1] = sigs[1].clone()
    sigs[1][5] = sigs[1][6]  # one forged signature under the second key

    def timed(fn, reps):
        ts = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = fn()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        return r, ts

    import ctypes

    import warnings

    from kyber_amd import _lib
    from kyber_amd.pairing._engine import _stream

    def stats():  # (hits, builds, next slot) of this stream's key cache: the mechanism, asserted by what it did
        out = (ctypes.c_uint32 * 3)()
        _lib.check(_lib.load().kyb_bls12381_debug_vkey_stats(_stream(), out), "kyb_bls12381_debug_vkey_stats")
        return tuple(out)

    h0, b0, _ = stats()
    for j in range(3):  # first sight of every key: the walk
        ok, st = bls.batch_verify_g1_same_key(keys[j], msgs, sigs[j])
        assert not st.any().item() and int(ok.sum().item()) == (n - 1 if j == 1 else n)
    h1, b1, _ = stats()
    assert b1 - b0 <= 3 and h1 == h0  # at most three walks (a key met by an earlier test may still be active), no hit yet
    _, steady = timed(lambda: bls.batch_verify_g1_same_key(keys[0], msgs, sigs[0]), 7)
    steady_ms = sorted(steady)[len(steady) // 2]
    seq = [1, 2, 0, 2, 1, 0, 1, 2, 0, 1]
    times = []
    for j in seq:
        (ok, st), ts = timed(lambda: bls.batch_verify_g1_same_key(keys[j], msgs, sigs[j]), 1)
        assert not st.any().item() and int(ok.sum().item()) == (n - 1 if j == 1 else n)
        assert (ok[5].item() == 0) == (j == 1)
        times.append(ts[0])
    # a walk costs ~9 ms on top of a ~7 ms call (2.3 x): the typical switch must stay within 1.2 x the steady state, and no
    # single one may look like a walk (the looser bound absorbs a scheduling hiccup on a shared box)
    # the mechanism: every switch of the sequence was a hit, none a walk (steady-state repeats of the active key are neither)
    h2, b2, _ = stats()
    assert b2 == b1 and h2 - h1 == len(seq) + 1, ((h1, b1), (h2, b2))
    # the timing: soft (ADVICE r5 -- a shared or throttled box can push a single un-repeated sample over any bound)
    times.sort()
    if not (times[len(times) // 2] <= 1.2 * steady_ms + 0.3 and times[-1] <= 1.8 * steady_ms + 0.5):
        warnings.warn(f"key-cache switches slower than expected: {times} against a steady state of {steady_ms} ms")
    # rejected keys and the point at infinity take no slot: a stream fed bad keys does not evict the committee's tables
    bad_key = bytearray(bytes(keys[0].cpu().numpy()))
    bad_key[5] ^= 0x55
    inf_key = bytes([0xC0]) + bytes(95)
    for _ in range(VKEY_SLOTS_PLUS):
        for kk in (bytes(bad_key), inf_key):
            ok, st = bls.batch_verify_g1_same_key(torch.from_numpy(np.frombuffer(kk, dtype=np.uint8).copy()).cuda(), msgs[:64].contiguous(), sigs[0][:64].contiguous())
            assert not ok.any().item()
    h3, b3, nx3 = stats()
    for j in (0, 1, 2):
        ok, st = bls.batch_verify_g1_same_key(keys[j], msgs, sigs[j])
        assert not st.any().item() and int(ok.sum().item()) == (n - 1 if j == 1 else n)
    h4, b4, nx4 = stats()
    assert b4 == b3 and nx4 == nx3 and h4 - h3 == 3, ((h3, b3, nx3), (h4, b4, nx4))
    # a key signatures do not belong to, between two cached ones: still rejected
    ok, _ = bls.batch_verify_g1_same_key(keys[2], msgs, sigs[0])
    assert not ok.any().item()
    # more distinct keys than slots: the oldest is rebuilt, nothing goes stale
    for j in range(10):
        x = torch.from_numpy(_scalars(b"committee/y/%d" % j, 1)).cuda()
        X = bls.g2_commit(x)[0][0].contiguous()
        ok, st = bls.batch_verify_g1_same_key(X, msgs[:64].contiguous(), bls.g1_batch_mul(x.repeat(64, 1), Hm[:64].contiguous())[0])
        assert ok.all().item() and not st.any().item()
    ok, st = bls.batch_verify_g1_same_key(keys[1], msgs, sigs[1])
    assert int(ok.sum().item()) == n - 1 and not st.any().item()
