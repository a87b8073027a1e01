"""One compact oracle-checked workload per environment switch of libkyberhip.so (the switches are read once per
process, so tests/test_gpu_switches.py runs this file in a subprocess per value).  Prints "switch-probe ok <what>".

  fb   : same-base batches (fixed_base.cuh) -- KYB_FB_MIN
  msm  : Pippenger pipeline tail (msm.cuh)  -- KYB_MSM_TAIL, KYB_MSM_SUB
  pipe : Ed25519 host-buffer batches cut in chunks over the page-locked slots (ed25519.hip mul_host) -- KYB_PIPE_CHUNK,
         KYB_PIPE_STREAMS
  g1split : BLS12-381 G1 Mul of a half-empty chip, test and multiplication in different workgroups -- KYB_G1_SPLIT
  unmw2 : BLS12-381 UnmarshalBinary of large batches through the two-wave kernels -- KYB_UNM_W2
  lvm  : G1 / G2 Mul dispatch (bls12381_lvm.cuh) -- KYB_LVM_MIN, KYB_G1_COOP_MAX (the small-batch kernel on cooperating lanes)
"""
import os
import random
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _be(ks):
    return np.frombuffer(b"".join(k.to_bytes(32, "big") for k in ks), dtype=np.uint8).reshape(len(ks), 32).copy()


def fb():
    import torch

    from kyber_amd.pairing import bls12381 as B, bn256 as N
    from oracle import bls12381 as OB, bn256 as ON

    rng = random.Random(5)
    for m, O, enc1, enc2 in ((B, OB, OB.g1_compress, OB.g2_compress), (N, ON, ON.g1_marshal, ON.g2_marshal)):
        for grp in (1, 2):
            n = 1 << 17
            h = rng.randrange(1, m.ORDER)
            base_pt = (O.g1_mul if grp == 1 else O.g2_mul)(h, O.G1_GEN if grp == 1 else O.G2_GEN)
            base = (enc1 if grp == 1 else enc2)(base_pt)
            edge = [0, 1, 511, 512, 513, 1023, 1024, m.ORDER - 1, m.ORDER, (1 << 256) - 1, (511 << 246) | 512]
            where = list(range(len(edge))) + [n - 1] + list(range(4099, n - 1, n // 56))  # first, last, strided: >= 64 lanes
            ks = edge + [rng.randrange(1 << 256) for _ in range(len(where) - len(edge))]
            assert len(where) >= 64
            s = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda")
            s[where] = torch.from_numpy(_be(ks)).cuda()
            b = torch.from_numpy(np.frombuffer(base, dtype=np.uint8).copy()).cuda()
            commit = m.g1_commit if grp == 1 else m.g2_commit
            for _ in range(2):  # built, then reused
                out, st = commit(s, b)
                torch.cuda.synchronize()
                assert not st.any().item()
                got = out[where].cpu().numpy()
                for j, k in enumerate(ks):
                    want = (enc1 if grp == 1 else enc2)((O.g1_mul if grp == 1 else O.g2_mul)(k % m.ORDER, base_pt))
                    assert bytes(got[j]) == want, (m.__name__, grp, hex(k))


def msm():
    import torch  # noqa: F401

    from kyber_amd.pairing import bls12381 as B, bn256 as N
    from oracle import bls12381 as OB, bn256 as ON

    rng = random.Random(6)
    for m, O, enc1, enc2 in ((B, OB, OB.g1_compress, OB.g2_compress), (N, ON, ON.g1_marshal, ON.g2_marshal)):
        for grp, n in ((1, 5000), (2, 700), (1, 3)):
            hs = [rng.randrange(1, m.ORDER) for _ in range(n)]
            ks = [rng.randrange(1 << 256) for _ in range(n)]
            ks[0], ks[1] = 0, m.ORDER - 1
            ks[2:2 + min(n - 2, 300)] = [ks[2]] * min(n - 2, 300)  # a long bucket (skewed digits)
            commit = m.g1_commit if grp == 1 else m.g2_commit
            pts, st = commit(_be(hs))
            assert not np.asarray(st).any()
            out, st = (m.g1_msm if grp == 1 else m.g2_msm)(_be(ks), pts)
            assert not np.asarray(st).any()
            tot = sum(k * h for k, h in zip(ks, hs)) % m.ORDER
            gen = O.G1_GEN if grp == 1 else O.G2_GEN
            want = (enc1 if grp == 1 else enc2)((O.g1_mul if grp == 1 else O.g2_mul)(tot, gen))
            assert bytes(np.asarray(out)) == want, (m.__name__, grp, n)


def msmbig():
    """2^17 BLS12-381 G1 points: a plan of 2^14 buckets per window (the two-pass sort, the split tail with fused tree levels,
    the light decode kernel), with a block of 30 000 equal scalars (a coarse bin too long for the second pass's LDS, a long
    bucket), zero scalars (empty top windows for some points) and scalars >= r"""
    import torch  # noqa: F401

    from kyber_amd.pairing import bls12381 as B
    from oracle import bls12381 as OB

    rng = random.Random(66)
    n = 1 << 17
    hs = [rng.randrange(1, B.ORDER) for _ in range(n)]
    ks = [rng.randrange(1 << 256) for _ in range(n)]
    ks[0], ks[1], ks[2] = 0, B.ORDER - 1, B.ORDER + 5
    ks[100:30100] = [ks[100]] * 30000
    ks[40000:40100] = [rng.randrange(1 << 40) for _ in range(100)]
    pts, st = B.g1_commit(_be(hs))
    assert not np.asarray(st).any()
    tot = sum(k * h for k, h in zip(ks, hs)) % B.ORDER
    want = OB.g1_compress(OB.g1_mul(tot, OB.G1_GEN))
    out, st = B.g1_msm(_be(ks), pts)
    assert not np.asarray(st).any() and bytes(np.asarray(out)) == want
    unc, st = B._mul(1, torch.from_numpy(_be(hs)).cuda(), torch.from_numpy(np.frombuffer(B.G1_BASE, dtype=np.uint8).copy()).cuda(), True,
                     B.F_UNCOMPRESSED_OUT)
    assert not st.any().item()
    out, st = B.g1_msm(torch.from_numpy(_be(ks)).cuda(), unc, B.F_TRUSTED(0) | B.F_UNCOMPRESSED)
    assert not st.any().item() and bytes(out.cpu().numpy()) == want


def msmgiant():
    """2^19 BLS12-381 G1 points with GIANT buckets (msm.cuh giant_*_kernel, bucket_long_coop*_kernel): half of the scalars
    equal (a bucket of 2^18 entries in every window of both GLV halves), and the same points under 128-bit coefficients
    with KYB_F_SCALAR_BITS(128) (the plain adapter: its ninth window holds only the recoding's carry)"""
    import torch

    from kyber_amd.pairing import bls12381 as B
    from oracle import bls12381 as OB

    rng = random.Random(67)
    n = 1 << 19
    hs = [rng.randrange(1, B.ORDER) for _ in range(n)]
    ks = [rng.randrange(1 << 256) for _ in range(n)]
    ks[1000:1000 + (n >> 1)] = [ks[1000]] * (n >> 1)
    g1b = torch.from_numpy(np.frombuffer(B.G1_BASE, dtype=np.uint8).copy()).cuda()
    unc, st = B._mul(1, torch.from_numpy(_be(hs)).cuda(), g1b, True, B.F_UNCOMPRESSED_OUT)
    assert not st.any().item()
    fl = B.F_TRUSTED(0) | B.F_UNCOMPRESSED
    tot = sum(k * h for k, h in zip(ks, hs)) % B.ORDER
    out, st = B.g1_msm(torch.from_numpy(_be(ks)).cuda(), unc, fl)
    assert not st.any().item() and bytes(out.cpu().numpy()) == OB.g1_compress(OB.g1_mul(tot, OB.G1_GEN))
    k128 = [k & ((1 << 128) - 1) for k in ks]
    tot = sum(k * h for k, h in zip(k128, hs)) % B.ORDER
    out, st = B.g1_msm(torch.from_numpy(_be(k128)).cuda(), unc, fl | B.F_SCALAR_BITS(128))
    assert not st.any().item() and bytes(out.cpu().numpy()) == OB.g1_compress(OB.g1_mul(tot, OB.G1_GEN))


def msmg2short():
    """BLS12-381 G2 MSM under KYB_F_SCALAR_BITS(128) on both adapters (KYB_BLS_G2_MSM_GLS: 0 plain windows, 1 quarters for
    full-length scalars only, 2 quarters always; default: quarters up to 2^15 points): 900 points with a long bucket, junk
    above bit 128 ignored, against the oracle's multiplication of the generator"""
    import torch  # noqa: F401

    from kyber_amd.pairing import bls12381 as B
    from oracle import bls12381 as OB

    rng = random.Random(68)
    n = 900
    hs = [rng.randrange(1, B.ORDER) for _ in range(n)]
    ks = [rng.randrange(1 << 128) for _ in range(n)]
    ks[0], ks[1] = 0, (1 << 128) - 1
    ks[10:310] = [ks[10]] * 300
    pts, st = B.g2_commit(_be(hs))
    assert not np.asarray(st).any()
    want = OB.g2_compress(OB.g2_mul(sum(k * h for k, h in zip(ks, hs)) % B.ORDER, OB.G2_GEN))
    junk = [k | (rng.randrange(1 << 120) << 130) for k in ks]
    for kk, fl in ((ks, 0), (ks, B.F_SCALAR_BITS(128)), (junk, B.F_SCALAR_BITS(128)), (ks, B.F_SCALAR_BITS(200))):
        out, st = B.g2_msm(_be(kk), pts, fl)
        assert not np.asarray(st).any() and bytes(np.asarray(out)) == want, fl


def lvm():
    import torch  # noqa: F401

    from kyber_amd.pairing import bls12381 as B
    from oracle import bls12381 as OB

    rng = random.Random(7)
    n = 2048
    for grp in (1, 2):
        hs = [rng.randrange(1, B.ORDER) for _ in range(n)]
        ks = [rng.randrange(1 << 256) for _ in range(n)]
        ks[:4] = [0, 1, B.ORDER, B.ORDER - 1]
        pts, _ = (B.g1_commit if grp == 1 else B.g2_commit)(_be(hs))
        out, st = (B.g1_batch_mul if grp == 1 else B.g2_batch_mul)(_be(ks), pts)
        assert not np.asarray(st).any()
        gen = OB.G1_GEN if grp == 1 else OB.G2_GEN
        mul, enc = (OB.g1_mul, OB.g1_compress) if grp == 1 else (OB.g2_mul, OB.g2_compress)
        for i in list(range(6)) + [777, n - 1]:
            assert bytes(np.asarray(out)[i]) == enc(mul(ks[i] * hs[i] % B.ORDER, gen)), (grp, i)
    _lvm_unmarshal()


def _lvm_unmarshal():
    """UnmarshalBinary through whichever kernel the switches select: members, a point outside the subgroup, an encoding
    the flag rules refuse, infinity -- both groups, compressed and uncompressed output"""
    from kyber_amd.pairing import bls12381 as B
    from oracle import bls12381 as OB

    rng = random.Random(9)
    x = 1
    while True:
        y = OB.fp_sqrt((x * x * x + 4) % OB.P)
        if y is not None and not OB.g1_in_subgroup((x, y)):
            off1 = OB.g1_compress((x, y))
            break
        x += 1
    x = 1
    while True:
        c = (x, 1)
        y = OB.f2_sqrt(OB.f2_add(OB.f2_mul(OB.f2_sqr(c), c), (4, 4)))
        if y is not None and not OB.g2_in_subgroup((c, y)):
            off2 = OB.g2_compress((c, y))
            break
        x += 1
    for grp, mul, gen, enc, unc, off, w in ((1, OB.g1_mul, OB.G1_GEN, OB.g1_compress, OB.g1_serialize_unc, off1, 48),
                                            (2, OB.g2_mul, OB.G2_GEN, OB.g2_compress, OB.g2_serialize_unc, off2, 96)):
        pts = [mul(rng.randrange(1, B.ORDER), gen) for _ in range(5)]
        wire = [enc(p) for p in pts] + [off, bytes(w), enc(None)]
        out, st = B.ENGINE.batch_unmarshal(grp, b"".join(wire))
        assert list(np.asarray(st)) == [0] * 5 + [2, 1, 0], (grp, list(np.asarray(st)))
        assert [bytes(r) for r in np.asarray(out)] == wire[:5] + [bytes(w), bytes(w), enc(None)]
        outu, st = B.ENGINE.batch_unmarshal(grp, b"".join(wire), B.F_UNCOMPRESSED_OUT)
        assert [bytes(r) for r in np.asarray(outu)[:5]] == [unc(p) for p in pts] and list(np.asarray(st)) == [0] * 5 + [2, 1, 0]
        outt, st = B.ENGINE.batch_unmarshal(grp, b"".join(wire), B.F_TRUSTED(0))   # vouched for: the subgroup rule is skipped
        assert list(np.asarray(st)) == [0] * 6 + [1, 0]


def g1split():
    """G1Elt.Mul, every operand re-validated, at a size between the cooperating-lane kernel and the lane machine: the test
    and the multiplication in different workgroups (bls12381_g1split.hip) -- members, a point outside the subgroup, a
    refused encoding, infinity; compressed and uncompressed input"""
    from kyber_amd.pairing import bls12381 as B
    from oracle import bls12381 as OB

    rng = random.Random(10)
    n = 20000 + 37
    x = 1
    while True:
        y = OB.fp_sqrt((x * x * x + 4) % OB.P)
        if y is not None and not OB.g1_in_subgroup((x, y)):
            off = (x, y)
            break
        x += 1
    hs = [rng.randrange(1, B.ORDER) for _ in range(n)]
    ks = [rng.randrange(1 << 256) for _ in range(n)]
    ks[:4] = [0, 1, B.ORDER, B.ORDER - 1]
    for unc in (False, True):
        fl = (B.F_UNCOMPRESSED | B.F_UNCOMPRESSED_OUT) if unc else 0
        w = 96 if unc else 48
        enc = OB.g1_serialize_unc if unc else OB.g1_compress
        pts, _ = B.g1_commit(_be(hs), None, B.F_UNCOMPRESSED_OUT if unc else 0)
        pts = np.asarray(pts).copy()
        bad = {63: (enc(off), 2), 64: (bytes(w), 1), 4099: (enc(None), 0), n - 1: (enc(off), 2), n - 2: (b"\xff" * w, 1)}
        for i, (wire, _) in bad.items():
            pts[i] = np.frombuffer(wire, dtype=np.uint8)
        out, st = B.g1_batch_mul(_be(ks), pts, fl)
        out, st = np.asarray(out), np.asarray(st)
        for i in list(range(6)) + [62, 65, 4098, 4100, 12345, n - 3]:
            assert st[i] == 0 and bytes(out[i]) == enc(OB.g1_mul(ks[i] * hs[i] % B.ORDER, OB.G1_GEN)), (unc, i)
        for i, (_, code) in bad.items():
            assert st[i] == code, (unc, i, st[i])
            assert bytes(out[i]) == (enc(None) if code == 0 else bytes(w)), (unc, i)
        assert int((st != 0).sum()) == 4


def unmw2():
    """UnmarshalBinary of batches large enough for the two-wave kernels (bls12381_unm2.hip): 2^17 + G1 points, 2^19 + G2
    points made by the engine's own Commit (compared with the oracle on a sample), with a point outside the subgroup, a
    refused encoding and infinity spliced in at the first, a middle and the last position"""
    from kyber_amd.pairing import bls12381 as B
    from oracle import bls12381 as OB

    rng = random.Random(14)
    x = 1
    while True:
        y = OB.fp_sqrt((x * x * x + 4) % OB.P)
        if y is not None and not OB.g1_in_subgroup((x, y)):
            off1 = OB.g1_compress((x, y))
            break
        x += 1
    x = 1
    while True:
        c = (x, 1)
        y = OB.f2_sqrt(OB.f2_add(OB.f2_mul(OB.f2_sqr(c), c), (4, 4)))
        if y is not None and not OB.g2_in_subgroup((c, y)):
            off2 = OB.g2_compress((c, y))
            break
        x += 1
    for grp, n, off, w, gen, mul, enc, unc in ((1, (1 << 17) + 77, off1, 48, OB.G1_GEN, OB.g1_mul, OB.g1_compress, OB.g1_serialize_unc),
                                               (2, (1 << 19) + 77, off2, 96, OB.G2_GEN, OB.g2_mul, OB.g2_compress, OB.g2_serialize_unc)):
        hs = np.frombuffer(rng.randbytes(32 * n), dtype=np.uint8).reshape(n, 32).copy()
        hs[:, 0] &= 0x3F
        pts, _ = (B.g1_commit if grp == 1 else B.g2_commit)(hs)
        pts = np.asarray(pts).copy()
        sample = [1, 2, 63, 64, n // 3, n - 2]
        for i in sample:
            assert bytes(pts[i]) == enc(mul(int.from_bytes(bytes(hs[i]), "big"), gen)), (grp, i)
        special = {0: (off, 2), n // 2: (bytes(w), 1), n - 1: (enc(None), 0), 4097: (off, 2)}
        for i, (wire, _) in special.items():
            pts[i] = np.frombuffer(wire, dtype=np.uint8)
        out, st = B.ENGINE.batch_unmarshal(grp, pts)
        out, st = np.asarray(out), np.asarray(st)
        assert int((st != 0).sum()) == 3
        for i, (wire, code) in special.items():
            assert st[i] == code and bytes(out[i]) == (wire if code == 0 else bytes(w)), (grp, i, st[i])
        keep = np.ones(n, dtype=bool)
        keep[list(special)] = False
        assert (out[keep] == pts[keep]).all()
        outu, st = B.ENGINE.batch_unmarshal(grp, pts, B.F_UNCOMPRESSED_OUT)
        outu = np.asarray(outu)
        for i in sample:
            assert bytes(outu[i]) == unc(mul(int.from_bytes(bytes(hs[i]), "big"), gen)), (grp, i)


def hashw2():
    """G1Elt.Hash / G2Elt.Hash of a batch large enough for the two-wave kernels (bls12381_unm2.hip): 2^17 + 5 messages,
    first / seam / last against the oracle's hash_to_curve"""
    import torch

    from kyber_amd.pairing import bls12381 as B
    from oracle import bls12381 as OB

    n = (1 << 17) + 5
    rng = random.Random(15)
    msgs = np.frombuffer(rng.randbytes(32 * n), dtype=np.uint8).reshape(n, 32).copy()
    m = torch.from_numpy(msgs).cuda()
    h1, s1 = B.batch_hash_g1(m)
    h2, s2 = B.batch_hash_g2(m)
    torch.cuda.synchronize()
    assert not s1.any().item() and not s2.any().item()
    h1, h2 = h1.cpu().numpy(), h2.cpu().numpy()
    for i in (0, 63, 64, 1 << 16, n - 1):
        assert bytes(h1[i]) == OB.g1_compress(OB.hash_to_g1(bytes(msgs[i]), B.DOMAIN_G1)), i
        assert bytes(h2[i]) == OB.g2_compress(OB.hash_to_g2(bytes(msgs[i]), B.DOMAIN_G2)), i


def pipe():
    from kyber_amd.group import edwards25519 as E
    from oracle import ed25519 as OE

    rng = random.Random(8)
    n = 5 * 4096 + 1000 + 37  # with KYB_PIPE_CHUNK=4096: five whole chunks and a ragged one; one resident call otherwise
    # (and above the 16 384 from which a same-base batch builds the base's table)
    s = np.frombuffer(rng.randbytes(32 * n), dtype=np.uint8).reshape(n, 32).copy()
    pts = np.empty((n, 32), dtype=np.uint8)
    seeds = [OE.mul_base(rng.randbytes(32)) for _ in range(16)]
    for i in range(n):
        pts[i] = np.frombuffer(seeds[i % 16], dtype=np.uint8)
    bad = [4095, 4096, 8191, n - 1]  # chunk seams and the last element
    y = 2
    while OE.decode(y.to_bytes(32, "little")) is not None:  # (kyber accepts non-canonical y: search for a non-square instead)
        y += 1
    for i in bad:
        pts[i] = np.frombuffer(y.to_bytes(32, "little"), dtype=np.uint8)  # no x for this y: UnmarshalBinary fails
    where = sorted(set([0, 1, 63, 64, 4094, 4097, 8190, 8192, 12287, 12288, 16383, 16384, 20479, 20480, n - 2] + list(range(5, n, n // 48)) + bad))
    out_b = E.batch_mul_base(s)
    out_v, st = E.batch_mul(s, pts)
    out_c = E.commit(s, seeds[3])
    for i in where:
        assert bytes(out_b[i]) == OE.mul_base(bytes(s[i])), ("base", i)
        assert bytes(out_c[i]) == OE.mul(bytes(s[i]), seeds[3]), ("same", i)
        if i in bad:
            assert st[i] != 0 and not out_v[i].any(), ("bad", i)
        else:
            assert st[i] == 0 and bytes(out_v[i]) == OE.mul(bytes(s[i]), bytes(pts[i])), ("var", i)
    assert int((st != 0).sum()) == len(bad)


def bncheck():
    """bn256 ValidatePairing -- product form + zero-Miller-value fallback (default) or the reference's two pairings
    (KYB_BN_CHECK=two): ordinary pairs and the degenerate ones a G2 point of order 13 makes"""
    from kyber_amd.pairing import bn256 as bn
    from oracle import bn256 as ON

    rng = random.Random(3)
    h = 2 * ON.P - ON.ORDER
    while True:
        x = (rng.randrange(ON.P), rng.randrange(ON.P))
        y = ON.f2_sqrt(ON.f2_add(ON.f2_mul(ON.f2_sqr(x), x), ON.TWIST_B))
        if y is not None:
            break
    Q13 = ON.g2_mul(ON.ORDER * h // 13, (x, y))
    p1, q1 = ON.g1_mul(9, ON.G1_GEN), ON.g2_mul(4, ON.G2_GEN)
    quads = [(p1, q1, ON.g1_mul(36, ON.G1_GEN), ON.G2_GEN), (p1, q1, ON.g1_mul(37, ON.G1_GEN), ON.G2_GEN),
             (p1, Q13, ON.g1_mul(2, ON.G1_GEN), ON.g2_mul(3, Q13)), (p1, Q13, p1, q1)] * 20
    ok, st = bn.batch_validate_pairing(b"".join(ON.g1_marshal(q[0]) for q in quads), b"".join(ON.g2_marshal(q[1]) for q in quads),
                                       b"".join(ON.g1_marshal(q[2]) for q in quads), b"".join(ON.g2_marshal(q[3]) for q in quads))
    assert not np.asarray(st).any()
    assert [bool(v) for v in np.asarray(ok)] == [True, False, True, False] * 20


def bnhash():
    """bn256 pointG1.Hash for a batch large enough for the queued kernel (default), the per-lane kernel
    (KYB_BN_HASH_QUEUE=0) or a forced number of messages per wave (KYB_BN_HASH_HQ): oracle lanes + a digest that must
    (computed once with oracle/bn256.py hash_to_g1 over all 131 081 messages) that holds whichever kernel ran"""
    import hashlib

    import torch

    from kyber_amd.pairing import bn256 as bn
    from oracle import bn256 as ON

    n = (1 << 17) + 9
    msgs = np.frombuffer(hashlib.shake_256(b"switch/bnhash").digest(n * 32), dtype=np.uint8).reshape(n, 32).copy()
    out, st = bn.batch_hash_g1(torch.from_numpy(msgs).cuda())
    assert not st.any().item()
    got = out.cpu().numpy()
    for i in [0, 1, n - 1] + list(range(777, n, n // 9)):
        assert bytes(got[i]) == ON.g1_marshal(ON.hash_to_g1(bytes(msgs[i]))), i
    assert hashlib.sha256(got.tobytes()).hexdigest()[:16] == BNHASH_DIGEST, hashlib.sha256(got.tobytes()).hexdigest()[:16]


BNHASH_DIGEST = "86f1a16dd7b32606"

if __name__ == "__main__":
    {"bnhash": bnhash, "fb": fb, "msm": msm, "msmbig": msmbig, "msmgiant": msmgiant, "msmg2short": msmg2short, "lvm": lvm, "bncheck": bncheck, "pipe": pipe, "g1split": g1split, "unmw2": unmw2, "hashw2": hashw2}[sys.argv[1]]()
    print("switch-probe ok", sys.argv[1])
