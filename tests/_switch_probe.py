"""One compact oracle-checked workload per environment switch of libkyberhip.so (the switches are read once per
process, so tests/test_gpu_switches.py runs this file in a subprocess per value).  Prints "switch-probe ok <what>".

  fb   : same-base batches (fixed_base.cuh) -- KYB_FB_MIN
  msm  : Pippenger pipeline tail (msm.cuh)  -- KYB_MSM_TAIL, KYB_MSM_SUB
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


if __name__ == "__main__":
    {"fb": fb, "msm": msm, "lvm": lvm, "bncheck": bncheck}[sys.argv[1]]()
    print("switch-probe ok", sys.argv[1])
