"""oracle/bls12381_pair_ref.c -- the C port of the BLS12-381 pairing that bench.py times as the CPU baseline of the
batched-pairing config -- held byte for byte against oracle/bls12381.py, whose GT bytes are pinned by the reference's
IBE interop vector (tests/test_oracle_bls12381.py): random pairs, multiples of one pair (bilinearity through the bytes),
points at infinity, and several threads."""
import random

import numpy as np

from oracle import bls12381 as O
from tests import _oracle_c as OC


def _pairs(rng, n):
    out = []
    for _ in range(n):
        out.append((O.g1_mul(rng.randrange(1, O.R), O.G1_GEN), O.g2_mul(rng.randrange(1, O.R), O.G2_GEN)))
    return out


def test_gt_bytes_match_the_python_oracle():
    rng = random.Random(21)
    pairs = _pairs(rng, 4) + [(None, O.G2_GEN), (O.G1_GEN, None), (None, None), (O.G1_GEN, O.G2_GEN)]
    g1 = b"".join(O.g1_serialize_unc(p) for p, _ in pairs)
    g2 = b"".join(O.g2_serialize_unc(q) for _, q in pairs)
    for threads in (1, 3):
        gt, st = OC.bls12381_pair(g1, g2, threads=threads)
        assert not st.any()
        for i, (p, q) in enumerate(pairs):
            assert bytes(gt[i]) == O.gt_to_bytes(O.pair(p, q)), (threads, i)


def test_bilinearity_through_the_bytes():
    rng = random.Random(22)
    a, b = rng.randrange(1, O.R), rng.randrange(1, O.R)
    P, Q = O.g1_mul(rng.randrange(1, O.R), O.G1_GEN), O.g2_mul(rng.randrange(1, O.R), O.G2_GEN)
    g1 = O.g1_serialize_unc(O.g1_mul(a, P)) + O.g1_serialize_unc(P) + O.g1_serialize_unc(O.g1_mul(a * b % O.R, P))
    g2 = O.g2_serialize_unc(O.g2_mul(b, Q)) + O.g2_serialize_unc(O.g2_mul(a * b % O.R, Q)) + O.g2_serialize_unc(Q)
    gt, st = OC.bls12381_pair(g1, g2, threads=1)
    assert not st.any() and bytes(gt[0]) == bytes(gt[1]) == bytes(gt[2])
    one, _ = OC.bls12381_pair(O.g1_serialize_unc(None), O.g2_serialize_unc(Q), threads=1)
    assert bytes(gt[0]) != bytes(one[0])


def test_rejects_out_of_range_coordinates():
    bad = bytearray(O.g1_serialize_unc(O.G1_GEN))
    bad[0:48] = (O.P + 1).to_bytes(48, "big")
    bad[0] &= 0x1F
    gt, st = OC.bls12381_pair(bytes(bad), O.g2_serialize_unc(O.G2_GEN), threads=1)
    assert st[0] == 1 and not gt.any()


def _edge_scalars(rng, n):
    ks = [0, 1, 2, O.R - 1, O.R, O.R + 1, (1 << 256) - 1, 1 << 255, (1 << 128) - 1] + [rng.randrange(1 << 256) for _ in range(n)]
    return ks


def test_g1_mul_bytes_match_the_python_oracle():
    """ora_bls12381_g1_mul (compressed in, compressed out) against oracle/bls12381.py: edge scalars as plain 256-bit
    integers, the point at infinity, both signs of y, several threads"""
    rng = random.Random(31)
    ks = _edge_scalars(rng, 6)
    pts = [O.g1_mul(rng.randrange(1, O.R), O.G1_GEN) for _ in ks]
    pts[3] = None
    pts[4] = O.g1_neg(pts[5])
    k = b"".join(x.to_bytes(32, "big") for x in ks)
    p = b"".join(O.g1_compress(x) for x in pts)
    for threads in (1, 4):
        out, st = OC.bls12381_g1_mul(k, p, threads=threads)
        assert not st.any()
        for i, (x, pt) in enumerate(zip(ks, pts)):
            assert bytes(out[i]) == O.g1_compress(O.g1_mul(x, pt)), (threads, i)


def test_g2_mul_bytes_match_the_python_oracle():
    rng = random.Random(32)
    ks = _edge_scalars(rng, 4)
    pts = [O.g2_mul(rng.randrange(1, O.R), O.G2_GEN) for _ in ks]
    pts[2] = None
    pts[4] = O.g2_neg(pts[5])
    k = b"".join(x.to_bytes(32, "big") for x in ks)
    p = b"".join(O.g2_compress(x) for x in pts)
    for threads in (1, 4):
        out, st = OC.bls12381_g2_mul(k, p, threads=threads)
        assert not st.any()
        for i, (x, pt) in enumerate(zip(ks, pts)):
            assert bytes(out[i]) == O.g2_compress(O.g2_mul(x, pt)), (threads, i)


def test_mul_decompression_follows_the_flag_rules():
    """the 16 + 18 ZCash fixtures the reference holds (tests/golden/bls12381_zcash.json): every encoding the Python oracle
    rejects for a reason other than the subgroup is rejected here, every accepted one multiplies by 1 to itself"""
    import json
    import os

    fx = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "bls12381_zcash.json")))
    for group, fn, dec, n in (("G1", OC.bls12381_g1_mul, O.g1_decompress, 48), ("G2", OC.bls12381_g2_mul, O.g2_decompress, 96)):
        for case in fx[group]:
            buf = bytes.fromhex(case["hex"])
            if len(buf) != n:
                continue
            try:
                dec(buf, subgroup_check=False)
                good = True
            except O.DecodeError:
                good = False
            out, st = fn((1).to_bytes(32, "big"), buf, threads=1)
            assert (st[0] == 0) == good, (group, case.get("name"))
            if good:
                assert bytes(out[0]) == buf


def test_compressed_pair_and_sum_match_the_uncompressed_entry_points():
    rng = random.Random(33)
    pairs = _pairs(rng, 3) + [(None, O.G2_GEN), (O.G1_GEN, None)]
    gt_u, _ = OC.bls12381_pair(b"".join(O.g1_serialize_unc(p) for p, _ in pairs), b"".join(O.g2_serialize_unc(q) for _, q in pairs), threads=2)
    gt_c, st = OC.bls12381_pair_compressed(b"".join(O.g1_compress(p) for p, _ in pairs), b"".join(O.g2_compress(q) for _, q in pairs), threads=2)
    assert not st.any() and (gt_u == gt_c).all()
    ks = [rng.randrange(1 << 256) for _ in range(7)]
    pts = [O.g1_mul(rng.randrange(1, O.R), O.G1_GEN) for _ in ks]
    pts[2] = None
    k = np.frombuffer(b"".join(x.to_bytes(32, "big") for x in ks), dtype=np.uint8)
    acc = None
    for x, p in zip(ks, pts):
        acc = O.g1_add(acc, O.g1_mul(x, p))
    for threads in (1, 3):
        out, st = OC.bls12381_g1_mul_sum_compressed(k, np.frombuffer(b"".join(O.g1_compress(p) for p in pts), dtype=np.uint8), threads=threads)
        assert not st.any() and bytes(out) == O.g1_compress(acc)
