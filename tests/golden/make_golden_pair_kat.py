#!/usr/bin/env python3
"""Known answers of the CPU oracles (oracle/bls12381.py, oracle/bn256.py, oracle/bn254.py -- themselves pinned by the reference-held
vectors) on SHAKE-derived operands, so that the GPU tests can compare pairing outputs BYTE FOR
BYTE inside config-size batches instead of checking the engine against itself.  The shape
follows the reference's own pairing test (pairing/bn256/suite_test.go:231-259: e(aP, bQ) for random a, b).

Writes tests/golden/{bls12381,bn256,bn254}_pair_kat.npz (numpy, uint8 arrays):
  g1 (n, G1), g2 (n, G2)       operands  P_i = a_i G1, Q_i = b_i G2 in the suite's wire format; entries 0..2 hold the
                               point at infinity (G1, G2, both); bn256 entries 3..6 hold on-curve G2 points OUTSIDE the
                               order-n subgroup, which pairing/bn256 accepts (point.go:466-499)
  gt (n, GT)                   Suite.Pair(P_i, Q_i).MarshalBinary()
  chk_idx (m, 2), chk_ok (m,)  ValidatePairing(P_i, Q_i, P_j, Q_j) for index pairs (i, j): constructed equal
                               (a_j = a_i c, b_j = b_i / c), unrelated, and infinity cases
  k (n, 32), g1k (n, G1), g2k (n, G2)   scalar-mul answers  k_i P_i, k_i Q_i  (k: edge values, then random 256-bit
                               integers -- not reduced mod the order, like the engine's scalar input)
Run anywhere (needs only oracle/); about 3 minutes of CPU.
"""
import hashlib
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import bls12381 as OB  # noqa: E402
from oracle import bn254 as ON4  # noqa: E402
from oracle import bn256 as ON  # noqa: E402

N = 384
NCHK = 192


def shake_ints(label: bytes, n: int, mod: int | None = None, nbytes: int = 64):
    raw = hashlib.shake_256(label).digest(n * nbytes)
    out = [int.from_bytes(raw[nbytes * i:nbytes * (i + 1)], "big") for i in range(n)]
    return [x % mod for x in out] if mod else out


def rows(bs, w):
    return np.frombuffer(b"".join(bs), dtype=np.uint8).reshape(len(bs), w).copy()


def bn_off_subgroup_g2(seed: int):
    """An on-curve point of the twist that is not in the order-n subgroup (the cofactor is 2p - n: almost every point)."""
    xs = shake_ints(b"kyber-amd/kat/bn256/offsub/%d" % seed, 64, ON.P)
    for i in range(0, 64, 2):
        x = (xs[i], xs[i + 1])
        y = ON.f2_sqrt(ON.f2_add(ON.f2_mul(ON.f2_sqr(x), x), ON.TWIST_B))
        if y is not None and ON.g2_mul(ON.ORDER, (x, y)) is not None:
            return (x, y)
    raise RuntimeError("no off-subgroup point found")


def build(name, O, order, enc1, enc2, inf1, inf2, pair_bytes, check, mul1, mul2, g1w, g2w, gtw):
    t0 = time.time()
    a = shake_ints(b"kyber-amd/kat/%s/a" % name, N, order)
    b = shake_ints(b"kyber-amd/kat/%s/b" % name, N, order)
    a = [x or 1 for x in a]
    b = [x or 1 for x in b]
    a[7], b[7] = 1, 1
    a[8], b[8] = order - 1, order - 1
    # second half: pairs (i, N/2 + i) with equal products  a_j b_j = a_i b_i  (used by the pairing-check answers)
    c = shake_ints(b"kyber-amd/kat/%s/c" % name, N // 2, order)
    for i in range(16, N // 2):
        ci = c[i] or 1
        a[N // 2 + i] = a[i] * ci % order
        b[N // 2 + i] = b[i] * pow(ci, -1, order) % order
    P = [O.g1_mul(x, O.G1_GEN) for x in a]
    Q = [O.g2_mul(x, O.G2_GEN) for x in b]
    g1 = [enc1(p) for p in P]
    g2 = [enc2(q) for q in Q]
    g1[0], g2[1], g1[2], g2[2] = inf1, inf2, inf1, inf2
    if name == b"bn256":
        for j in range(3, 7):
            g2[j] = enc2(bn_off_subgroup_g2(j))
    gt = [pair_bytes(g1[i], g2[i]) for i in range(N)]
    # pairing checks: (i, N/2 + i) equal by construction for i >= 16; (i, i + 1) unrelated; infinity rows against
    # each other (1 == 1) and against a finite pair
    idx = [(i, N // 2 + i) for i in range(16, 16 + NCHK // 2)] + [(i, i + 1) for i in range(20, 20 + NCHK // 2 - 6)]
    idx += [(0, 1), (1, 2), (0, 9), (9, 9), (7, 8), (8, 7)]
    # ValidatePairing(p1, p2, inv1, inv2) is e(p1, p2) == e(inv1, inv2) (pairing/pairing.go:13-15): read off the GT
    # answers; the oracle's own check function is evaluated on a sample as a cross-check
    ok = [1 if gt[i] == gt[j] else 0 for i, j in idx]
    for t in (0, 1, NCHK // 2, NCHK // 2 + 1, len(idx) - 6, len(idx) - 4, len(idx) - 2):
        i, j = idx[t]
        assert bool(ok[t]) == bool(check(g1[i], g2[i], g1[j], g2[j])), (t, i, j)
    assert all(ok[:NCHK // 2]) and ok[NCHK // 2] == 0 and ok[-6] == 1 and ok[-4] == 0 and ok[-3] == 1
    # scalar multiplication: plain 256-bit integers (edge values first)
    ks = [0, 1, 2, 15, 16, 17, order - 1, order, order + 1, (1 << 255) - 19, (1 << 256) - 1, 1 << 128, (1 << 128) - 1]
    ks += shake_ints(b"kyber-amd/kat/%s/k" % name, N - len(ks), None, 32)
    kb = [k.to_bytes(32, "big") for k in ks]
    # (bn256's off-subgroup G2 points multiply like any other point in the reference, twist.go:162: they stay in)
    g1k = [mul1(kb[i], g1[i]) for i in range(N)]
    g2k = [mul2(kb[i], g2[i]) for i in range(N)]
    out = os.path.join(HERE, "%s_pair_kat.npz" % name.decode())
    np.savez_compressed(out, g1=rows(g1, g1w), g2=rows(g2, g2w), gt=rows(gt, gtw), chk_idx=np.array(idx, dtype=np.int32),
                        chk_ok=np.array(ok, dtype=np.uint8), k=rows(kb, 32), g1k=rows(g1k, g1w), g2k=rows(g2k, g2w))
    print("wrote", out, os.path.getsize(out), "bytes in %.0f s" % (time.time() - t0))


def main():
    which = set(sys.argv[1:]) or {"bls12381", "bn256", "bn254"}  # python make_golden_pair_kat.py [suite ...]

    def bls_check(p1, q1, p2, q2):
        return OB.pair_check(OB.g1_decompress(p1), OB.g2_decompress(q1), OB.g1_decompress(p2), OB.g2_decompress(q2))

    if "bls12381" in which:
        build(b"bls12381", OB, OB.R, OB.g1_compress, OB.g2_compress, OB.g1_compress(None), OB.g2_compress(None),
              OB.pair_bytes, bls_check, OB.g1_mul_bytes, OB.g2_mul_bytes, 48, 96, 576)

    def bn_check(p1, q1, p2, q2):
        return ON.validate_pairing(ON.g1_unmarshal(p1), ON.g2_unmarshal(q1), ON.g1_unmarshal(p2), ON.g2_unmarshal(q2))

    if "bn256" in which:
        build(b"bn256", ON, ON.ORDER, ON.g1_marshal, ON.g2_marshal, ON.g1_marshal(None), ON.g2_marshal(None),
              ON.pair_bytes, bn_check, ON.g1_mul_bytes, ON.g2_mul_bytes, 64, 128, 384)

    def bn4_check(p1, q1, p2, q2):
        return ON4.validate_pairing(ON4.g1_unmarshal(p1), ON4.g2_unmarshal(q1), ON4.g1_unmarshal(p2), ON4.g2_unmarshal(q2))

    if "bn254" in which:  # every G2 operand is in the subgroup: pairing/bn254 rejects the others (twist.go:47-66)
        build(b"bn254", ON4, ON4.ORDER, ON4.g1_marshal, ON4.g2_marshal, ON4.g1_marshal(None), ON4.g2_marshal(None),
              ON4.pair_bytes, bn4_check, ON4.g1_mul_bytes, ON4.g2_mul_bytes, 64, 128, 384)


if __name__ == "__main__":
    main()
