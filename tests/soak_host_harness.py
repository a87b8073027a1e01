"""Randomised soak of the device headers compiled for the host (tests/host_harness.cpp) against the Python oracles:
python tests/soak_host_harness.py [seed] (it lives under tests/ because it uses the oracles).  About 1 000 scalar multiplications over the five groups (edge scalars around
the group orders, every BLS12-381 flag combination), the Ed25519 window walk in both scalar semantics; prints the number of mismatches.  Not part of the test suite (the suite runs a fixed subset); run it after
touching mont.cuh / tower.cuh / curve.cuh / fe25519.cuh / ge25519.cuh.  No GPU needed."""
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import _host_harness as H
from oracle import bls12381 as B, bn256 as N, ed25519 as E
rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 2026)
t0 = time.time(); bad = 0
def rs(bits=256):
    k = rng.choice([rng.getrandbits(bits), rng.getrandbits(64), rng.getrandbits(128), B.R - rng.randrange(1, 5), B.R + rng.randrange(0, 5), (1 << 256) - 1 - rng.getrandbits(16), rng.getrandbits(255) | (1 << 255)])
    return k % (1 << 256)
# BLS G1 / G2
for i in range(250):
    k = rs(); h = rng.randrange(1, B.R)
    P = B.g1_mul(h, B.G1_GEN)
    flags = rng.choice([0, 2, 4, 6, 0x100, 0x102])
    inp = B.g1_serialize_unc(P) if flags & 2 else B.g1_compress(P)
    exp = B.g1_mul(k % B.R, P)
    want = B.g1_serialize_unc(exp) if flags & 4 else B.g1_compress(exp)
    st, out = H.call("hh_bls_g1_mul_f", k.to_bytes(32, "big"), inp, flags, out_sizes=(96 if flags & 4 else 48,))
    if (st, out) != (0, want): bad += 1; print("BLS G1 mismatch", hex(k), flags)
print("bls g1 done", round(time.time() - t0, 1), bad)
for i in range(80):
    k = rs(); h = rng.randrange(1, B.R)
    Q = B.g2_mul(h, B.G2_GEN)
    exp = B.g2_compress(B.g2_mul(k % B.R, Q))
    st, out = H.call("hh_bls_g2_mul_f", k.to_bytes(32, "big"), B.g2_compress(Q), 0, out_sizes=(96,))
    if (st, out) != (0, exp): bad += 1; print("BLS G2 mismatch", hex(k))
print("bls g2 done", round(time.time() - t0, 1), bad)
# bn256 G1 / G2 (scalars are plain 256-bit integers: k P for any k)
for i in range(250):
    k = rng.choice([rng.getrandbits(256), rng.getrandbits(254), N.ORDER - rng.randrange(1, 4), N.ORDER + rng.randrange(0, 4), rng.getrandbits(127), (1 << 256) - 1 - rng.getrandbits(8)])
    P = N.g1_mul(rng.randrange(1, N.ORDER), N.G1_GEN)
    st, out = H.call("hh_bn_g1_mul", k.to_bytes(32, "big"), N.g1_marshal(P), out_sizes=(64,))
    if (st, out) != (0, N.g1_marshal(N.g1_mul(k, P))): bad += 1; print("bn G1 mismatch", hex(k))
for i in range(60):
    k = rng.getrandbits(256)
    Q = N.g2_mul(rng.randrange(1, N.ORDER), N.G2_GEN)
    st, out = H.call("hh_bn_g2_mul", k.to_bytes(32, "big"), N.g2_marshal(Q), out_sizes=(128,))
    if (st, out) != (0, N.g2_marshal(N.g2_mul(k, Q))): bad += 1; print("bn G2 mismatch", hex(k))
print("bn done", round(time.time() - t0, 1), bad)
# Ed25519 walk
Bp = E.encode(E.B)
for i in range(400):
    s = rng.getrandbits(256).to_bytes(32, "little")
    p = E.mul((rng.getrandbits(252)).to_bytes(32, "little"), Bp, vartime=True)
    vt = i & 1
    exp = E.mul(s, p, vartime=bool(vt))
    st, out = H.call("hh_ed_mul", s, p, vt, out_sizes=(32,))
    if (st, out) != (0, exp): bad += 1; print("ed mismatch", s.hex(), vt)
print("ed done", round(time.time() - t0, 1), bad)
print("TOTAL BAD", bad, "time", round(time.time() - t0, 1))
