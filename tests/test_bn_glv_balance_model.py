"""The balanced GLV halves of the BN G1 MSM adapter (kyber_amd/csrc/bn_msm.inc G1MsmGlv::decode_split) as an integer model,
with the constants of the built headers (bn256_params.h / bn254_params.h): bn_suite.inc's glv_split (rounded DOWN, with
truncated reciprocals) followed by the adapter's moves along the lattice vectors v1 = (A1, -|B1|), v2 = (A2, B2).  What the
MSM needs from it, for EVERY 32-byte scalar: k = k1 + k2 lambda (mod n), both halves below 2^127 with a top 16-bit window
below 2^15 (no ninth window), within the six steps per coordinate the kernel makes.  No GPU: host logic."""
import os
import random
import re

import pytest

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "kyber_amd", "csrc")


def _consts(header):
    text = open(os.path.join(CSRC, header)).read()

    def arr(name):
        m = re.search(r"static constexpr uint32_t %s\[\d+\] = \{([^}]*)\}" % name, text)
        words = [int(w.strip().rstrip("u"), 16) for w in m.group(1).split(",")]
        return sum(w << (32 * i) for i, w in enumerate(words))

    return {k: arr(k) for k in ("ORDER", "GLV_A1", "GLV_B1N", "GLV_A2", "GLV_B2", "GLV_G1", "GLV_G2")}


def split_balanced(c, k):
    n, A1, B1, A2, B2 = c["ORDER"], c["GLV_A1"], c["GLV_B1N"], c["GLV_A2"], c["GLV_B2"]
    c1, c2 = (k * c["GLV_G1"]) >> 256, (k * c["GLV_G2"]) >> 256  # glv_split: floor, truncated reciprocals
    k1, k2 = k - c1 * A1 - c2 * A2, c1 * B1 - c2 * B2
    assert abs(k1) < 1 << 130 and abs(k2) < 1 << 130  # the bound bn_suite.inc states (five words, two's complement)
    steps2 = steps1 = 0
    for _ in range(6):
        if k2 > B1 >> 1:
            k2, k1, steps2 = k2 - B1, k1 + A1, steps2 + 1
        elif k2 < -(B1 >> 1):
            k2, k1, steps2 = k2 + B1, k1 - A1, steps2 + 1
    for _ in range(6):
        if k1 > A2 >> 1:
            k1, k2, steps1 = k1 - A2, k2 - B2, steps1 + 1
        elif k1 < -(A2 >> 1):
            k1, k2, steps1 = k1 + A2, k2 + B2, steps1 + 1
    return k1, k2, steps1, steps2


@pytest.mark.parametrize("header", ["bn256_params.h", "bn254_params.h"])
def test_balanced_halves_fit_eight_windows_for_every_scalar(header):
    c = _consts(header)
    n, A1, B1, A2, B2 = c["ORDER"], c["GLV_A1"], c["GLV_B1N"], c["GLV_A2"], c["GLV_B2"]
    lam = A1 * pow(B1, -1, n) % n  # v1 = (A1, -|B1|) is in the lattice: A1 - |B1| lambda = 0 (mod n)
    assert (A2 + B2 * lam) % n == 0  # and so is v2 = (A2, B2)
    assert (lam * lam + lam + 1) % n == 0  # lambda is a primitive cube root of unity mod n: the eigenvalue of (beta x, y)
    rng = random.Random(2024)
    ks = [0, 1, 2, n - 1, n, n + 1, 2 * n, (1 << 256) - 1, (1 << 255), (1 << 255) - 1, (1 << 254) + 12345, lam, n - lam, B1, A2]
    ks += [rng.randrange(1 << 256) for _ in range(20000)] + [rng.randrange(n) for _ in range(20000)]
    ks += [(1 << 256) - 1 - rng.randrange(1 << 40) for _ in range(2000)] + [rng.randrange(1 << 130) for _ in range(2000)]
    worst = 0
    for k in ks:
        k1, k2, s1, s2 = split_balanced(c, k)
        assert (k1 + k2 * lam - k) % n == 0
        assert abs(k1) <= (A2 >> 1) and abs(k2) <= (B1 >> 1) + 6 * B2  # k2 moves by B2 ~ 2^64 per step of the second loop
        for h in (abs(k1), abs(k2)):
            assert h < 1 << 127 and (h >> 112) < 1 << 15  # the top window never reaches 2^15: the recoding does not carry
        assert s1 <= 5 and s2 <= 5  # the kernel makes six
        worst = max(worst, s1, s2)
    assert worst >= 1  # the correction is exercised


def test_bls12381_g2_quarters_fit_four_windows_for_every_scalar():
    """bls12381_msm.hip BlsG2MsmGls::decode_split: k = a0 + a1 |z| + a2 |z|^2 + a3 |z|^3 by long division, the quarters moved
    into (-|z| / 2, |z| / 2] with carries, the overflow of a3 folded back through |z|^4 = z^2 - 1 (mod r)"""
    Z = 0xd201000000010000
    r = Z**4 - Z**2 + 1
    assert r == 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001  # kilic/scalar.go:11-12
    H = Z >> 1
    rng = random.Random(381)
    ks = [0, 1, r - 1, r, r + 1, (1 << 256) - 1, 1 << 255, Z, Z - 1, Z * Z, Z**3, Z**3 - 1, H, H + 1, Z**3 * 2 + H * Z * Z + H * Z + H + 1]
    ks += [rng.randrange(1 << 256) for _ in range(20000)] + [rng.randrange(r) for _ in range(20000)]
    ks += [(1 << 256) - 1 - rng.randrange(1 << 64) for _ in range(2000)]
    for k in ks:
        q1, a0 = divmod(k, Z)
        q2, a1 = divmod(q1, Z)
        a3, a2 = divmod(q2, Z)
        assert a3 < 1 << 65
        A = [a0, a1, a2, a3]
        for i in range(3):
            if A[i] > H:
                A[i] -= Z
                A[i + 1] += 1
        a4 = 0
        for _ in range(4):
            if A[3] > H:
                A[3] -= Z
                a4 += 1
        assert A[3] <= H and a4 <= 3
        A[2] += a4
        A[0] -= a4
        assert (sum(a * Z**i for i, a in enumerate(A)) - k) % r == 0
        for a in A:
            assert abs(a) < 1 << 63 and (abs(a) >> 48) < 1 << 15  # four 16-bit windows, the top one never reaches 2^15
