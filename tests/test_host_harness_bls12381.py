"""The device arithmetic headers (mont.cuh / tower.cuh / curve.cuh / bls12381.cuh), compiled for the
host, against the big-integer oracle.  This is the GPU-less debugging loop for the code each lane of
the HIP kernels runs; the GPU parity tests proper are tests/test_gpu_bls12381.py."""
import ctypes as C
import json
import os
import random

import pytest

from oracle import bls12381 as O
from tests import _host_harness as H


def _fp(x):
    return x.to_bytes(48, "big")


def test_fp_ops_random_and_edges():
    rng = random.Random(1)
    vals = [0, 1, 2, O.P - 1, O.P - 2, (1 << 380) - 1, 1 << 379] + [rng.randrange(O.P) for _ in range(200)]
    for i in range(0, len(vals) - 1):
        a, b = vals[i], vals[(i * 7 + 3) % len(vals)]
        assert H.call("hh_bls_fp_op", 0, _fp(a), _fp(b), out_sizes=(48,))[1] == _fp(a * b % O.P)
        assert H.call("hh_bls_fp_op", 1, _fp(a), _fp(b), out_sizes=(48,))[1] == _fp((a + b) % O.P)
        assert H.call("hh_bls_fp_op", 2, _fp(a), _fp(b), out_sizes=(48,))[1] == _fp((a - b) % O.P)
        assert H.call("hh_bls_fp_op", 3, _fp(a), _fp(b), out_sizes=(48,))[1] == _fp(-a % O.P)
    for a in vals + [3, 1 << 32, (1 << 64) - 1, O.P - 3, (O.P + 1) // 2]:  # Kaliski inversion: every step count / edge
        exp = pow(a, -1, O.P) if a else 0
        assert H.call("hh_bls_fp_op", 4, _fp(a), _fp(0), out_sizes=(48,))[1] == _fp(exp)


def test_zcash_fixtures(golden_dir):
    d = json.load(open(os.path.join(golden_dir, "bls12381_zcash.json")))
    for grp, fn, size in (("G1", "hh_bls_g1_decode", 48), ("G2", "hh_bls_g2_decode", 96)):
        for e in d[grp]:
            buf = bytes.fromhex(e["hex"])
            if len(buf) != size:
                continue  # wrong-length cases are rejected by the host wrapper, not the kernel
            st = H.call(fn, buf, 1)[0]
            assert (st == 0) == e["valid"], (grp, e["name"], st)


def test_decode_encode_roundtrip_and_signs():
    rng = random.Random(2)
    for _ in range(6):
        k = rng.randrange(1, O.R)
        for neg in (False, True):
            p1 = O.g1_mul(k, O.G1_GEN)
            p2 = O.g2_mul(k, O.G2_GEN)
            if neg:
                p1, p2 = O.g1_neg(p1), O.g2_neg(p2)
            b1, b2 = O.g1_compress(p1), O.g2_compress(p2)
            assert H.call("hh_bls_g1_recode", b1, out_sizes=(48,)) == (0, b1)
            assert H.call("hh_bls_g2_recode", b2, out_sizes=(96,)) == (0, b2)
    inf1, inf2 = O.g1_compress(None), O.g2_compress(None)
    assert H.call("hh_bls_g1_recode", inf1, out_sizes=(48,)) == (0, inf1)
    assert H.call("hh_bls_g2_recode", inf2, out_sizes=(96,)) == (0, inf2)


def test_subgroup_check_rejects_cofactor_points():
    # points on the curve but outside the r-torsion: the fast endomorphism checks must agree with [r]P = inf
    rng = random.Random(3)
    found1 = found2 = 0
    while found1 < 3:
        x = rng.randrange(O.P)
        y = O.fp_sqrt(x**3 + 4)
        if y is None:
            continue
        pt = (x, y)
        if O.g1_in_subgroup(pt):
            continue
        found1 += 1
        assert H.call("hh_bls_g1_decode", O.g1_compress(pt), 1)[0] == 2
        assert H.call("hh_bls_g1_decode", O.g1_compress(pt), 0)[0] == 0
        # clearing the cofactor lands in G1
        assert H.call("hh_bls_g1_decode", O.g1_compress(O.g1_mul(O.H1, pt)), 1)[0] == 0
    while found2 < 2:
        x = (rng.randrange(O.P), rng.randrange(O.P))
        y = O.f2_sqrt(O.f2_add(O.f2_mul(O.f2_sqr(x), x), (4, 4)))
        if y is None:
            continue
        found2 += 1
        pt = (x, y)
        assert not O.g2_in_subgroup(pt)
        assert H.call("hh_bls_g2_decode", O.g2_compress(pt), 1)[0] == 2
        assert H.call("hh_bls_g2_decode", O.g2_compress(pt), 0)[0] == 0


def test_scalar_mul_matches_oracle():
    rng = random.Random(4)
    scalars = [0, 1, 2, 15, 16, 17, O.R - 1, O.R, O.R + 5, (1 << 256) - 1, 8 << 252] + [rng.randrange(O.R) for _ in range(6)]
    h = rng.randrange(1, O.R)
    p1 = O.g1_compress(O.g1_mul(h, O.G1_GEN))
    p2 = O.g2_compress(O.g2_mul(h, O.G2_GEN))
    for k in scalars:
        kb = k.to_bytes(32, "big")
        assert H.call("hh_bls_g1_mul", kb, p1, out_sizes=(48,)) == (0, O.g1_mul_bytes(kb, p1)), hex(k)
    for k in scalars[:8] + scalars[-2:]:
        kb = k.to_bytes(32, "big")
        assert H.call("hh_bls_g2_mul", kb, p2, out_sizes=(96,)) == (0, O.g2_mul_bytes(kb, p2)), hex(k)
    # infinity in, bad point in
    kb = (5).to_bytes(32, "big")
    assert H.call("hh_bls_g1_mul", kb, O.g1_compress(None), out_sizes=(48,)) == (0, O.g1_compress(None))
    st, out = H.call("hh_bls_g1_mul", kb, bytes(48), out_sizes=(48,))
    assert st == 1 and out == bytes(48)



def test_fp_sqr_dedicated_path():
    rng = random.Random(12)
    vals = [0, 1, O.P - 1, (1 << 380) - 1, (1 << 381) - 1 - (1 << 200)] + [rng.randrange(O.P) for _ in range(300)]
    for a in vals:
        a %= O.P
        assert H.call("hh_bls_fp_op", 5, _fp(a), _fp(0), out_sizes=(48,))[1] == _fp(a * a % O.P)


def test_hash_to_curve_vs_oracle():
    dst1 = b"BLS_SIG_BLS12381G1_XMD:SHA-256_SSWU_RO_NUL_"
    dst2 = b"BLS_SIG_BLS12381G2_XMD:SHA-256_SSWU_RO_NUL_"
    for msg in (b"", b"abc", bytes(range(32)), b"x" * 55, b"y" * 64, b"z" * 200):
        for dst in (dst1, dst2, b"QUUX-V01-CS02-with-BLS12381G1_XMD:SHA-256_SSWU_RO_"):
            exp = O.g1_compress(O.hash_to_g1(msg, dst))
            assert H.call("hh_bls_hash_g1", msg or b"\\x00", len(msg), dst, len(dst), out_sizes=(48,)) == (0, exp), (msg, dst)
    for msg in (b"", b"abc", bytes(range(32)), b"w" * 100):
        exp = O.g2_compress(O.hash_to_g2(msg, dst2))
        assert H.call("hh_bls_hash_g2", msg or b"\\x00", len(msg), dst2, len(dst2), out_sizes=(96,)) == (0, exp), msg



# ------------------------------------------------------------------ call flags (include/kyber_hip.h)
F_UNC, F_UNC_OUT = 2, 4


def F_TRUSTED(i):
    return 0x100 << i


def _cofactor_points():
    """On-curve points OUTSIDE the prime-order subgroups (found by trial x)."""
    x = 1
    while True:
        y = O.fp_sqrt((x * x * x + 4) % O.P)
        if y is not None and not O.g1_in_subgroup((x, y)):
            p1 = (x, y)
            break
        x += 1
    x = 1
    while True:
        xx = (x, 1)
        y = O.f2_sqrt(O.f2_add(O.f2_mul(O.f2_sqr(xx), xx), (4, 4)))
        if y is not None and not O.g2_in_subgroup((xx, y)):
            return p1, (xx, y)
        x += 1


def test_uncompressed_decode_rules():
    rng = random.Random(11)
    k = rng.randrange(1, O.R)
    p1, p2 = O.g1_mul(k, O.G1_GEN), O.g2_mul(k, O.G2_GEN)
    u1, u2 = O.g1_serialize_unc(p1), O.g2_serialize_unc(p2)
    assert O.g1_deserialize_unc(u1) == p1 and O.g2_deserialize_unc(u2) == p2
    for fn, buf in (("hh_bls_g1_decode_unc", u1), ("hh_bls_g2_decode_unc", u2)):
        assert H.call(fn, buf, 1)[0] == 0
        assert H.call(fn, bytes([buf[0] | 0x80]) + buf[1:], 1)[0] == 1  # compression bit on an uncompressed form
        assert H.call(fn, bytes([buf[0] | 0x20]) + buf[1:], 1)[0] == 1  # sort bit
        bad = bytearray(buf)
        bad[-1] ^= 1  # y no longer satisfies the curve equation
        assert H.call(fn, bytes(bad), 1)[0] == 1
        assert H.call(fn, bytes(bad), 0)[0] == 0  # trusted: not checked
        n = len(buf)
        inf = bytes([0x40]) + bytes(n - 1)
        assert H.call(fn, inf, 1)[0] == 0
        assert H.call(fn, bytes([0x40]) + bytes(n - 2) + b"\x01", 1)[0] == 1
        big = bytes([0x1f]) + b"\xff" * 47 + buf[48:]  # x >= p
        assert H.call(fn, big, 1)[0] == 1
    c1, c2 = _cofactor_points()
    assert H.call("hh_bls_g1_decode_unc", O.g1_serialize_unc(c1), 1)[0] == 2
    assert H.call("hh_bls_g2_decode_unc", O.g2_serialize_unc(c2), 1)[0] == 2
    assert H.call("hh_bls_g1_decode_unc", O.g1_serialize_unc(c1), 0)[0] == 0


def test_mul_flags_uncompressed_in_out_and_trusted():
    rng = random.Random(12)
    for _ in range(2):
        k, h = rng.randrange(O.R), rng.randrange(1, O.R)
        kb = k.to_bytes(32, "big")
        p1, p2 = O.g1_mul(h, O.G1_GEN), O.g2_mul(h, O.G2_GEN)
        r1, r2 = O.g1_mul(k, p1), O.g2_mul(k, p2)
        for flags in (0, F_UNC, F_UNC_OUT, F_UNC | F_UNC_OUT, F_UNC | F_UNC_OUT | F_TRUSTED(0), F_TRUSTED(0)):
            i1 = O.g1_serialize_unc(p1) if flags & F_UNC else O.g1_compress(p1)
            i2 = O.g2_serialize_unc(p2) if flags & F_UNC else O.g2_compress(p2)
            e1 = O.g1_serialize_unc(r1) if flags & F_UNC_OUT else O.g1_compress(r1)
            e2 = O.g2_serialize_unc(r2) if flags & F_UNC_OUT else O.g2_compress(r2)
            assert H.call("hh_bls_g1_mul_f", kb, i1, flags, out_sizes=(len(e1),)) == (0, e1), flags
            assert H.call("hh_bls_g2_mul_f", kb, i2, flags, out_sizes=(len(e2),)) == (0, e2), flags
    # infinity round-trips in the uncompressed form
    z = (0).to_bytes(32, "big")
    assert H.call("hh_bls_g1_mul_f", z, O.g1_serialize_unc(p1), F_UNC | F_UNC_OUT, out_sizes=(96,)) == (0, O.g1_serialize_unc(None))
    # a point outside the subgroup: rejected unless the caller vouches for it
    c1, _ = _cofactor_points()
    kb = (5).to_bytes(32, "big")
    st, out = H.call("hh_bls_g1_mul_f", kb, O.g1_compress(c1), 0, out_sizes=(48,))
    assert st == 2 and out == bytes(48)
    # vouched-for input: no check, and (the GLV split assumes phi(P) = [-z^2] P) no defined result either
    st, out = H.call("hh_bls_g1_mul_f", kb, O.g1_compress(c1), F_TRUSTED(0), out_sizes=(48,))
    assert st == 0



def test_glv_gls_mul_edge_scalars():
    """The endomorphism split (k = k1 z^2 + k0 on G1, base-|z| quarters on G2) at the scalars where a quotient or a
    digit carry is extreme: multiples of z^2 and |z|^i, values >= r, all-ones."""
    z = 0xD201000000010000
    z2 = z * z
    edge = [0, 1, 2, 15, 16, z - 1, z, z + 1, z2 - 1, z2, z2 + 1, z ** 3 - 1, z ** 3, z ** 3 + 1, O.R - 1, O.R, O.R + 5,
            (1 << 256) - 1, 8 << 252, (1 << 255) + z2 * 3 + 7, z2 * ((1 << 128) - 1), (z ** 4 - 1) % (1 << 256),
            0x8888888888888888888888888888888888888888888888888888888888888888]
    h = 0x1234567
    p1, p2 = O.g1_mul(h, O.G1_GEN), O.g2_mul(h, O.G2_GEN)
    b1, b2 = O.g1_compress(p1), O.g2_compress(p2)
    for k in edge:
        kb = k.to_bytes(32, "big")
        assert H.call("hh_bls_g1_mul", kb, b1, out_sizes=(48,)) == (0, O.g1_compress(O.g1_mul(k % O.R, p1))), hex(k)
    for k in edge[::2] + [edge[-1]]:
        kb = k.to_bytes(32, "big")
        assert H.call("hh_bls_g2_mul", kb, b2, out_sizes=(96,)) == (0, O.g2_compress(O.g2_mul(k % O.R, p2))), hex(k)
    inf1, inf2 = O.g1_compress(None), O.g2_compress(None)
    kb = (12345).to_bytes(32, "big")
    assert H.call("hh_bls_g1_mul", kb, inf1, out_sizes=(48,)) == (0, inf1)
    assert H.call("hh_bls_g2_mul", kb, inf2, out_sizes=(96,)) == (0, inf2)


def test_unmarshal_wire_on_the_zcash_fixtures_and_flags(golden_dir):
    """g*_unmarshal_wire = UnmarshalBinary + MarshalBinary (kilic/g1.go:119-131): every fixture of the right length,
    both output encodings, uncompressed input, and what F_TRUSTED lets through."""
    d = json.load(open(os.path.join(golden_dir, "bls12381_zcash.json")))
    U, UO = 2, 4
    for grp, fn, size, dec, unc in (("G1", "hh_bls_g1_unmarshal", 48, O.g1_decompress, O.g1_serialize_unc),
                                    ("G2", "hh_bls_g2_unmarshal", 96, O.g2_decompress, O.g2_serialize_unc)):
        for e in d[grp]:
            buf = bytes.fromhex(e["hex"])
            if len(buf) != size:
                continue
            st, out = H.call(fn, buf, 0, out_sizes=(size,))
            assert (st == 0) == e["valid"], (grp, e["name"])
            assert out == (buf if e["valid"] else bytes(size))
            st2, out2 = H.call(fn, buf, UO, out_sizes=(2 * size,))
            assert st2 == st
            if e["valid"]:
                assert out2 == unc(dec(buf))
                assert H.call(fn, out2, U, out_sizes=(size,)) == (0, buf)  # and back
            else:
                assert out2 == bytes(2 * size)
    c1, c2 = _cofactor_points()
    assert H.call("hh_bls_g1_unmarshal", O.g1_compress(c1), 0, out_sizes=(48,)) == (2, bytes(48))
    assert H.call("hh_bls_g2_unmarshal", O.g2_compress(c2), 0, out_sizes=(96,)) == (2, bytes(96))
    assert H.call("hh_bls_g1_unmarshal", O.g1_compress(c1), F_TRUSTED(0), out_sizes=(48,)) == (0, O.g1_compress(c1))
    assert H.call("hh_bls_g1_unmarshal", O.g1_compress(None), UO, out_sizes=(96,)) == (0, O.g1_serialize_unc(None))


def _xyzz_cases(rng, gen, mul, neg):
    """Runs of +-points for the MSM's XYZZ piece accumulator, the rare branches included."""
    pts = [mul(rng.randrange(1, O.R), gen) for _ in range(6)]
    P, Q = pts[0], pts[1]
    yield [(p, rng.random() < 0.5) for p in pts]                      # plain run
    yield [(P, False)]                                                # one point
    yield []                                                          # empty run -> infinity
    yield [(P, False), (P, False), (Q, True)]                         # second addition is a doubling
    yield [(P, False), (P, True)]                                     # cancels to infinity
    yield [(P, False), (P, True), (Q, False), (Q, False), (Q, False)]  # infinity, restart, doubling, 2Q + Q
    yield [(None, False), (P, True), (None, True), (neg(P), True), (Q, False)]  # inputs at infinity are skipped


def test_xyzz_piece_accumulator_vs_oracle():
    rng = random.Random(41)
    for grp in (1, 2):
        gen, mul, add, neg, comp, size, fn = (
            (O.G1_GEN, O.g1_mul, O.g1_add, O.g1_neg, O.g1_compress, 48, "hh_bls_g1_xyzz_sum") if grp == 1 else
            (O.G2_GEN, O.g2_mul, O.g2_add, O.g2_neg, O.g2_compress, 96, "hh_bls_g2_xyzz_sum"))
        for run in _xyzz_cases(rng, gen, mul, neg):
            exp = None
            for p, s in run:
                exp = add(exp, neg(p) if s else p)
            wire = b"".join(comp(p) for p, _ in run) or b"\x00"
            signs = bytes(int(s) for _, s in run) or b"\x00"
            assert H.call(fn, len(run), wire, signs, out_sizes=(size,)) == (0, comp(exp)), (grp, run)


def test_endomorphism_split_division_by_reciprocal():
    """divmod_z (bls12381.cuh): floor(k / d), k mod d for d = z^2 and |z| through the reciprocal 2^256 + MR -- the
    constants are recomputed here, and the quotient estimate's one-off cases (k just below a multiple of d) are hit."""
    import re
    z = 0xD201000000010000
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "kyber_amd", "csrc", "bls12381.cuh")).read()
    for dw, d in ((4, z * z), (2, z)):
        blk = src[src.index("struct ZDiv<%d>" % dw):]
        words = lambda name: [int(x, 16) for x in re.findall(r"0x[0-9a-f]+", blk[blk.index(name):blk.index(";", blk.index(name))])]
        assert sum(w << (32 * i) for i, w in enumerate(words("D["))) == d
        assert (1 << 256) + sum(w << (32 * i) for i, w in enumerate(words("MR["))) == (1 << (256 + 32 * dw)) // d
        rng = random.Random(100 + dw)
        ks = [0, 1, d - 1, d, d + 1, 2 * d - 1, 2 * d, (1 << 256) - 1, (1 << 255), ((1 << 256) // d) * d - 1, ((1 << 256) // d) * d]
        ks += [rng.randrange(1 << 256) for _ in range(300)]
        ks += [min((1 << 256) - 1, rng.randrange(1, (1 << 256) // d) * d + e) for _ in range(150) for e in (-1, 0)]
        ks += [rng.randrange(1 << b) for b in (1, 31, 32, 33, 64, 127, 128, 129, 192) for _ in range(6)]
        for k in ks:
            _, q, rem = H.call("hh_bls_divmod_z", dw, k.to_bytes(32, "little"), out_sizes=(32, 16))
            assert int.from_bytes(q, "little") == k // d and int.from_bytes(rem, "little") == k % d, (dw, hex(k))


def _fb_scalars(rng, order):
    ks = [0, 1, 2, 127, 128, 129, 255, 256, 257, 511, 512, 513, 1023, 1024, 1025, (1 << 250) + 512, (512 << 10) + 513, order - 1, order, order + 1, (1 << 256) - 1, 1 << 255,
          int.from_bytes(b"\x80" * 32, "big"), int.from_bytes(b"\x81" * 32, "big"), int.from_bytes(b"\x7f" * 32, "big")]
    return ks + [rng.randrange(1 << 256) for _ in range(12)] + [rng.randrange(order) for _ in range(8)]


def test_fixed_base_table_multiplication_vs_oracle():
    """fixed_base.cuh: k P from the table of P's multiples under the group's policy -- G1: k = q z^2 + rem over the images
    (x, y), (beta x, -y), 2 x 13 signed radix-1024 digits; G2: four base-|z| digits over the psi images, 4 x 7 -- XYZZ
    additions; digit boundaries (511 / 512 / 513: the recoding's carries), scalars at and above the order, multiples of
    z^2 and |z| and their neighbours (a sub-scalar of zero), the infinity base."""
    rng = random.Random(77)
    z = O.X_ABS
    ks = _fb_scalars(rng, O.R)
    ks += [z, z - 1, z + 1, z * z, z * z - 1, z * z + 1, z**3, z**3 - 1, z**4 - 1, (1 << 256) - (1 << 256) % (z * z), ((1 << 256) // z**3) * z**3,
           512 * z * z + 512, (255 << 120) * z * z + (1 << 127), 3 * z**3 + 2 * z**2 + z]
    kb = b"".join(k.to_bytes(32, "big") for k in ks)
    for grp in (1, 2):
        gen, mul, comp, size, fn = ((O.G1_GEN, O.g1_mul, O.g1_compress, 48, "hh_bls_g1_fb_mul") if grp == 1 else
                                    (O.G2_GEN, O.g2_mul, O.g2_compress, 96, "hh_bls_g2_fb_mul"))
        P = mul(rng.randrange(1, O.R), gen)
        st, out = H.call(fn, comp(P), len(ks), kb, out_sizes=(size * len(ks),))
        assert st == 0
        for i, k in enumerate(ks):
            assert out[size * i:size * i + size] == comp(mul(k % O.R, P)), (grp, hex(k))
        st, out = H.call(fn, comp(None), 3, kb[:96], out_sizes=(size * 3,))
        assert st == 0 and out == comp(None) * 3


def test_fixed_base_table_answers_subgroup_membership():
    """fixed_base.cuh round 4: the base's r-torsion rule is read off the finished table (Scott's criteria over the plain
    image: z^2 P = -phi(P), |z| Q = -psi(Q)) instead of being laddered in the decoding lane -- members, random cofactor
    points, a member plus a small-order point, the point at infinity, compressed and uncompressed forms."""
    import ctypes

    rng = random.Random(78)

    def verdict(grp, wire, flags=0):
        lib = H.lib()
        m = ctypes.c_int(-1)
        st = lib.hh_bls_fb_member(grp, ctypes.c_char_p(wire), flags, ctypes.byref(m))
        return st, m.value

    c1, c2 = _cofactor_points()
    for grp, gen, mul, add, comp, unc, off, cof in ((1, O.G1_GEN, O.g1_mul, O.g1_add, O.g1_compress, O.g1_serialize_unc, c1, O.H1),
                                                    (2, O.G2_GEN, O.g2_mul, O.g2_add, O.g2_compress, O.g2_serialize_unc, c2, O.H_EFF_G2)):
        P = mul(rng.randrange(1, O.R), gen)
        assert verdict(grp, comp(P)) == (0, 1)
        assert verdict(grp, unc(P), 2) == (0, 1)
        assert verdict(grp, comp(None)) == (0, 1)                       # infinity: nothing to build, a member
        assert verdict(grp, comp(off)) == (0, 0)                        # on the curve, outside the subgroup
        assert verdict(grp, unc(off), 2) == (0, 0)
        small = mul(O.R, off)                                           # the cofactor component alone
        assert small is not None and verdict(grp, comp(small)) == (0, 0)
        assert verdict(grp, comp(add(P, small))) == (0, 0)              # a member plus a cofactor component
        bad = bytearray(comp(P)); bad[0] &= 0x7F                        # compression bit clear: the other rules still apply
        assert verdict(grp, bytes(bad))[0] == 1


def test_cooperative_slot_arithmetic_with_threads_as_lanes():
    """coop_slots.cuh -- the MSM's reduce / fold additions on four lanes per point through LDS slots -- run on the CPU
    with four threads as the lanes and a pthread barrier as the workgroup barrier: the slot schedule of the addition
    (5 product levels) and the doubling (3) against the oracle, for general operands, P = Q (the one-lane fallback),
    P = -Q, either operand at infinity, uncommitted steps, and chains of steps, on G1 and on G2 (Fp2)."""
    rng = random.Random(91)
    for grp in (1, 2):
        gen, mul, add, neg, comp, size, fn = (
            (O.G1_GEN, O.g1_mul, O.g1_add, O.g1_neg, O.g1_compress, 48, "hh_bls_g1_coop") if grp == 1 else
            (O.G2_GEN, O.g2_mul, O.g2_add, O.g2_neg, O.g2_compress, 96, "hh_bls_g2_coop"))
        A, B = mul(rng.randrange(1, O.R), gen), mul(rng.randrange(1, O.R), gen)
        dbl = lambda p: add(p, p)
        cases = [
            ("a", A, B, add(A, B), B), ("n", A, B, A, B), ("d", A, B, dbl(A), B), ("x", A, B, A, B),
            ("a", A, A, dbl(A), A), ("a", A, neg(A), None, neg(A)), ("a", None, B, B, B), ("a", A, None, A, None),
            ("a", None, None, None, None), ("d", None, B, None, B),
            ("aa", A, B, add(add(A, B), B), B), ("ada", A, B, add(dbl(add(A, B)), B), B),
            ("as", A, B, add(A, B), add(B, add(A, B))), ("ddda", A, B, add(mul(8, A), B), B),
            ("adnxa", A, B, add(dbl(add(A, B)), B), B),
            ("d" * 10, A, B, mul(1024, A), B),  # one window of the fixed-base table's doubling chain (fb::chain_kernel)
        ]
        for ops, p, q, ep, eq in cases:
            st, out = H.call(fn, ops.encode() + b"\0", comp(p), comp(q), out_sizes=(2 * size,))
            assert st == 0
            assert out[:size] == comp(ep) and out[size:] == comp(eq), (grp, ops)


def test_window_table_normalisation_with_entries_at_infinity():
    """jac_table8_to_affine: the ladders' table brought to affine form by ONE shared inversion; entries at infinity (a base
    of small order, or the base itself at infinity) must not poison the shared product and must come back as infinity."""
    rng = random.Random(95)
    P = O.g1_mul(rng.randrange(1, O.R), O.G1_GEN)
    for mask in (0, 0b00000100, 0b10000001, 0b01111110, 0xFF):
        st, out = H.call("hh_bls_g1_table8", O.g1_compress(P), mask, out_sizes=(48 * 8,))
        assert st == 0
        for j in range(8):
            exp = None if (mask >> j) & 1 else O.g1_mul(j + 1, P)
            assert out[48 * j:48 * j + 48] == O.g1_compress(exp), (mask, j)


def test_key_line_table_matches_the_generator_side_walk():
    """bls12381_keylines.cuh (the per-key table of the same-key verification program, computed on the device by one lane)
    against gen_tower_vm.py bls_fixed_line_table (which makes the generator's table at build time): every one of the
    68 x 4 entries, as the integer c 2^392 mod p, for the generator and for random keys."""
    import os
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "kyber_amd", "csrc"))
    import gen_tower_vm as G

    rng = random.Random(31)
    for k in range(3):
        Q = O.G2_GEN if k == 0 else O.g2_mul(rng.randrange(1, O.R), O.G2_GEN)
        st, out = H.call("hh_bls_g2_key_lines", O.g2_compress(Q), out_sizes=(68 * 4 * 48,))
        assert st == 0
        want = [c for line in G.bls_fixed_line_table(O.P, Q) for z in line for c in z]
        got = [int.from_bytes(out[48 * i:48 * i + 48], "little") for i in range(68 * 4)]
        assert got == [c * (1 << 392) % O.P for c in want], k


def test_g1_mul_on_four_cooperating_lanes_vs_oracle():
    """bls12381_g1coop.cuh -- the small-batch G1Elt.Mul: table of (j + 1) P and the GLV ladder on four lanes through LDS
    slots -- run on the CPU with four threads as the lanes: edge scalars (0, 1, the order and its neighbours, 2^256 - 1,
    multiples of z^2 so that a half is zero, digits of -8), scalars that make the accumulator meet a table entry
    (the addition's P = Q case), random ones; compressed and uncompressed forms."""
    rng = random.Random(41)
    z2 = O.X_ABS**2
    P = O.g1_mul(rng.randrange(1, O.R), O.G1_GEN)
    ks = [0, 1, 2, 7, 8, 9, 15, 16, 17, O.R - 1, O.R, O.R + 1, (1 << 256) - 1, z2, z2 - 1, z2 + 1, 5 * z2, 8 * z2 + 8, 0x88888888 * z2 + 0x8888,
          (1 << 255), int.from_bytes(b"\x88" * 32, "big"), int.from_bytes(b"\x77" * 32, "big")]
    ks += [rng.randrange(1 << 256) for _ in range(10)] + [rng.randrange(O.R) for _ in range(4)]
    for k in ks:
        st, out = H.call("hh_bls_g1_mul_coop", k.to_bytes(32, "big"), O.g1_compress(P), 0, out_sizes=(48,))
        assert st == 0 and out == O.g1_compress(O.g1_mul(k % O.R, P)), hex(k)
    k = ks[-1]
    st, out = H.call("hh_bls_g1_mul_coop", k.to_bytes(32, "big"), O.g1_serialize_unc(P), 2 | 4 | 0x100, out_sizes=(96,))
    assert st == 0 and out == O.g1_serialize_unc(O.g1_mul(k % O.R, P))
    st, out = H.call("hh_bls_g1_mul_coop", (5).to_bytes(32, "big"), O.g1_compress(None), 0, out_sizes=(48,))
    assert st == 0 and out == O.g1_compress(None)
    assert H.call("hh_bls_g1_mul_coop", (5).to_bytes(32, "big"), bytes(48), 0, out_sizes=(48,))[0] == 1


def test_g1_subgroup_rule_on_four_cooperating_lanes():
    """g1coop::member (the small-batch kernel's r-torsion test: two cooperative multiplications by |z|, Scott's criterion)
    with threads as lanes: members, cofactor points found by trial, a member plus a cofactor component."""
    import ctypes

    lib = H.lib()

    def verdict(pt):
        v = ctypes.c_int(-1)
        st = lib.hh_bls_g1_member_coop(ctypes.c_char_p(O.g1_compress(pt)), ctypes.byref(v))
        return st, v.value

    rng = random.Random(43)
    c1, _ = _cofactor_points()
    P = O.g1_mul(rng.randrange(1, O.R), O.G1_GEN)
    assert verdict(P) == (0, 1) and verdict(O.G1_GEN) == (0, 1)
    assert verdict(c1) == (0, 0)
    small = O.g1_mul(O.R, c1)
    assert verdict(small) == (0, 0) and verdict(O.g1_add(P, small)) == (0, 0)
    assert verdict(O.g1_mul(O.H1, c1)) == (0, 1)  # clearing the cofactor lands in G1


def test_g2_mul_and_subgroup_rule_on_four_cooperating_lanes():
    """g2coop (the small-batch G2Elt.Mul): the GLS ladder with psi^j applied while a table entry is staged, and the
    subgroup rule |z| Q = -psi(Q), four threads as the lanes, against the oracle: edge scalars, multiples of |z|^j (zero
    sub-scalars), digits of -8, random scalars; members and cofactor points."""
    import ctypes

    lib = H.lib()
    rng = random.Random(47)
    z = O.X_ABS
    Qp = O.g2_mul(rng.randrange(1, O.R), O.G2_GEN)
    ks = [0, 1, 8, 9, 16, O.R - 1, O.R, O.R + 1, (1 << 256) - 1, z, z - 1, z + 1, z * z, z**3, z**3 - 1, 8 * z**3 + 8 * z * z + 8 * z + 8,
          int.from_bytes(b"\x88" * 32, "big")] + [rng.randrange(1 << 256) for _ in range(6)]
    for k in ks:
        out = ctypes.create_string_buffer(96)
        st = lib.hh_bls_g2_mul_coop(ctypes.c_char_p(k.to_bytes(32, "big")), ctypes.c_char_p(O.g2_compress(Qp)), 0, out, None)
        assert st == 0 and out.raw == O.g2_compress(O.g2_mul(k % O.R, Qp)), hex(k)
    out = ctypes.create_string_buffer(96)
    assert lib.hh_bls_g2_mul_coop(ctypes.c_char_p((7).to_bytes(32, "big")), ctypes.c_char_p(O.g2_compress(None)), 0, out, None) == 0
    assert out.raw == O.g2_compress(None)

    def verdict(pt):
        v = ctypes.c_int(-1)
        st = lib.hh_bls_g2_mul_coop(ctypes.c_char_p(bytes(32)), ctypes.c_char_p(O.g2_compress(pt)), 0, None, ctypes.byref(v))
        return st, v.value

    _, c2 = _cofactor_points()
    assert verdict(Qp) == (0, 1) and verdict(O.G2_GEN) == (0, 1)
    assert verdict(c2) == (0, 0)
    small = O.g2_mul(O.R, c2)
    assert verdict(small) == (0, 0) and verdict(O.g2_add(Qp, small)) == (0, 0)


# ---- the limb-form ("lazy") base-field layer of round 5: fp_limbs.cuh, curve.cuh XyzzL ---------------------------------
_NL, _WL = 13, 30


def _limbs(v):
    assert 0 <= v < 1 << (_NL * _WL)
    return (C.c_uint32 * _NL)(*[(v >> (_WL * j)) & ((1 << _WL) - 1) if j + 1 < _NL else v >> (_WL * j) for j in range(_NL)])


def _fpl(op, a, b=0, c=0, d=0, sign=0):
    lib = H.lib()
    out = (C.c_uint32 * _NL)()
    cc = _limbs(c) if op != 1 else (C.c_uint32 * _NL)(sign)
    rv = lib.hh_bls_fpl_op(op, _limbs(a), _limbs(b), cc, _limbs(d), out)
    return rv, sum(int(x) << (_WL * j) for j, x in enumerate(out)), list(out)


def test_limb_form_primitives_vs_integers():
    """fpl_sub / fpl_sub_signed / fpl_sub_b_2c are exact integer identities with normalised limbs out; fpl_mul2sum is
    (a b + c d) R^-1 mod p below 2p; fpl_is_zero_mod_p is exact on multiples of p and their neighbours."""
    rng = random.Random(5)
    p, R = O.P, 1 << (_NL * _WL)
    norm = lambda limbs: all(x < 1 << _WL for x in limbs[:-1])
    for _ in range(300):
        a, b = rng.randrange(2 * p), rng.randrange(8 * p)
        _, v, l = _fpl(0, a, b)
        assert v == a - b + 8 * p and norm(l)
        a, b, s = rng.randrange(2 * p), rng.randrange(2 * p), rng.randrange(2)
        _, v, l = _fpl(1, a, b, sign=s)
        assert v == (-a if s else a) - b + 4 * p and norm(l)
        a, b, c = rng.randrange(2 * p), rng.randrange(2 * p), rng.randrange(2 * p)
        _, v, l = _fpl(2, a, b, c)
        assert v == a - b - 2 * c + 6 * p and norm(l)
        a, b, c, d = rng.randrange(6 * p), rng.randrange(10 * p), rng.randrange(2 * p), rng.randrange(2 * p)
        _, v, l = _fpl(3, a, b, c, d)
        assert v < 2 * p and norm(l) and (v * R - a * b - c * d) % p == 0
    for a, b, c, d in ((6 * p - 1, 10 * p - 1, 2 * p - 1, 2 * p), (0, 0, 0, 0), (p, p, p, p), (10 * p - 1, 10 * p - 1, 0, 0)):
        _, v, l = _fpl(3, a, b, c, d)
        assert v < 2 * p and norm(l) and (v * R - a * b - c * d) % p == 0
    for k in range(10):
        assert _fpl(4, k * p)[0] == 1
        for delta in (1, -1, 1 << 30, -(1 << 30), 1 << 360, p // 2):
            v = k * p + delta
            if 0 <= v < 10 * p:
                assert _fpl(4, v)[0] == 0, (k, delta)
    # same low limb as k p but a different value: the filter passes, the exact comparison must reject
    for k in (1, 5, 9):
        assert _fpl(4, k * p + (1 << 30))[0] == 0 and _fpl(4, (k * p + (3 << 90)) % (10 * p))[0] == (1 if (3 << 90) % p == 0 else 0)


def test_limb_form_piece_accumulator_vs_oracle():
    """xyzzl_madd over the exceptional runs of the packed accumulator's test and over long random runs (the lazy bounds
    are reached only after a few additions); X's top limb stays below 8p's."""
    rng = random.Random(43)
    lib = H.lib()
    runs = list(_xyzz_cases(rng, O.G1_GEN, O.g1_mul, O.g1_neg))
    base = [O.g1_mul(rng.randrange(1, O.R), O.G1_GEN) for _ in range(24)]
    for n in (3, 17, 64):
        runs.append([(rng.choice(base), rng.random() < 0.5) for _ in range(n)])
    runs.append([(base[0], False)] * 9)                                   # P, 2P (doubling), 3P ...
    runs.append([(base[1], True), (base[1], False), (base[1], False), (base[1], False)])
    for run in runs:
        exp = None
        for pt, s in run:
            exp = O.g1_add(exp, O.g1_neg(pt) if s else pt)
        wire = b"".join(O.g1_compress(pt) for pt, _ in run) or b"\x00"
        signs = bytes(int(s) for _, s in run) or b"\x00"
        out = C.create_string_buffer(48)
        top = C.c_int(0)
        bad = lib.hh_bls_g1_xyzzl_sum(len(run), wire, signs, out, C.byref(top))
        assert bad == 0 and out.raw == O.g1_compress(exp), run
        assert top.value <= (8 * O.P) >> (30 * 12)
