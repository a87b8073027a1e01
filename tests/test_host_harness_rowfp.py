"""kyber_amd/csrc/rowfp.cuh -- field arithmetic with one limb per lane (rows of 16 lanes, four per wave) for the
latency-bound chains -- emulated lane by lane on the CPU (tests/host_harness.cpp compiles the same source with V32 = 64
lanes) and held against big integers: the Montgomery product with redundant limbs at the worst bounds the formulas reach,
the lazy additions / subtractions, and chains of doublings against the oracle's group law (pairing/bn256/curve.go:156-187's
dbl-2009-l is the formula; BLS12-381 G1 the curve), each row its own point and the four rows sharing one.  The emulation
counts every 64-bit accumulator overflow: the column bounds stated in rowfp.cuh are tested, not only argued."""
import ctypes as C
import random

import pytest

from oracle import bls12381 as O
from tests import _host_harness as H

N, W = 13, 30
R = 1 << (N * W)
RINV = pow(R, -1, O.P)
U32x64 = C.c_uint32 * 64


def _limbs(x, excess=None, rng=None):
    """13 limbs of x in one row's 16 lanes; with `excess`, a redundant representation: limbs pushed up to 2^30 + excess by
    borrowing from the limb above (same value)"""
    l = [(x >> (W * j)) & ((1 << W) - 1) for j in range(N)]
    l[N - 1] = x >> (W * (N - 1))
    if excess:
        for j in range(N - 1):
            if l[j + 1] > 0 and rng.random() < 0.7:
                l[j + 1] -= 1
                l[j] += 1 << W
                if l[j] > (1 << W) + excess:  # keep below the bound: undo
                    l[j] -= 1 << W
                    l[j + 1] += 1
    return l + [0, 0, 0]


def _val(row):
    assert row[13] == row[14] == row[15] == 0, row[13:]
    return sum(v << (W * j) for j, v in enumerate(row[:N]))


def _rows(vals4, excess=None, rng=None):
    out = []
    for v in vals4:
        out += _limbs(v, excess, rng)
    return U32x64(*out)


def _op(op, a4, b4, excess=None, rng=None):
    out = U32x64()
    lib = H.lib()
    lib.hh_row_op.restype = C.c_int
    ov = lib.hh_row_op(op, _rows(a4, excess, rng), _rows(b4, excess, rng), out)
    assert ov == 0, "a 64-bit accumulator wrapped"
    rows = [list(out[16 * r:16 * r + 16]) for r in range(4)]
    return rows


def test_montgomery_product_with_redundant_limbs_at_the_formula_bounds():
    rng = random.Random(71)
    p = O.P
    cases = []
    for _ in range(60):
        ka, kb = rng.choice([(1, 1), (2, 2), (7, 7), (14, 4), (10, 2), (6, 6), (6, 10), (4, 8), (22, 22)])
        cases.append((rng.randrange(ka * p), rng.randrange(kb * p)))
    cases += [(0, 0), (0, 5), (1, 1), (p - 1, p - 1), (7 * p - 1, 7 * p - 1), (22 * p - 1, 22 * p - 1), (R // 2 % (22 * p), 3),
              ((1 << 381) - 1, (1 << 381) - 1)]
    while len(cases) % 4:
        cases.append((1, 2))
    for excess in (None, 60):
        for i in range(0, len(cases), 4):
            a4, b4 = [c[0] for c in cases[i:i + 4]], [c[1] for c in cases[i:i + 4]]
            rows = _op(0, a4, b4, excess, rng)
            for r in range(4):
                got = _val(rows[r])
                assert got % p == a4[r] * b4[r] * RINV % p, (i, r)
                assert got < (a4[r] * b4[r] // R + p) + (p >> 18), "value bound (a b / R + p, plus q's excess over R)"
                assert max(rows[r]) < (1 << W) + 40, "limb bound of a product"


def test_lazy_additions_and_subtractions():
    rng = random.Random(72)
    p = O.P
    for _ in range(20):
        a4 = [rng.randrange(7 * p) for _ in range(4)]
        for op, kb, f in ((1, 7, lambda a, b: a + b), (2, 0, lambda a, b: 2 * a), (3, 0, lambda a, b: 3 * a),
                          (4, 3, lambda a, b: a - b + 3 * p), (5, 5, lambda a, b: a - b + 5 * p), (6, 8, lambda a, b: a - b + 8 * p)):
            b4 = [rng.randrange(kb * p) if kb else 0 for _ in range(4)]
            if kb and rng.random() < 0.3:
                b4[0] = kb * p - 1
            rows = _op(op, a4, b4, 33, rng)
            for r in range(4):
                assert _val(rows[r]) == f(a4[r], b4[r]), (op, r)   # the exact integer, not only its residue
                assert max(rows[r][:N - 1]) < (1 << W) + 8


def _mont(x):
    return x * R % O.P


def _jac_rows(pts, zs):
    X, Y, Z = [], [], []
    for (x, y), z in zip(pts, zs):
        X.append(_mont(x * z * z % O.P))
        Y.append(_mont(y * z * z * z % O.P))
        Z.append(_mont(z))
    return X, Y, Z


def _affine(Xm, Ym, Zm):
    x, y, z = (v * RINV % O.P for v in (Xm, Ym, Zm))
    if z == 0:
        return None
    zi = pow(z, -1, O.P)
    return (x * zi * zi % O.P, y * zi * zi * zi % O.P)


@pytest.mark.parametrize("wave", [0, 1])
def test_doubling_chains_against_the_group_law(wave):
    """128 doublings -- the length of the MSM's final chain at 2^20 points -- of four points (wave = 0) or of one point held
    by all four rows (wave = 1), from projective inputs with random Z; the point at infinity stays at infinity; limbs stay
    within the bound the next product tolerates."""
    rng = random.Random(73 + wave)
    lib = H.lib()
    lib.hh_row_dbl_chain.restype = C.c_int
    for n in (1, 2, 5, 128):
        pts = [O.g1_mul(rng.randrange(1, O.R), O.G1_GEN) for _ in range(4)]
        if wave:
            pts = [pts[0]] * 4
        zs = [rng.randrange(1, O.P) for _ in range(4)] if not wave else [7] * 4
        X, Y, Z = _jac_rows(pts, zs)
        if not wave:
            Z[3] = 0  # infinity in row 3
        oX, oY, oZ, mx = U32x64(), U32x64(), U32x64(), C.c_uint32()
        ov = lib.hh_row_dbl_chain(wave, n, _rows(X), _rows(Y), _rows(Z), oX, oY, oZ, C.byref(mx))
        assert ov == 0
        assert mx.value < (1 << W) + 64
        for r in range(4):
            got = _affine(*(_val(list(o[16 * r:16 * r + 16])) for o in (oX, oY, oZ)))
            want = None if (not wave and r == 3) else O.g1_mul(1 << n, pts[r])
            assert got == want, (wave, n, r)
        # the value bounds of the chain: X < 7p, Y < 5p, Z < 2p (+ the products' hair)
        for r in range(4):
            assert _val(list(oX[16 * r:16 * r + 16])) < 7 * O.P + (O.P >> 10)
            assert _val(list(oY[16 * r:16 * r + 16])) < 5 * O.P + (O.P >> 10)
            assert _val(list(oZ[16 * r:16 * r + 16])) < 2 * O.P + (O.P >> 10)


def test_leaving_the_row_form_gives_the_canonical_residue():
    """below_2p + finish_limbs (what final_rows_kernel stores) and load_packed (what it loads): any lazy value below 22p in
    a redundant representation comes out as the fully reduced packed residue, and a packed element reloads to the same"""
    rng = random.Random(75)
    lib = H.lib()
    lib.hh_row_finish.restype = C.c_int
    for v in [0, 1, O.P - 1, O.P, O.P + 1, 2 * O.P - 1, 7 * O.P - 1, 22 * O.P - 1] + [rng.randrange(22 * O.P) for _ in range(40)]:
        out = (C.c_uint32 * 12)()
        assert lib.hh_row_finish(_rows([v, 3, 5, 7], 40, rng), out) == 0
        assert sum(w << (32 * j) for j, w in enumerate(out)) == v % O.P, hex(v)


def test_key_lines_on_rows_equal_the_lone_lane_walk():
    """bls12381_keylines.cuh g2_key_lines_rows (one wave, an Fp2 product per level of its four rows) against g2_key_lines
    (one lane) on the CPU: the 68 x 4 x 12 words of the line table, equal word for word, for the generator and random keys;
    no 64-bit accumulator of the emulation wraps (the value bounds in the header hold)"""
    rng = random.Random(76)
    lib = H.lib()
    n = 68 * 4 * 48
    for x in (1, 2, O.R - 1, rng.randrange(1, O.R), rng.randrange(1, O.R)):
        key = O.g2_compress(O.g2_mul(x, O.G2_GEN))
        a, b, ov = C.create_string_buffer(n), C.create_string_buffer(n), C.c_int(-1)
        assert lib.hh_bls_g2_key_lines(key, a) == 0
        assert lib.hh_bls_g2_key_lines_rows(key, b, C.byref(ov)) == 0
        assert ov.value == 0
        assert a.raw == b.raw, x


def test_r_torsion_rule_read_off_the_walks_end():
    """g2_key_lines_rows(member_test): the walk's last point is |z| q -- held against the oracle's multiplication -- and
    g2_walk_end_is_minus_psi on it decides exactly what g2_in_subgroup decides: keys of G2 accepted; curve points off the
    subgroup (random points of E'(Fp2), a point of order 13, sums of both kinds) rejected"""
    rng = random.Random(77)
    lib = H.lib()
    lib.hh_bls_g2_key_walk_member.restype = C.c_int

    def walk(Q):
        tw, ov = C.create_string_buffer(6 * 48), C.c_int(-1)
        rc = lib.hh_bls_g2_key_walk_member(O.g2_serialize_unc(Q), tw, C.byref(ov))
        assert ov.value == 0
        v = [int.from_bytes(tw.raw[48 * j:48 * j + 48], "little") for j in range(6)]
        return rc, ((v[0], v[1]), (v[2], v[3]), (v[4], v[5]))

    def off_subgroup_point():
        while True:
            x = (rng.randrange(O.P), rng.randrange(O.P))
            y = O.f2_sqrt(O.f2_add(O.f2_mul(O.f2_sqr(x), x), (4, 4)))
            if y is not None:
                return (x, y)

    for k in (1, 2, O.R - 1, rng.randrange(1, O.R), rng.randrange(1, O.R)):
        Q = O.g2_mul(k, O.G2_GEN)
        rc, (X, Y, Z) = walk(Q)
        assert rc == 0, k
        zi = O.f2_inv(Z)
        assert (O.f2_mul(X, zi), O.f2_mul(Y, zi)) == O.g2_mul(O.X_ABS, Q), k
    W = off_subgroup_point()
    assert O.g2_on_curve(W) and not O.g2_in_subgroup(W)
    rc, (X, Y, Z) = walk(W)
    assert rc == 65
    zi = O.f2_inv(Z)
    assert (O.f2_mul(X, zi), O.f2_mul(Y, zi)) == O.g2_mul(O.X_ABS, W)   # the walk itself is exact off the subgroup too
    assert O.H2 % 169 == 0 and O.H2 % 2197 != 0   # the 13-part of E'(Fp2) is Z/13 x Z/13
    S = None
    while S is None:
        S = O.g2_mul(O.R * O.H2 // 169, off_subgroup_point())            # order 13
    assert O.g2_mul(13, S) is None
    assert walk(S)[0] == 65
    assert walk(O.g2_add(S, O.g2_mul(5, O.G2_GEN)))[0] == 65
    for _ in range(3):
        assert walk(off_subgroup_point())[0] == 65
    assert walk(O.g2_add(O.g2_mul(O.R, off_subgroup_point()), O.g2_mul(7, O.G2_GEN)))[0] == 65  # cofactor part + a key


def test_compressed_key_decode_with_its_powers_on_the_rows():
    """bls12381_keylines.cuh g2_decode_rows (lane 0 parses, the rows run the square root's two 379-bit powers through
    rowfp::pow_words) against the oracle's decompression without the subgroup rule: keys of G2, curve points off the
    subgroup, both sort flags, x with no point above it, x >= p, the flag combinations kilic rejects, infinity"""
    rng = random.Random(78)
    lib = H.lib()
    lib.hh_bls_g2_decode_rows.restype = C.c_int

    def dec(buf):
        out, inf, ov = C.create_string_buffer(4 * 48), C.c_int(-1), C.c_int(-1)
        st = lib.hh_bls_g2_decode_rows(bytes(buf), out, C.byref(inf), C.byref(ov))
        assert ov.value == 0
        v = [int.from_bytes(out.raw[48 * j:48 * j + 48], "little") for j in range(4)]
        return st, inf.value, ((v[0], v[1]), (v[2], v[3]))

    pts = [O.g2_mul(k, O.G2_GEN) for k in (1, 2, O.R - 1, rng.randrange(1, O.R))]
    tries = 0
    while len(pts) < 9:                                    # curve points off the subgroup; x real (c1 = 0) and x imaginary too
        tries += 1
        x = [(rng.randrange(O.P), rng.randrange(O.P)), (rng.randrange(O.P), 0), (0, rng.randrange(O.P))][tries % 3]
        y = O.f2_sqrt(O.f2_add(O.f2_mul(O.f2_sqr(x), x), (4, 4)))
        if y is not None:
            pts.append((x, y))
    for Q in pts:
        for R_ in (Q, O.g2_neg(Q)):
            buf = O.g2_compress(R_)
            assert O.g2_decompress(buf, subgroup_check=False) == R_
            assert dec(buf) == (0, 0, R_)
    bad = 0
    while bad < 4:                                         # no point above x: status 1
        x = (rng.randrange(O.P), rng.randrange(O.P))
        if O.f2_sqrt(O.f2_add(O.f2_mul(O.f2_sqr(x), x), (4, 4))) is None:
            buf = bytearray(x[1].to_bytes(48, "big") + x[0].to_bytes(48, "big"))
            buf[0] |= 0x80 | (0x20 if bad & 1 else 0)
            assert dec(buf)[0] == 1
            bad += 1
    good = bytearray(O.g2_compress(pts[3]))
    assert dec(b"\xc0" + bytes(95))[:2] == (0, 1)          # infinity
    assert dec(b"\xe0" + bytes(95))[0] == 1                # infinity with the sort flag
    assert dec(b"\xc0" + bytes(94) + b"\x01")[0] == 1      # infinity with a coordinate
    nocomp = bytearray(good)
    nocomp[0] &= 0x7F
    assert dec(nocomp)[0] == 1                              # compression bit clear
    big = bytearray((O.P).to_bytes(48, "big") + (5).to_bytes(48, "big"))
    big[0] |= 0x80
    assert dec(big)[0] == 1                                 # x.c1 = p
    big = bytearray((5).to_bytes(48, "big") + (O.P + 1).to_bytes(48, "big"))
    big[0] |= 0x80
    assert dec(big)[0] == 1                                 # x.c0 = p + 1
