"""The lazy bounds of the limb-form formulas (fp_limbs.cuh; curve.cuh XyzzL, jac_lazy.cuh) checked on real runs: the host
harness built with -DKYB_LZ_AUDIT makes every lazy element carry an upper bound of its value as a multiple of p; every
multiplication checks that the product of its operands' bounds (the sum of both products for a two-product
multiplication) stays below R / p, every subtraction that its subtrahend is below the multiple of p it adds, every
exact-zero test that its operand is inside the range it searches.  One failure anywhere is an error here."""
import ctypes as C
import random

from oracle import bls12381 as OB, bn254 as ON4, bn256 as ON
from tests import _host_harness as H


def _reset(lib):
    lib.hh_lz_audit_reset()


def test_bn_ladders_stay_inside_their_bounds():
    lib = H.lib_audit()
    assert lib.hh_lz_audit_failures() >= 0, "audit build expected"
    rng = random.Random(7)
    for name, O, g1fn, g2fn, enc1, enc2, head in (
            ("bn256", ON, "hh_bn_g1_mul", "hh_bn_g2_mul_f", ON.g1_marshal, ON.g2_marshal, 29186),
            ("bn254", ON4, "hh_bn4_g1_mul", "hh_bn4_g2_mul", ON4.g1_marshal, ON4.g2_marshal, 86000)):
        _reset(lib)
        ks = [0, 1, 2, O.ORDER - 1, O.ORDER, (1 << 256) - 1, 15, 16, 0x8888888888888888] + [rng.randrange(1 << 256) for _ in range(12)]
        h1, h2 = rng.randrange(1, O.ORDER), rng.randrange(1, O.ORDER)
        P, Q = O.g1_mul(h1, O.G1_GEN), O.g2_mul(h2, O.G2_GEN)
        for k in ks:
            kb = k.to_bytes(32, "big")
            out = C.create_string_buffer(64)
            assert getattr(lib, g1fn)(kb, enc1(P), out) == 0
            assert out.raw == enc1(O.g1_mul(k % O.ORDER, P)), (name, "g1", hex(k))
            out = C.create_string_buffer(128)
            assert getattr(lib, g2fn)(kb, enc2(Q), C.c_int(0x100), out) == 0  # vouched for: the GLS walk
            assert out.raw == enc2(O.g2_mul(k % O.ORDER, Q)), (name, "g2", hex(k))
        # flags = 0: the membership relation first ([u] Q on lazy limbs), then the same walk
        out = C.create_string_buffer(128)
        assert getattr(lib, g2fn)(ks[-1].to_bytes(32, "big"), enc2(Q), C.c_int(0), out) == 0
        assert out.raw == enc2(O.g2_mul(ks[-1] % O.ORDER, Q)), (name, "g2 flags = 0")
        assert lib.hh_lz_audit_failures() == 0, name
        assert 0 < lib.hh_lz_audit_max_product() < head, (name, lib.hh_lz_audit_max_product())


def test_bls12381_limb_form_piece_accumulator_stays_inside_its_bounds():
    lib = H.lib_audit()
    _reset(lib)
    rng = random.Random(9)
    base = [OB.g1_mul(rng.randrange(1, OB.R), OB.G1_GEN) for _ in range(12)]
    run = [(rng.choice(base), rng.random() < 0.5) for _ in range(80)] + [(base[0], False)] * 3 + [(base[0], True)] * 2
    exp = None
    for pt, s in run:
        exp = OB.g1_add(exp, OB.g1_neg(pt) if s else pt)
    out = C.create_string_buffer(48)
    top = C.c_int(0)
    bad = lib.hh_bls_g1_xyzzl_sum(len(run), b"".join(OB.g1_compress(pt) for pt, _ in run), bytes(int(s) for _, s in run), out, C.byref(top))
    assert bad == 0 and out.raw == OB.g1_compress(exp)
    assert lib.hh_lz_audit_failures() == 0
    assert 0 < lib.hh_lz_audit_max_product() <= 100  # P^2 with P < 10p: the largest product of the formula


def test_bls12381_g1_ladder_on_lazy_limbs_stays_inside_its_bounds():
    lib = H.lib_audit()
    _reset(lib)
    rng = random.Random(11)
    P = OB.g1_mul(rng.randrange(1, OB.R), OB.G1_GEN)
    for k in [0, 1, OB.R - 1, OB.R, (1 << 256) - 1] + [rng.randrange(1 << 256) for _ in range(8)]:
        out = C.create_string_buffer(48)
        assert lib.hh_bls_g1_mul(k.to_bytes(32, "big"), OB.g1_compress(P), out) == 0
        assert out.raw == OB.g1_compress(OB.g1_mul(k % OB.R, P)), hex(k)
    assert lib.hh_lz_audit_failures() == 0
    assert 0 < lib.hh_lz_audit_max_product() <= 400  # H^2, rr^2 with H, rr < 20p: the largest products of the native-limb formulas (R / p = 630)


def test_bls12381_g2_membership_on_lazy_limbs_stays_inside_its_bounds():
    """G2Elt.UnmarshalBinary's r-torsion test: [z]Q through jaclz_mul_u64_aff over fourteen-limb Fp2 (R' / p = 2^39)"""
    lib = H.lib_audit()
    _reset(lib)
    rng = random.Random(13)
    Q = OB.g2_mul(rng.randrange(1, OB.R), OB.G2_GEN)
    k = rng.randrange(OB.R)
    out = C.create_string_buffer(96)
    assert lib.hh_bls_g2_mul(k.to_bytes(32, "big"), OB.g2_compress(Q), out) == 0
    assert out.raw == OB.g2_compress(OB.g2_mul(k, Q))
    assert lib.hh_lz_audit_failures() == 0
    assert 0 < lib.hh_lz_audit_max_product() <= 23104  # the generic formulas' largest product sum, (2 * 76)^2


def test_lazy_jacobian_formulas_exceptional_branches_vs_oracle():
    """jaclz_madd / jaclz_dbl (and the _t pair) driven op by op: first point into an accumulator at infinity, the same
    point again (the doubling branch), the opposite point (cancellation, then a restart), doublings in between, long
    mixed runs -- every form that ships, against the oracle's plain sum, under the bound audit."""
    lib = H.lib_audit()
    rng = random.Random(21)
    forms = [(0, ON, ON.g1_mul, ON.g1_add, ON.g1_neg, ON.g1_marshal, ON.G1_GEN, ON.ORDER, 64),
             (1, ON, ON.g2_mul, ON.g2_add, ON.g2_neg, ON.g2_marshal, ON.G2_GEN, ON.ORDER, 128),
             (2, OB, OB.g1_mul, OB.g1_add, OB.g1_neg, OB.g1_compress, OB.G1_GEN, OB.R, 48),
             (3, OB, OB.g1_mul, OB.g1_add, OB.g1_neg, OB.g1_compress, OB.G1_GEN, OB.R, 48),
             (4, OB, OB.g2_mul, OB.g2_add, OB.g2_neg, OB.g2_compress, OB.G2_GEN, OB.R, 96)]
    for which, O, mul, add, neg, enc, gen, order, size in forms:
        _reset(lib)
        P, Q = mul(rng.randrange(1, order), gen), mul(rng.randrange(1, order), gen)
        P2 = add(P, P)
        runs = [
            [("a", P)],
            [("a", P), ("a", P)],                                   # equal point: the doubling branch
            [("a", P), ("s", P)],                                   # opposite point: infinity
            [("a", P), ("s", P), ("a", Q), ("d", None), ("a", Q)],  # restart after a cancellation
            [("s", P), ("d", None), ("a", P2)],                     # -2P + 2P through a doubling
            [("a", P), ("d", None), ("s", P2), ("a", Q)],           # 2P - 2P, then Q
            [("d", None), ("a", Q), ("d", None), ("d", None), ("s", P), ("a", Q), ("d", None), ("a", P)],
            [(rng.choice("as"), rng.choice([P, Q, P2])) if rng.random() < 0.6 else ("d", None) for _ in range(40)],
        ]
        for run in runs:
            exp = None
            for op, pt in run:
                if op == "d":
                    exp = add(exp, exp) if exp is not None else None
                else:
                    exp = add(exp, neg(pt) if op == "s" else pt)
            ops = "".join(op for op, _ in run).encode()
            wire = b"".join(enc(pt) if pt is not None else bytes(size) for _, pt in run)
            out = C.create_string_buffer(size)
            assert lib.hh_lz_chain(which, len(run), ops, wire, out) == 0, (which, ops)
            assert out.raw == enc(exp), (which, ops)
        assert lib.hh_lz_audit_failures() == 0, which
