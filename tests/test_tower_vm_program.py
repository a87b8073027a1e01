"""The generated programs of the cooperative tower machine (kyber_amd/csrc/gen_tower_vm.py -> tower_vm.cuh) replayed on
the CPU against the oracle: the exact meaning of every instruction on residues mod p, the device's own arithmetic
(balanced 28-bit limbs, 64-bit columns, signed Montgomery reduction) with overflow assertions, and the worst-case
bounds for ANY input.  The GPU tests then only have to show that the kernel executes these instructions."""
import os
import random
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "kyber_amd", "csrc"))
import gen_tower_vm as G  # noqa: E402

from oracle import bls12381 as O  # noqa: E402


@pytest.fixture(scope="module")
def pair_prog():
    return G.build_bls12381_pair()


@pytest.fixture(scope="module")
def check_prog():
    return G.build_bls12381_check()


def _inputs(f, p, q):
    vals = [p[0], p[1], q[0][0], q[0][1], q[1][0], q[1][1]]
    return [v * f.R1 % f.p for v in vals]  # what the per-lane decode leaves: Montgomery residues, radix 2^390


def _gt_bytes(res):
    out = bytearray(576)
    for off, (v, _) in res["gt"].items():
        out[off:off + 48] = v.to_bytes(48, "big")
    return bytes(out)


def test_pair_program_equals_the_oracle_pairing(pair_prog):
    rng = random.Random(11)
    f = pair_prog.f
    for k in range(4):
        a, b = (1, 1) if k == 0 else (rng.randrange(1, O.R), rng.randrange(1, O.R))
        p, q = O.g1_mul(a, O.G1_GEN), O.g2_mul(b, O.G2_GEN)
        _, res = pair_prog.simulate(_inputs(f, p, q))
        assert _gt_bytes(res) == O.gt_to_bytes(O.pair(p, q)), k
        assert len(res["gt"]) == 12 and sum(c0 for _, c0 in res["gt"].values()) == 1


def test_pair_program_in_device_arithmetic(pair_prog):
    """limb for limb what the kernel does; asserts no column / limb / top-limb overflow on the way"""
    f = pair_prog.f
    p, q = O.g1_mul(0xC0FFEE, O.G1_GEN), O.g2_mul(0xBADC0DE, O.G2_GEN)
    _, res = pair_prog.simulate_limbs(_inputs(f, p, q))
    assert _gt_bytes(res) == O.gt_to_bytes(O.pair(p, q))
    # extreme residues (p - 1 everywhere): garbage in, but nothing may overflow
    pair_prog.simulate_limbs([f.p - 1] * 6)


def test_check_program_truth_table(check_prog):
    f = check_prog.f
    p1, q1 = O.g1_mul(5, O.G1_GEN), O.g2_mul(7, O.G2_GEN)

    def run(p2, q2, flags=0, pair_a=True):
        ins = (_inputs(f, p1, q1) if pair_a else [0] * 6) + (_inputs(f, O.g1_neg(p2), q2) if p2 else [0] * 6)
        return not check_prog.simulate(ins, flags)[1]["not_one"]

    assert run(O.g1_mul(35, O.G1_GEN), O.G2_GEN)            # e(5P, 7Q) == e(35P, Q)
    assert not run(O.g1_mul(36, O.G1_GEN), O.G2_GEN)
    assert run(O.g1_mul(7, O.G1_GEN), O.g2_mul(5, O.G2_GEN))
    assert not run(None, None, flags=2)                       # pair B at infinity: e(5P, 7Q) == 1 is false
    assert run(None, None, flags=3, pair_a=False)             # both at infinity: 1 == 1
    _, res = check_prog.simulate_limbs(_inputs(f, p1, q1) + _inputs(f, O.g1_neg(O.g1_mul(35, O.G1_GEN)), O.G2_GEN))
    assert not res["not_one"]


@pytest.fixture(scope="module")
def verify_prog():
    return G.build_bls12381_verify()


def test_fixed_point_line_table_is_the_generator_s_walk():
    """the table's walk ends where the general loop's point ends: [|x|] Q (projective), and Q is the suite's generator"""
    assert G.BLS_G2_GEN == O.G2_GEN
    lines = G.bls_fixed_line_table(O.P, G.BLS_G2_GEN)
    assert len(lines) == 63 + 5  # doublings + the ones of the parameter below its top bit


def test_verify_program_equals_the_check_program_with_the_generator(check_prog, verify_prog):
    """e(P1, Q1) e(P2, g2) == 1 decided by the fixed-point program as by the general one: bls.Verify's shape
    (P1 = H(m), Q1 = pk = [s] g2, -P2 = sig = [s] H(m)), a wrong signature, a wrong key, pair B dead, both dead; and the
    Miller values themselves agree (same lines, not merely the same verdict)"""
    f = verify_prog.f
    rng = random.Random(5)

    def run(prog, p1, q1, p2, flags=0):
        ins = (_inputs(f, p1, q1) if p1 else [0] * 6) + ([v * f.R1 % f.p for v in O.g1_neg(p2)[:2]] if p2 else [0, 0])
        if prog is check_prog:
            ins = ins + [v * f.R1 % f.p for v in (O.G2_GEN[0][0], O.G2_GEN[0][1], O.G2_GEN[1][0], O.G2_GEN[1][1])]
        S, res = prog.simulate(ins, flags)
        return not res["not_one"], S

    for k in range(3):
        s, h = rng.randrange(1, O.R), rng.randrange(1, O.R)
        hm, pk = O.g1_mul(h, O.G1_GEN), O.g2_mul(s, O.G2_GEN)
        sig = O.g1_mul(s, hm)
        ok_v, Sv = run(verify_prog, hm, pk, sig)
        ok_c, Sc = run(check_prog, hm, pk, sig)
        assert ok_v and ok_c
        bad = O.g1_mul(s + 1, hm)
        assert not run(verify_prog, hm, pk, bad)[0]
        assert not run(verify_prog, hm, O.g2_mul(s + 1, O.G2_GEN), sig)[0]
    assert not run(verify_prog, hm, pk, None, flags=2)[0]      # signature at infinity: e(H, pk) == 1 is false
    assert run(verify_prog, None, None, None, flags=3)[0]      # both pairs dead: 1 == 1
    assert not run(verify_prog, None, None, sig, flags=1)[0]   # pair A dead: e(-sig, g2) == 1 only for sig = O


def test_verify_program_in_device_arithmetic_and_bounds(verify_prog):
    f = verify_prog.f
    s, h = 0xC0FFEE, 0xBADC0DE
    hm, pk = O.g1_mul(h, O.G1_GEN), O.g2_mul(s, O.G2_GEN)
    sig = O.g1_mul(s, hm)
    ins = _inputs(f, hm, pk) + [v * f.R1 % f.p for v in O.g1_neg(sig)[:2]]
    _, res = verify_prog.simulate_limbs(ins)
    assert not res["not_one"]
    verify_prog.simulate_limbs([f.p - 1] * 8)  # garbage in: nothing may overflow
    col, val = verify_prog.check_bounds()
    assert col < 63 and val < 1024
    words, sched = verify_prog.encode()
    assert sum(ln * rep for _, ln, rep in sched) == verify_prog.stats()["executed"]
    # every table load stays inside the table for every repetition of its block, and is encoded as (base, stride)
    loads = 0
    for start, ln, rep in verify_prog.sched:
        for ins_ in verify_prog.ins[start:start + ln]:
            for r in ins_:
                if r["op"] == G.OP_GCLOAD:
                    loads += rep
                    assert r["arg"][0] + r["arg"][1] * (rep - 1) + G.GC_ENTRIES <= len(verify_prog.gconsts)
    assert loads == 68
    n = 0
    for i in range(0, len(words), G.REC_WORDS):
        if (words[i] >> 21) & 15 == G.OP_GCLOAD:
            n += 1
            assert words[i] & 63 == verify_prog.dyn_base and (words[i + 1] & 0xffff) + G.GC_ENTRIES <= len(verify_prog.gconsts)
    assert n >= 10  # one per run of doublings and per addition
    assert verify_prog.dyn_base + G.GC_ENTRIES <= 32  # tvm::MAX_CONSTS


def test_worst_case_bounds(pair_prog, check_prog):
    """for ANY residues: operand limbs < 2^31, accumulator columns < 2^63, |value| < 1024 p, canonicaliser inputs < 4p"""
    for prog in (pair_prog, check_prog):
        col, val = prog.check_bounds()
        assert col < 63 and val < 1024
        st = prog.stats()
        assert st["executed"] < 1000


def test_encoding_round_trip(pair_prog):
    """the emitted words decode back to the instruction stream (header fields, term operands, schedule lengths)"""
    words, sched = pair_prog.encode()
    assert len(words) % (G.WAVES * G.REC_WORDS) == 0
    assert sum(ln * rep for _, ln, rep in sched) == pair_prog.stats()["executed"]
    nins = len(words) // (G.WAVES * G.REC_WORDS)
    for start, ln, rep in sched:
        assert 0 <= start and start + ln <= nins and rep >= 1
    for i in range(0, len(words), G.REC_WORDS):
        hdr = words[i]
        op, nterm = (hdr >> 21) & 15, (hdr >> 6) & 63
        assert op <= G.OP_CMP_NZ2 and nterm <= 31
        if op == G.OP_DOT:
            for k in range(nterm):
                w0 = words[i + 1 + 2 * k]
                assert (w0 >> 24) & 3 in (G.K_PROD, G.K_LIN, G.K_PROD_CONST)
                assert (w0 & 63) < pair_prog.nslots


# ---- the BN programs (bn256: xi = 3 + i, 28-bit limbs; bn254: xi = 9 + i, 27-bit limbs)
@pytest.fixture(scope="module", params=["bn256", "bn254"])
def bn_case(request):
    import importlib

    build = {"bn256": (G.build_bn256_pair, G.build_bn256_check), "bn254": (G.build_bn254_pair, G.build_bn254_check)}[request.param]
    return importlib.import_module("oracle." + request.param), build[0](), build[1]()  # the programs the kernels run


def _bn_gt_bytes(res):
    out = bytearray(384)
    for off, (v, _) in res["gt"].items():
        out[off:off + 32] = v.to_bytes(32, "big")
    return bytes(out)


def test_bn_pair_programs_equal_the_oracle_pairing(bn_case):
    ON, pair, _ = bn_case
    f = pair.f
    rng = random.Random(12)
    for k in range(3):
        a, b = (1, 1) if k == 0 else (rng.randrange(1, ON.ORDER), rng.randrange(1, ON.ORDER))
        p, q = ON.g1_mul(a, ON.G1_GEN), ON.g2_mul(b, ON.G2_GEN)
        _, res = pair.simulate(_inputs(f, p, q))
        assert _bn_gt_bytes(res) == ON.gt_marshal(ON.pair(p, q)), k
    p, q = ON.g1_mul(0xC0FFEE, ON.G1_GEN), ON.g2_mul(0xBADC0DE, ON.G2_GEN)
    _, res = pair.simulate_limbs(_inputs(f, p, q))            # limb for limb what the kernel does, overflow asserts on
    assert _bn_gt_bytes(res) == ON.gt_marshal(ON.pair(p, q))
    pair.simulate_limbs([f.p - 1] * 6)


def test_bn_check_programs_and_bounds(bn_case):
    ON, pair, check = bn_case
    f = check.f
    p1, q1 = ON.g1_mul(5, ON.G1_GEN), ON.g2_mul(7, ON.G2_GEN)

    product = ON.__name__.endswith("bn254")  # bn254: e(p1, p2) e(-inv1, inv2) == 1 (its G2 operands are in the subgroup)

    def run(p2, q2, flags=0):
        if p2 and product:
            p2 = ON.g1_neg(p2)                                # the operand kernel's part
        ins = _inputs(f, p1, q1) + (_inputs(f, p2, q2) if p2 else [0] * 6)
        return not check.simulate(ins, flags)[1]["not_one"]

    assert run(ON.g1_mul(35, ON.G1_GEN), ON.G2_GEN)           # e(5P, 7Q) == e(35P, Q)
    assert run(ON.g1_mul(7, ON.G1_GEN), ON.g2_mul(5, ON.G2_GEN))
    assert not run(ON.g1_mul(36, ON.G1_GEN), ON.G2_GEN)
    assert not run(None, None, flags=2)                       # pair B at infinity pairs to one; e(5P, 7Q) != 1
    for prog in (pair, check):
        col, val = prog.check_bounds()
        assert col < 63 and val < 1024
    assert pair.f.W == (27 if ON.__name__.endswith("bn254") else 28)


def test_bn256_product_form_check_and_degenerate_g2_points():
    """bn256 keeps the reference's two pairings + Equal by default and takes the product form when the caller vouches
    for both G2 operands; same truth table.  And Pair on the G2 inputs bn256 alone accepts -- twist points of small
    order (the cofactor 2p - n has the factors 13 and 7369) and subgroup points with such a component -- equals the
    reference's formulas (oracle), zero Miller values included."""
    from oracle import bn256 as ON

    check = G.build_bn256_check_product()
    f = check.f
    p1, q1 = ON.g1_mul(5, ON.G1_GEN), ON.g2_mul(7, ON.G2_GEN)

    def run(p2, q2, flags=0):
        ins = _inputs(f, p1, q1) + (_inputs(f, ON.g1_neg(p2), q2) if p2 else [0] * 6)
        return not check.simulate(ins, flags)[1]["not_one"]

    assert run(ON.g1_mul(35, ON.G1_GEN), ON.G2_GEN) and run(ON.g1_mul(7, ON.G1_GEN), ON.g2_mul(5, ON.G2_GEN))
    assert not run(ON.g1_mul(36, ON.G1_GEN), ON.G2_GEN) and not run(None, None, flags=2)
    col, val = check.check_bounds()
    assert col < 63 and val < 1024
    pair = G.build_bn256_pair()
    h = 2 * ON.P - ON.ORDER
    assert h % 13 == 0 and h % 7369 == 0
    rng = random.Random(5)
    while True:
        x = (rng.randrange(ON.P), rng.randrange(ON.P))
        y = ON.f2_sqrt(ON.f2_add(ON.f2_mul(ON.f2_sqr(x), x), ON.TWIST_B))
        if y is not None:
            break
    p = ON.g1_mul(5, ON.G1_GEN)
    for q in (13, 7369):
        small = ON.g2_mul(ON.ORDER * h // q, (x, y))
        if small is None:
            continue
        for Q in (small, ON.g2_add(ON.g2_mul(7, ON.G2_GEN), small)):
            _, res = pair.simulate(_inputs(f, p, Q))
            assert _bn_gt_bytes(res) == ON.gt_marshal(ON.pair(p, Q)), q


def test_bn256_product_form_marks_zero_miller_values():
    """Round 4: bn256's default ValidatePairing is the product form + a fallback.  The product program leaves result-flag
    bit 8 (FLAG_MILLER_NONZERO) set unless the joint Miller value is zero -- which takes a G2 operand with a component of
    order 13 -- and exactly then the reference's "two pairings + Equal" can differ (0 == 0 is TRUE there): those lanes go
    to the two-pairing program, fed -inv1 (zero-ness is untouched by the sign).  Both simulators; the fallback program's
    verdict on the prepared inputs equals the oracle's ValidatePairing."""
    from oracle import bn256 as ON

    prod, two = G.build_bn256_check_product(), G.build_bn256_check()
    f = prod.f
    h = 2 * ON.P - ON.ORDER
    rng = random.Random(5)
    while True:
        x = (rng.randrange(ON.P), rng.randrange(ON.P))
        y = ON.f2_sqrt(ON.f2_add(ON.f2_mul(ON.f2_sqr(x), x), ON.TWIST_B))
        if y is not None:
            break
    Q13 = ON.g2_mul(ON.ORDER * h // 13, (x, y))
    assert Q13 is not None and ON.g2_mul(13, Q13) is None
    p1, q1 = ON.g1_mul(5, ON.G1_GEN), ON.g2_mul(7, ON.G2_GEN)
    cases = [
        (p1, q1, ON.g1_mul(35, ON.G1_GEN), ON.G2_GEN),              # ordinary, true
        (p1, q1, ON.g1_mul(36, ON.G1_GEN), ON.G2_GEN),              # ordinary, false
        (p1, Q13, ON.g1_mul(35, ON.G1_GEN), ON.G2_GEN),             # one Miller value zero: false either way
        (p1, Q13, ON.g1_mul(3, ON.G1_GEN), ON.g2_mul(2, Q13)),      # both zero: the reference says TRUE
        (p1, ON.g2_add(q1, Q13), ON.g1_mul(7, ON.G1_GEN), ON.g2_add(ON.g2_mul(5, ON.G2_GEN), Q13)),  # a small component, values non-zero
    ]
    for k, (pa, qa, pb, qb) in enumerate(cases):
        ins = _inputs(f, pa, qa) + _inputs(f, ON.g1_neg(pb), qb)      # what check_prep_kernel leaves for the product form
        want = ON.validate_pairing(pa, qa, pb, qb)
        _, res = prod.simulate(ins)
        marked = not (res["flags"] & G.FLAG_MILLER_NONZERO)
        assert marked == (k in (2, 3)), k
        if marked:
            assert not two.simulate(ins)[1]["not_one"] == want, k   # the fallback's verdict on the SAME inputs
        else:
            assert (not res["not_one"]) == want, k
    _, res = prod.simulate_limbs(_inputs(f, *cases[3][:2]) + _inputs(f, ON.g1_neg(cases[3][2]), cases[3][3]))
    assert not (res["flags"] & G.FLAG_MILLER_NONZERO)
    _, res = prod.simulate_limbs(_inputs(f, *cases[0][:2]) + _inputs(f, ON.g1_neg(cases[0][2]), cases[0][3]))
    assert (res["flags"] & G.FLAG_MILLER_NONZERO) and not res["not_one"]
    col, val = prod.check_bounds()
    assert col < 63 and val < 1024


# ---------------------------------------------------------------- G2 membership decided at the end of the Miller loop
def _twist_points():
    """members and non-members of G2 on the twist: a member; a random twist point (order r h' with h' huge); points of
    every small prime order dividing the cofactor (the Miller loop's point passes through +-Q / infinity on these:
    the exceptional steps must end in Z = 0); a member plus a point of small order"""
    rng = random.Random(31)
    member = O.g2_mul(rng.randrange(1, O.R), O.G2_GEN)
    while True:
        x = (rng.randrange(O.P), rng.randrange(O.P))
        y = O.f2_sqrt(O.f2_add(O.f2_mul(O.f2_sqr(x), x), (4, 4)))
        if y is not None:
            big = (x, y)
            break
    cof = O.g2_mul(O.R, big)                      # in the cofactor group: order divides H2
    small, h = [], O.H2
    for q in (13, 23, 2713, 11953, 262069):
        assert h % q == 0
        e = h
        while e % q == 0:
            e //= q
        pt = O.g2_mul(e, cof)                      # order a power of q
        while pt is not None and O.g2_mul(q, pt) is not None:
            pt = O.g2_mul(q, pt)
        if pt is not None:
            small.append((q, pt))
    assert len(small) >= 3
    return member, big, small


def _q_inputs(f, q):
    return [v * f.R1 % f.p for v in (q[0][0], q[0][1], q[1][0], q[1][1])]


def test_g2_membership_at_the_end_of_the_miller_loop(pair_prog, check_prog, verify_prog):
    f = pair_prog.f
    member, big, small = _twist_points()
    p = O.g1_mul(7, O.G1_GEN)
    pin = [v * f.R1 % f.p for v in p]
    cases = [(member, True), (big, False), (O.g2_add(member, small[0][1]), False)] + [(pt, False) for _, pt in small]
    for q, ok in cases:
        assert O.g2_in_subgroup(q) == ok
        _, res = pair_prog.simulate(pin + _q_inputs(f, q))
        assert bool(res["flags"] & G.FLAG_G2_A) == (not ok), (q, ok)
        if ok:
            assert res["flags"] == 0 and _gt_bytes(res) == O.gt_to_bytes(O.pair(p, q))
    # the membership of Q does not depend on P: a lane whose G1 operand is at infinity (pair dead, garbage P) still decides it
    _, res = pair_prog.simulate([0, 0] + _q_inputs(f, big), flags=1)
    assert res["flags"] & G.FLAG_G2_A
    _, res = pair_prog.simulate([0, 0] + _q_inputs(f, member), flags=1)
    assert not res["flags"] & G.FLAG_G2_A
    # CHECK: each pair raises its own bit; VERIFY: pair A only (pair B's G2 point is the generator's table)
    neg = [v * f.R1 % f.p for v in O.g1_neg(p)]
    for qa, qb, want in ((member, member, 0), (big, member, G.FLAG_G2_A), (member, small[0][1], G.FLAG_G2_B),
                         (small[1][1], big, G.FLAG_G2_A | G.FLAG_G2_B)):
        _, res = check_prog.simulate(pin + _q_inputs(f, qa) + neg + _q_inputs(f, qb))
        assert res["flags"] & (G.FLAG_G2_A | G.FLAG_G2_B) == want
    _, res = check_prog.simulate(pin + _q_inputs(f, member) + neg + _q_inputs(f, member))
    assert res["flags"] == 0                        # e(P, Q) e(-P, Q) == 1 and both members
    for qa, want in ((member, 0), (big, G.FLAG_G2_A), (small[2][1], G.FLAG_G2_A)):
        _, res = verify_prog.simulate(pin + _q_inputs(f, qa) + neg)
        assert res["flags"] & (G.FLAG_G2_A | G.FLAG_G2_B) == want
    # device arithmetic on a non-member and on a small-order point (zeros everywhere): nothing overflows, same verdict
    for q in (big, small[0][1]):
        _, res = pair_prog.simulate_limbs(pin + _q_inputs(f, q))
        assert res["flags"] & G.FLAG_G2_A


def test_bn254_g2_membership_at_the_end_of_the_ate_loop():
    """pairing/bn254 rejects G2 points outside the order-n subgroup (twist.go:47-66); the programs decide it on the ate
    loop's final point: T = [6u+2]Q + pi(Q) - pi^2(Q) = -pi^3(Q) exactly for members.  Members, a random twist point,
    points of each prime order of the cofactor (exceptional steps), a member plus such a point; bn256's programs carry no
    such test (its reference has none)."""
    from oracle import bn254 as O4

    pair, check = G.build_bn254_pair(), G.build_bn254_check()
    f = pair.f
    rng = random.Random(41)
    member = O4.g2_mul(rng.randrange(1, O4.ORDER), O4.G2_GEN)
    while True:
        xx = (rng.randrange(O4.P), rng.randrange(O4.P))
        yy = O4.f2_sqrt(O4.f2_add(O4.f2_mul(O4.f2_sqr(xx), xx), O4.TWIST_B))
        if yy is not None:
            break
    big = (xx, yy)
    h = 2 * O4.P - O4.ORDER
    small = [s for s in (O4.g2_mul(O4.ORDER * h // q, big) for q in O4.G2_COFACTOR_PRIMES) if s is not None]
    assert len(small) >= 3
    p = O4.g1_mul(7, O4.G1_GEN)
    cases = [(member, True), (big, False), (O4.g2_add(member, small[0]), False)] + [(s, False) for s in small]
    for q, ok in cases:
        _, res = pair.simulate(_inputs(f, p, q))
        assert bool(res["flags"] & G.FLAG_G2_A) == (not ok), ok
        if ok:
            assert res["flags"] == 0 and _bn_gt_bytes(res) == O4.gt_marshal(O4.pair(p, q))
    _, res = pair.simulate_limbs(_inputs(f, p, small[0]))
    assert res["flags"] & G.FLAG_G2_A
    neg = O4.g1_neg(p)
    for qa, qb, want in ((member, member, 0), (big, member, G.FLAG_G2_A), (member, small[1], G.FLAG_G2_B), (small[0], big, G.FLAG_G2_A | G.FLAG_G2_B)):
        _, res = check.simulate(_inputs(f, p, qa) + _inputs(f, neg, qb))
        assert res["flags"] & (G.FLAG_G2_A | G.FLAG_G2_B) == want
    _, res = check.simulate(_inputs(f, p, member) + _inputs(f, neg, member))
    assert res["flags"] == 0
    # bn256's programs carry no membership verdicts (its UnmarshalBinary has none); the product form's only other flag is
    # the non-zero mark of the joint Miller value
    for prog in (G.build_bn256_pair(), G.build_bn256_check(), G.build_bn256_check_product()):
        assert not any(r.get("flag", 1) & (G.FLAG_G2_A | G.FLAG_G2_B) for ins in prog.ins for r in ins)
        assert {r.get("flag", 1) for ins in prog.ins for r in ins} <= {1, G.FLAG_MILLER_NONZERO}


# ---- VERIFYK: both G2 operands fixed (one public key for every lane), both Miller loops from line tables
def _samekey_inputs(f, hm, sig):
    vals = (list(hm[:2]) if hm else [0, 0]) + (list(O.g1_neg(sig)[:2]) if sig else [0, 0])
    return [v * f.R1 % f.p for v in vals]


def test_same_key_verify_program_equals_the_verify_program(verify_prog):
    """e(H(m), X) e(-sig, g2) == 1 with X's lines from a table: the verdicts AND the Miller-loop value's verdict agree with
    VERIFY (whose first loop walks X) for a valid signature, a wrong signature, a signature under another key, either
    pair dead -- for several keys (a program is built per key here; on the device the table half is data)."""
    rng = random.Random(17)
    for k in range(2):
        s = rng.randrange(1, O.R)
        pk = O.g2_mul(s, O.G2_GEN)
        prog = G.build_bls12381_verify_same_key(pk)
        f = prog.f
        assert len(prog.gconsts) == 2 * 68 * 4 and prog.key_table_base == 68 * 4 and prog.ndyn == 8
        for j in range(2):
            hm = O.g1_mul(rng.randrange(1, O.R), O.G1_GEN)
            sig = O.g1_mul(s, hm)
            assert not prog.simulate(_samekey_inputs(f, hm, sig))[1]["not_one"]
            assert prog.simulate(_samekey_inputs(f, hm, O.g1_mul(s + 1, hm)))[1]["not_one"]
            assert prog.simulate(_samekey_inputs(f, O.g1_mul(3, hm), sig))[1]["not_one"]
        # the same triple through VERIFY (the key as an operand): same verdicts
        ins_v = _inputs(f, hm, pk) + [v * f.R1 % f.p for v in O.g1_neg(sig)[:2]]
        assert not verify_prog.simulate(ins_v)[1]["not_one"]
        assert prog.simulate(_samekey_inputs(f, hm, None), flags=2)[1]["not_one"]       # signature at infinity
        assert not prog.simulate(_samekey_inputs(f, None, None), flags=3)[1]["not_one"]  # both pairs dead: 1 == 1
        assert prog.simulate(_samekey_inputs(f, None, sig), flags=1)[1]["not_one"]       # pair A dead (key at infinity)


def test_same_key_verify_program_in_device_arithmetic_and_bounds():
    prog = G.build_bls12381_verify_same_key(O.g2_mul(0xC0FFEE, O.G2_GEN))
    f = prog.f
    hm = O.g1_mul(0xBADC0DE, O.G1_GEN)
    sig = O.g1_mul(0xC0FFEE, hm)
    _, res = prog.simulate_limbs(_samekey_inputs(f, hm, sig))
    assert not res["not_one"]
    _, res = prog.simulate_limbs(_samekey_inputs(f, hm, O.g1_mul(2, sig)))
    assert res["not_one"]
    prog.simulate_limbs([f.p - 1] * 4)  # garbage in: nothing may overflow
    col, val = prog.check_bounds()
    assert col < 63 and val < 1024
    words, sched = prog.encode()
    assert sum(ln * rep for _, ln, rep in sched) == prog.stats()["executed"]
    loads = {0: 0, G.GC_ENTRIES: 0}
    for start, ln, rep in prog.sched:
        for ins_ in prog.ins[start:start + ln]:
            for r in ins_:
                if r["op"] == G.OP_GCLOAD:
                    loads[r["dst"]] += rep
                    assert r["arg"][0] + r["arg"][1] * (rep - 1) + G.GC_ENTRIES <= len(prog.gconsts)
                    assert (r["arg"][0] >= prog.key_table_base) == (r["dst"] == 0)  # the key's lines feed constants 0 .. 3
    assert loads == {0: 68, G.GC_ENTRIES: 68}
    # cheaper than VERIFY, which walks the key's point: the reason the entry point exists
    assert prog.mads() < 0.9 * G.build_bls12381_verify().mads()
