"""Ed25519 hash-to-curve device code (ed25519_h2c.cuh over fe25519.cuh / ge25519.cuh / sha512.cuh) compiled for
the host, against the reference's RFC 9380 vectors (group/edwards25519/point_test.go:369-445) and the oracle."""
import json
import os
import random

from oracle import ed25519 as O
from tests import _host_harness as H


def test_hash_to_curve_rfc9380_vectors_and_oracle(golden_dir):
    M = json.load(open(os.path.join(golden_dir, "ed25519_misc.json")))
    dst = M["rfc9380_dst"].encode()
    for v in M["rfc9380"]:
        msg = v["msg"].encode()
        out = H.call("hh_ed_hash", msg or b"\x00", len(msg), dst, len(dst), out_sizes=(32,))[1]
        assert O.decode(out) == (int(v["x"], 16), int(v["y"], 16)), v["msg"][:8]
    for ln in (0, 1, 14, 15, 16, 100, 111, 112, 113, 127, 128, 129, 300):
        msg = bytes((11 * i + ln) & 0xFF for i in range(ln))
        for d in (dst, b"kyber-test-DST"):
            out = H.call("hh_ed_hash", msg or b"\x00", ln, d, len(d), out_sizes=(32,))[1]
            assert out == O.hash_to_curve(msg, d), (ln, d)


def _limbs_value(limbs):
    """Integer value of ten radix-2^25.5 limbs (limb i has weight 2^ceil(25.5 i))."""
    return sum(int(v) << ((51 * i + 1) // 2) for i, v in enumerate(limbs))


def test_fe_mul_sq_at_the_input_bounds():
    """fe_mul / fe_sq / fe_sq2 / fe_sq_sel with limbs at the bounds fe25519.cuh states (|even| <= 1.65 * 2^26,
    |odd| <= 1.65 * 2^25): exact product mod p, and outputs inside the bounds a following add + multiply relies on.
    The 64-bit columns carry a half-limb bias into the carry chain (fe_carry_store); extremes of both signs are the
    cases that would expose an overflow or a mis-rounded carry."""
    import random

    import numpy as np

    P = (1 << 255) - 19
    be, bo = int(1.65 * (1 << 26)), int(1.65 * (1 << 25))
    rng = random.Random(99)

    def pattern(kind):
        if kind == "max":
            return [be if i % 2 == 0 else bo for i in range(10)]
        if kind == "min":
            return [-be if i % 2 == 0 else -bo for i in range(10)]
        if kind == "alt":
            return [(be if i % 2 == 0 else bo) * (1 if (i // 2) % 2 == 0 else -1) for i in range(10)]
        if kind == "alt2":
            return [(be if i % 2 == 0 else bo) * (-1 if i % 3 == 0 else 1) for i in range(10)]
        if kind == "zero":
            return [0] * 10
        if kind == "one":
            return [1] + [0] * 9
        return [rng.randint(-be, be) if i % 2 == 0 else rng.randint(-bo, bo) for i in range(10)]

    kinds = ["max", "min", "alt", "alt2", "zero", "one"] + ["rand"] * 40
    cases = [(pattern(a), pattern(b)) for a in kinds[:6] for b in kinds[:6]]
    cases += [(pattern("rand"), pattern("rand")) for _ in range(200)]
    for f, g in cases:
        fb = np.array(f, dtype=np.int32).tobytes()
        gb = np.array(g, dtype=np.int32).tobytes()
        fv, gv = _limbs_value(f), _limbs_value(g)
        for op, want in ((0, fv * gv), (1, fv * fv), (2, 2 * fv * fv), (3, fv * fv), (4, 2 * fv * fv)):
            out = np.frombuffer(H.call("hh_ed_fe_op", op, fb, gb, out_sizes=(40,))[1], dtype=np.int32)
            assert _limbs_value(out) % P == want % P, (op, f, g)
            for i, v in enumerate(out):
                assert abs(int(v)) <= (1.01 * (1 << 25) if i % 2 == 0 else 1.01 * (1 << 24)), (op, i, int(v))


def test_scalar_mul_walk_vs_oracle(golden_dir):
    """The kernels' variable-base walk (decode, signed radix-16 recoding, cached table, 4 doublings + 1 addition per
    window, encode) assembled on the host from the same ge25519.cuh / fe25519.cuh routines, against the reference's
    golden public keys (sign.input) and the Python oracle on edge scalars / points, both scalar semantics."""
    import hashlib

    import numpy as np

    kat = np.load(os.path.join(golden_dir, "ed25519_sign_input.npy"))
    B = O.encode(O.B)
    for i in range(0, 48):  # columns: a, A = a B, r, R = r B, h, S with S B = R + h A (eddsa.go:219-227)
        a, A, r, R, h, S = (bytes(kat[i, c]) for c in range(6))
        assert H.call("hh_ed_mul", a, B, 0, out_sizes=(32,)) == (0, A)
        assert H.call("hh_ed_mul", r, B, 0, out_sizes=(32,)) == (0, R)
        hA = H.call("hh_ed_mul", h, A, i & 1, out_sizes=(32,))[1]
        SB = H.call("hh_ed_mul", S, B, 0, out_sizes=(32,))[1]
        assert O.add(O.decode(R), O.decode(hA)) == O.decode(SB)
    misc = json.load(open(os.path.join(golden_dir, "ed25519_misc.json")))
    scalars = [bytes(32), (1).to_bytes(32, "little"), O.L.to_bytes(32, "little"), (O.L - 1).to_bytes(32, "little"),
               bytes([0xFF] * 32), (2**255).to_bytes(32, "little"), (2**255 - 1).to_bytes(32, "little"),
               bytes([0x88] * 32), bytes([0x08] * 32), (2**252).to_bytes(32, "little")]
    raw = hashlib.shake_256(b"hh-ed-mul").digest(32 * 24)
    scalars += [raw[32 * i:32 * i + 32] for i in range(24)]
    points = [B, b"\x01" + bytes(31), bytes(kat[3, 1]), (O.P + 1).to_bytes(32, "little"), (2).to_bytes(32, "little")]
    points += [bytes.fromhex(h) for h in misc["small_order"][:3]]
    for vt in (0, 1):
        for s in scalars:
            for p in points:
                exp = O.mul(s, p, vartime=bool(vt))
                st, out = H.call("hh_ed_mul", s, p, vt, out_sizes=(32,))
                if exp is None:
                    assert st == 1 and out == bytes(32)
                else:
                    assert (st, out) == (0, exp), (vt, s.hex(), p.hex())


def test_effective_scalar_equals_the_reference_recoding():
    """ge25519.cuh ed_effective_scalar (what the MSM's digit cutter is given) == oracle effective_scalar_consttime (the
    integer geScalarMult's radix-16 recoding multiplies by, ge.go:374-390, 419-435) on 20 000 random 256-bit scalars
    and the boundaries of the top digit"""
    rng = random.Random(77)
    C8 = int("0" + "8" * 63, 16)
    edge = [0, 1, (1 << 256) - 1, 1 << 255, (1 << 255) - 1, C8, C8 - 1, (1 << 256) - 1 - C8, 8 << 252, 9 << 252,
            (9 << 252) - 1, (8 << 252) + C8, (1 << 252) - 1, (15 << 252) + (1 << 252) - 1, 15 << 252,
            (7 << 252) + (1 << 252) - C8, (8 << 252) + (1 << 252) - C8, (8 << 252) + (1 << 252) - C8 - 1, O.L, O.L - 1]
    for i in range(20000):
        a = edge[i] if i < len(edge) else rng.getrandbits(256)
        neg, mag = H.call("hh_ed_effective_scalar", a.to_bytes(32, "little"), out_sizes=(32,))
        v = int.from_bytes(mag, "little")
        assert (-v if neg else v) == O.effective_scalar_consttime(a.to_bytes(32, "little")), hex(a)
