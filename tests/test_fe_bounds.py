"""Structural bound audit of the GF(2^255 - 19) code (fe25519.cuh built with -DKYB_FE_AUDIT for the host): every field
element carries a magnitude bound in units of 2^25 / 2^24 per even / odd limb -- 1.01 for a carried result, 2 for a
canonical constant or a decoded value, sums add up -- and every multiplication / squaring checks its operands in.
A 64-bit column cannot overflow while mag_f * mag_g (* 2 for 2 f^2) < 2^13 / 124.5 = 65.8, and the operand whose limbs
are pre-multiplied by 19 / 38 in 32 bits (g of fe_mul, f of a squaring) needs mag < 2^31 / (19 * 2^25) = 3.36 -- the
binding one, and the reason the carries are rounded to signed limbs.  The code paths are data independent, so one walk
through each formula covers it for all inputs."""
import hashlib
import json
import os

from oracle import ed25519 as O
from tests import _host_harness as H

LIMIT = 65.8
LIMIT19 = 3.36


def _audit_max():
    return (H.call("hh_fe_audit_max_micro", audit=True)[0] / 1e6, H.call("hh_fe_audit_max19_micro", audit=True)[0] / 1e6)


def test_every_multiplication_operand_stays_far_inside_the_64_bit_columns(golden_dir):
    H.call("hh_fe_audit_reset", audit=True)
    B = O.encode(O.B)
    raw = hashlib.shake_256(b"fe-audit").digest(32 * 6)
    misc = json.load(open(os.path.join(golden_dir, "ed25519_misc.json")))
    pts = [B, (O.P + 1).to_bytes(32, "little"), bytes.fromhex(misc["small_order"][2])]
    seen = {}
    for i in range(6):
        s = raw[32 * i:32 * i + 32]
        for p in pts:
            for vt in (0, 1):  # the variable-base walk of ed25519_mul_kernel, both scalar semantics
                exp = O.mul(s, p, vartime=bool(vt))
                assert H.call("hh_ed_mul", s, p, vt, out_sizes=(32,), audit=True) == (0, exp)
    seen["window walk (decode, cached table, ge_add, ge_dbl, encode)"] = _audit_max()
    H.call("hh_fe_audit_reset", audit=True)
    for i in range(3):  # mixed additions: fixed-base table entries / MSM buckets
        s = raw[32 * i:32 * i + 32]
        exp = O.mul(s, pts[0], vartime=True)
        assert H.call("hh_ed_mul_madd", s, pts[0], out_sizes=(32,), audit=True) == (0, exp)
    seen["mixed additions (ge_madd)"] = _audit_max()
    H.call("hh_fe_audit_reset", audit=True)
    M = misc
    dst = M["rfc9380_dst"].encode()
    for msg in (b"", b"abc", b"a" * 200):
        H.call("hh_ed_hash", msg or b"\x00", len(msg), dst, len(dst), out_sizes=(32,), audit=True)
    seen["hash to curve (Elligator 2, cofactor clearing)"] = _audit_max()
    for what, (m, m19) in seen.items():
        assert 0 < m < LIMIT / 4, (what, m)  # at least two bits of headroom in the columns
        assert 0 < m19 < LIMIT19, (what, m19)
    # the reference's own analysis allows operands up to 3.3 (3.3 x 3.3 = 10.9); record what this code reaches
    print({k: (round(a, 2), round(b, 2)) for k, (a, b) in seen.items()})
