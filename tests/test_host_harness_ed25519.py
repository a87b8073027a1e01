"""Ed25519 hash-to-curve device code (ed25519_h2c.cuh over fe25519.cuh / ge25519.cuh / sha512.cuh) compiled for
the host, against the reference's RFC 9380 vectors (group/edwards25519/point_test.go:369-445) and the oracle."""
import json
import os

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
