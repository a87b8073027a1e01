"""Host-side mirror of ``sign/bls`` (bls.go:22-96) over the batch engine, signatures on G1 / keys on G2
(NewSchemeOnG1, bls.go:30-43): Sign = x * H(m), Verify = ValidatePairing(H(m), X, sig, G2.Base()).

``hash_to_g1`` is the suite's HashablePoint.Hash.  For bn256 it is the reference's try-and-increment
over SHA-256 (pairing/bn256/point.go:261-313), which the reference itself computes with host big
integers, not with its field code; it stays host plumbing here too.  BLS12-381 hash-to-curve (RFC 9380
SSWU) is not built (isogeny constants unavailable offline).
"""
from __future__ import annotations

import hashlib

import numpy as np


def bn256_hash_to_g1(msg: bytes) -> bytes:
    """pointG1.Hash -> hashToPoint (pairing/bn256/point.go:261-313): 64-byte G1 encoding."""
    from ..pairing.bn256 import _P as P

    x = int.from_bytes(hashlib.sha256(msg).digest(), "big") % P
    while True:
        t = (x * x * x + 3) % P
        y = pow(t, (P + 1) // 4, P)  # big.Int.ModSqrt for p = 3 mod 4
        if y * y % P == t:
            return x.to_bytes(32, "big") + y.to_bytes(32, "big")
        x = (x + 1) % P


class SchemeOnG1:
    def __init__(self, suite_module, hash_to_g1):
        self.m, self.hash = suite_module, hash_to_g1

    def sign(self, private_be: bytes, msg: bytes) -> bytes:
        """bls.go:67-80: sig = x * H(msg)."""
        out, st = self.m.g1_batch_mul(private_be, self.hash(msg))
        if st.any():
            raise ValueError("bls: hash-to-point produced an invalid point")
        return bytes(out[0])

    def batch_verify(self, publics, msgs, sigs):
        """N x Verify (bls.go:82-96) as ONE pairing-check launch: ok[i] = e(H(m_i), X_i) == e(sig_i, B2).
        Returns a bool array; an undecodable key / signature verifies false (the reference returns an error)."""
        n = len(msgs)
        H = b"".join(self.hash(m) for m in msgs)
        X = b"".join(publics)
        S = b"".join(sigs)
        ok, st = self.m.batch_validate_pairing(H, X, S, self.m.G2_BASE * n)
        return (np.asarray(ok) == 1) & (np.asarray(st) == 0)

    def verify(self, public: bytes, msg: bytes, sig: bytes) -> bool:
        return bool(self.batch_verify([public], [msg], [sig])[0])


def NewSchemeOnG1_bn256() -> SchemeOnG1:
    from ..pairing import bn256

    return SchemeOnG1(bn256, bn256_hash_to_g1)
