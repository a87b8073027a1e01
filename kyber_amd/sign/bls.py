"""Host-side mirror of ``sign/bls`` (bls.go:22-96) over the batch engine, signatures on G1 / keys on G2
(NewSchemeOnG1, bls.go:30-43): Sign = x * H(m), Verify = ValidatePairing(H(m), X, sig, G2.Base()).

``batch_hash`` is the suite's HashablePoint.Hash over a list of messages.  For bn256 it is the reference's
try-and-increment over SHA-256 (pairing/bn256/point.go:261-313), run on the device (kyb_bn256_hash_g1).
BLS12-381 uses RFC 9380 hash_to_curve on the device (kyb_bls12381_hash_g1/g2).

Operands the engine produced itself in the same call -- H(m), the group generator -- are passed with
KYB_F_TRUSTED (they are valid by construction, exactly like the kyber.Point values the reference hands to
ValidatePairing); public keys are too when the caller says they were unmarshalled (validated) before
(`keys_validated=True`, the reference's situation: Verify receives a kyber.Point).  Signatures never are.
"""
from __future__ import annotations

import numpy as np

from ..pairing._engine import pack_fixed


def bn256_batch_hash_g1(msgs) -> np.ndarray:
    """HashablePoint.Hash for a list of messages through the engine (kyb_bn256_hash_g1: SHA-256 +
    try-and-increment on the device, pairing/bn256/point.go:261-313).  Messages are grouped by length,
    one launch per distinct length."""
    from ..pairing import bn256

    out = np.empty((len(msgs), 64), dtype=np.uint8)
    by_len = {}
    for i, m in enumerate(msgs):
        by_len.setdefault(len(m), []).append(i)
    for ln, idx in by_len.items():
        h, st = bn256.batch_hash_g1([bytes(msgs[i]) for i in idx])
        if st.any():
            raise ValueError("bn256: hash-to-point did not terminate")
        out[idx] = h
    return out


class SchemeOnG1:
    def __init__(self, suite_module, batch_hash):
        self.m, self.batch_hash = suite_module, batch_hash

    def sign(self, private_be: bytes, msg: bytes) -> bytes:
        """bls.go:67-80: sig = x * H(msg)."""
        out, st = self.m.g1_batch_mul(private_be, self.batch_hash([msg]))
        if st.any():
            raise ValueError("bls: hash-to-point produced an invalid point")
        return bytes(out[0])

    def batch_verify(self, publics, msgs, sigs, keys_validated: bool = False):
        """N x Verify (bls.go:82-96) as ONE pairing-check launch: ok[i] = e(H(m_i), X_i) == e(sig_i, B2).
        Returns a bool array; an undecodable key / signature verifies false (the reference returns an error)."""
        n = len(msgs)
        if len(publics) != n or len(sigs) != n:
            raise ValueError(f"batch_verify: {n} messages, {len(publics)} public keys, {len(sigs)} signatures")
        H = self.batch_hash(msgs)
        # element by element: a wrong-length key / signature fails alone instead of shifting the rest of the batch
        X, bad_x = pack_fixed(publics, self.m.G2_LEN)
        S, bad_s = pack_fixed(sigs, self.m.G1_LEN)
        flags = self.m.F_TRUSTED(0) | self.m.F_TRUSTED(3) | (self.m.F_TRUSTED(1) if keys_validated else 0)
        ok, st = self.m.batch_validate_pairing(H, X, S, self.m.G2_BASE * n, flags)
        res = (np.asarray(ok) == 1) & (np.asarray(st) == 0)
        res[bad_x + bad_s] = False
        return res

    def verify(self, public: bytes, msg: bytes, sig: bytes, keys_validated: bool = False) -> bool:
        return bool(self.batch_verify([public], [msg], [sig], keys_validated)[0])


def _grouped(hash_fn, width):
    def batch_hash(msgs):
        out = np.empty((len(msgs), width), dtype=np.uint8)
        by_len = {}
        for i, m in enumerate(msgs):
            by_len.setdefault(len(m), []).append(i)
        for _, idx in by_len.items():
            h, st = hash_fn([bytes(msgs[i]) for i in idx])
            if np.asarray(st).any():
                raise ValueError("hash-to-curve failed")
            out[idx] = h
        return out

    return batch_hash


class _Bls12381SchemeOnG1(SchemeOnG1):
    """Same interface, but batch_verify is the fused kernel (kyb_bls12381_verify_g1): hash, unmarshal checks,
    two Miller loops and one final exponentiation per lane in one launch per distinct message length."""

    def __init__(self, suite_module, batch_hash, dst):
        super().__init__(suite_module, batch_hash)
        self.dst = dst

    def batch_verify(self, publics, msgs, sigs, keys_validated: bool = False):
        out = np.zeros(len(msgs), dtype=bool)
        by_len = {}
        for i, m in enumerate(msgs):
            by_len.setdefault(len(m), []).append(i)
        flags = self.m.F_TRUSTED(0) if keys_validated else 0
        # one signer for the whole batch (a drand chain; the loop of bls.go:82-96 with the same X): both Miller loops
        # from line tables (kyb_bls12381_verify_g1_same_key), from the batch size at which the key's table pays
        one_key = len(publics) >= self.SAME_KEY_MIN and all(bytes(p) == bytes(publics[0]) for p in publics)
        for _, idx in by_len.items():
            if one_key:
                ok, st = self.m.batch_verify_g1_same_key(bytes(publics[0]), [bytes(msgs[i]) for i in idx],
                                                         [sigs[i] for i in idx], self.dst, flags)
            else:
                ok, st = self.m.batch_verify_g1([publics[i] for i in idx], [bytes(msgs[i]) for i in idx],
                                                [sigs[i] for i in idx], self.dst, flags)
            out[idx] = (np.asarray(ok) == 1) & (np.asarray(st) == 0)
        return out

    SAME_KEY_MIN = 256

    def batch_verify_same_msg(self, publics, msg: bytes, sigs, keys_validated: bool = False):
        """Verify of many (key, signature) pairs over ONE message -- the loop of tbls.Recover (sign/tbls/tbls.go:118-131):
        H(msg) once per call (kyb_bls12381_verify_g1_same_msg)."""
        if not len(publics):
            return np.zeros(0, dtype=bool)
        flags = self.m.F_TRUSTED(0) if keys_validated else 0
        ok, st = self.m.batch_verify_g1_same_msg(list(publics), bytes(msg), list(sigs), self.dst, flags)
        return (np.asarray(ok) == 1) & (np.asarray(st) == 0)

    def batch_verify_same_key(self, public: bytes, msgs, sigs, key_validated: bool = False):
        """Verify for many messages of ONE signer (public: its key): what a drand client does with a chain of beacons."""
        return self.batch_verify([public] * len(msgs), msgs, sigs, key_validated)


def NewSchemeOnG1_bls12381(dst: bytes | None = None) -> SchemeOnG1:
    """sign/bls NewSchemeOnG1 over the BLS12-381 suite: signatures on G1 (hash_to_curve with the G1 DST of
    kilic/g1.go:17 unless `dst` is given, as NewBLS12381SuiteWithDST allows), keys on G2."""
    from ..pairing import bls12381

    d = bls12381.DOMAIN_G1 if dst is None else dst
    return _Bls12381SchemeOnG1(bls12381, _grouped(lambda m: bls12381.batch_hash_g1(m, d), 48), d)


class SchemeOnG2:
    """sign/bls NewSchemeOnG2 (bls.go:45-58): signatures on G2, keys on G1; Verify =
    ValidatePairing(G1.Base(), sig, X, H(m)) (the argument order of bls.go:51-53)."""

    def __init__(self, suite_module, batch_hash):
        self.m, self.batch_hash = suite_module, batch_hash

    def sign(self, private_be: bytes, msg: bytes) -> bytes:
        out, st = self.m.g2_batch_mul(private_be, self.batch_hash([msg]))
        if st.any():
            raise ValueError("bls: hash-to-point produced an invalid point")
        return bytes(out[0])

    def batch_verify(self, publics, msgs, sigs, keys_validated: bool = False):
        n = len(msgs)
        if len(publics) != n or len(sigs) != n:
            raise ValueError(f"batch_verify: {n} messages, {len(publics)} public keys, {len(sigs)} signatures")
        H = self.batch_hash(msgs)
        S, bad_s = pack_fixed(sigs, self.m.G2_LEN)
        X, bad_x = pack_fixed(publics, self.m.G1_LEN)
        flags = self.m.F_TRUSTED(0) | self.m.F_TRUSTED(3) | (self.m.F_TRUSTED(2) if keys_validated else 0)
        ok, st = self.m.batch_validate_pairing(self.m.G1_BASE * n, S, X, H, flags)
        res = (np.asarray(ok) == 1) & (np.asarray(st) == 0)
        res[bad_x + bad_s] = False
        return res

    def verify(self, public: bytes, msg: bytes, sig: bytes, keys_validated: bool = False) -> bool:
        return bool(self.batch_verify([public], [msg], [sig], keys_validated)[0])


class _Bls12381SchemeOnG2(SchemeOnG2):
    """Same interface, batch_verify through the fused kernel (kyb_bls12381_verify_g2)."""

    def __init__(self, suite_module, batch_hash, dst):
        super().__init__(suite_module, batch_hash)
        self.dst = dst

    def batch_verify(self, publics, msgs, sigs, keys_validated: bool = False):
        out = np.zeros(len(msgs), dtype=bool)
        by_len = {}
        for i, m in enumerate(msgs):
            by_len.setdefault(len(m), []).append(i)
        flags = self.m.F_TRUSTED(0) if keys_validated else 0
        for _, idx in by_len.items():
            ok, st = self.m.batch_verify_g2([publics[i] for i in idx], [bytes(msgs[i]) for i in idx],
                                            [sigs[i] for i in idx], self.dst, flags)
            out[idx] = (np.asarray(ok) == 1) & (np.asarray(st) == 0)
        return out


def NewSchemeOnG2_bls12381(dst: bytes | None = None) -> SchemeOnG2:
    from ..pairing import bls12381

    d = bls12381.DOMAIN_G2 if dst is None else dst
    return _Bls12381SchemeOnG2(bls12381, _grouped(lambda m: bls12381.batch_hash_g2(m, d), 96), d)


def NewSchemeOnG1_bn256() -> SchemeOnG1:
    from ..pairing import bn256

    return SchemeOnG1(bn256, bn256_batch_hash_g1)


def NewSchemeOnG1_bn254(dst: bytes | None = None) -> SchemeOnG1:
    """bls.NewSchemeOnG1(bn254.NewSuite()) (pairing/bn254/bls_test.go:12-16): signatures on G1 hashed with the suite's
    Keccak-256 / Shallue-van de Woestijne hash_to_curve, keys on G2."""
    from ..pairing import bn254

    d = bn254.DOMAIN_G1 if dst is None else dst

    def hash_one_length(msgs):
        h, st = bn254.batch_hash_g1(msgs, d)
        return h, st

    return SchemeOnG1(bn254, _grouped(hash_one_length, 64))
