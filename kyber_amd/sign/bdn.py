"""Host-side mirror of ``sign/bdn`` (rogue-key-safe BLS aggregation) over the engine's MSM:

  hashPointToR            bdn.go:29-63      128-bit coefficients from BLAKE2Xs over all public keys
  NewMask publicTerms     mask.go:57-61     c_i * P_i + P_i
  AggregateSignatures     bdn.go:126-161    sum over enabled signers of (c_i + 1) * sig_i   -> ONE MSM
  AggregatePublicKeys     bdn.go:166-181    sum over enabled signers of (c_i + 1) * P_i     -> ONE MSM

Scheme on G1: signatures on G1, keys on G2 (NewSchemeOnG1, bdn.go:74-85).  Coefficients are computed over the
whole roster, aggregation only over the participants enabled in the mask.
"""
from __future__ import annotations

from ..util.blake2xs import blake2xs


def hash_point_to_r(publics) -> list[int]:
    """bdn.go:29-63: 16 output bytes per key, read as a little-endian integer (big-endian scalars reverse
    the chunk before SetBytes, which is the same value)."""
    out = blake2xs(b"".join(publics), 16 * len(publics))
    return [int.from_bytes(out[16 * i:16 * i + 16], "little") for i in range(len(publics))]


class Scheme:
    def __init__(self, suite_module):
        self.m = suite_module

    def _scalars(self, coefs, enabled):
        return b"".join(((coefs[i] + 1) % self.m.ORDER).to_bytes(32, "big") for i in enabled)

    def aggregate_public_keys(self, publics, mask_bits) -> bytes:
        coefs = hash_point_to_r(publics)
        enabled = [i for i, b in enumerate(mask_bits) if b]
        # c_i + 1 <= 2^128: the MSM runs 129-bit scalars (half the windows of a 256-bit one)
        out, st = self.m.g2_msm(self._scalars(coefs, enabled), b"".join(publics[i] for i in enabled), self.m.F_SCALAR_BITS(129))
        if st.any():
            raise ValueError("bdn: invalid public key")
        return bytes(out)

    def aggregate_signatures(self, sigs, publics, mask_bits) -> bytes:
        """`sigs` are the signatures of the enabled participants, in roster order (bdn.go:126-161)."""
        coefs = hash_point_to_r(publics)
        enabled = [i for i, b in enumerate(mask_bits) if b]
        if len(sigs) != len(enabled):
            raise ValueError("bdn: length of signatures and public keys must match")
        out, st = self.m.g1_msm(self._scalars(coefs, enabled), b"".join(sigs), self.m.F_SCALAR_BITS(129))
        if st.any():
            raise ValueError("bdn: invalid signature")
        return bytes(out)
