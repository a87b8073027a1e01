"""Host-side mirror of ``sign/tbls`` (threshold BLS, tbls.go:72-151) over the engine, signatures on G1:

  Sign           tbls.go:72-86     uint16 big-endian share index || x_i * H(m)
  VerifyPartial  tbls.go:99-106    Verify(public.Eval(i).V, msg, sig)
  Recover        tbls.go:118-151   public.Eval(i) for every partial (ONE poly_eval launch), verify them (ONE
                                   launch; the one message hashed once where the suite has the entry point), then share.RecoverCommit over the first t valid ones
                                   (ONE MSM)
"""
from __future__ import annotations

import struct

from ..share import poly
from . import bls


class Scheme:
    def __init__(self, suite, bls_scheme: bls.SchemeOnG1):
        self.suite, self.bls = suite, bls_scheme
        self.key_group, self.sig_group = suite.G2(), suite.G1()

    def sign(self, private: poly.PriShare, msg: bytes) -> bytes:
        return struct.pack(">H", private.I) + self.bls.sign(private.V.MarshalBinary(), msg)

    def index_of(self, sig: bytes) -> int:
        if len(sig) != self.sig_group.PointLen() + 2:
            raise ValueError("invalid partial signature length")
        return struct.unpack(">H", sig[:2])[0]

    def verify_partial(self, public: poly.PubPoly, msg: bytes, sig: bytes) -> bool:
        i = self.index_of(sig)
        return self.bls.verify(public.Eval(i).V.MarshalBinary(), msg, sig[2:])

    def recover(self, public: poly.PubPoly, msg: bytes, sigs, t: int, n: int) -> bytes:
        cand = []
        for s in sigs:
            try:
                cand.append((self.index_of(s), s[2:]))
            except ValueError:
                continue
        keys = [s.V.MarshalBinary() for s in public.EvalMany([i for i, _ in cand])] if cand else []
        # the evaluated keys are the engine's own outputs: validated by construction
        # ONE message, a different public share per partial signature (tbls.go:118-131): schemes with a same-message
        # entry point hash it once (BLS12-381: kyb_bls12381_verify_g1_same_msg), the others get the message repeated
        if not cand:
            ok = []
        elif hasattr(self.bls, "batch_verify_same_msg"):
            ok = self.bls.batch_verify_same_msg(keys, msg, [v for _, v in cand], keys_validated=True)
        else:
            ok = self.bls.batch_verify(keys, [msg] * len(cand), [v for _, v in cand], keys_validated=True)
        shares = []
        for (i, v), good in zip(cand, ok):
            if not good:
                continue
            shares.append(poly.PubShare(i, type(self.sig_group.Point())(v)))
            if len(shares) >= t:
                break
        if len(shares) < t:
            raise ValueError("not enough valid partial signatures")
        return poly.recover_commit(self.sig_group, shares, t, n).MarshalBinary()


def NewThresholdSchemeOnG1_bn256() -> Scheme:
    from ..pairing import bn256

    return Scheme(bn256.NewSuite(), bls.NewSchemeOnG1_bn256())


def NewThresholdSchemeOnG1_bls12381(dst: bytes | None = None) -> Scheme:
    """sign/tbls over the BLS12-381 suite (signatures on G1, public sharing polynomial on G2): Recover verifies its
    partial signatures through kyb_bls12381_verify_g1_same_msg (one message, one hash)."""
    from ..pairing import bls12381

    return Scheme(bls12381.NewSuite(), bls.NewSchemeOnG1_bls12381(dst))
