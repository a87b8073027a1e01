"""Host-side mirror of ``sign/eddsa`` verification (eddsa.go:143-229 VerifyWithChecks) over the engine: the
reference checks S*B == R + h*A with one fixed-base and one variable-base Point.Mul per signature; here a whole
batch is three engine launches (batch_mul_base, batch_mul, batch_add).  The byte-level canonicality / small-order
checks and SHA-512 are host plumbing exactly as in the reference (scalar.go:2308-2333, point.go:262-323)."""
from __future__ import annotations

import hashlib

import numpy as np

from ..group import edwards25519 as ed

_P = 2**255 - 19
# group/edwards25519/const.go:1453-1473 weakKeys: the y-coordinates (sign bit cleared) of the small-order points
_SMALL_ORDER_Y = (
    0, 1, _P - 1,
    0x7A03AC9277FDC74EC6CC392CFA53202A0F67100D760B3CBA4FD84D3D706A17C7,
    0x05FC536D880238B13933C6D305ACDFD5F098EFF289F4C345B027B2C28F95E826,
)


def _scalar_is_canonical(sb: bytes) -> bool:  # scalar.go:2308-2333
    return int.from_bytes(sb, "little") < ed.ORDER


def _point_is_canonical(pb: bytes) -> bool:  # point.go:296-323: y < p (sign bit ignored)
    return (int.from_bytes(pb, "little") & ((1 << 255) - 1)) < _P


def _has_small_order(enc: bytes) -> bool:  # point.go:262-294 on the canonical encoding
    return (int.from_bytes(enc, "little") & ((1 << 255) - 1)) in _SMALL_ORDER_Y


def batch_verify_with_checks(pubs, msgs, sigs) -> np.ndarray:
    """ok[i] = (VerifyWithChecks(pubs[i], msgs[i], sigs[i]) == nil)."""
    n = len(sigs)
    ok = np.zeros(n, dtype=bool)
    idx = []
    for i in range(n):
        pub, sig = bytes(pubs[i]), bytes(sigs[i])
        if len(sig) != 64 or len(pub) != 32:
            continue
        if not _scalar_is_canonical(sig[32:]) or not _point_is_canonical(sig[:32]) or not _point_is_canonical(pub):
            continue
        idx.append(i)
    if not idx:
        return ok
    R = np.frombuffer(b"".join(bytes(sigs[i])[:32] for i in idx), dtype=np.uint8).reshape(-1, 32)
    S = np.frombuffer(b"".join(bytes(sigs[i])[32:] for i in idx), dtype=np.uint8).reshape(-1, 32)
    A = np.frombuffer(b"".join(bytes(pubs[i]) for i in idx), dtype=np.uint8).reshape(-1, 32)
    h = np.frombuffer(b"".join(
        (int.from_bytes(hashlib.sha512(bytes(sigs[i])[:32] + bytes(pubs[i]) + bytes(msgs[i])).digest(), "little")
         % ed.ORDER).to_bytes(32, "little") for i in idx), dtype=np.uint8).reshape(-1, 32)
    one = np.zeros_like(S)
    one[:, 0] = 1
    Rc, st_r = ed.batch_mul(one, R)  # UnmarshalBinary(R): canonical re-encoding + decodability
    Ac, st_a = ed.batch_mul(one, A)
    SB = ed.batch_mul_base(S)
    hA, _ = ed.batch_mul(h, A)
    RhA, st_s = ed.batch_add(R, hA)
    for k, i in enumerate(idx):
        if st_r[k] or st_a[k] or st_s[k]:
            continue
        if _has_small_order(bytes(Rc[k])) or _has_small_order(bytes(Ac[k])):
            continue
        ok[i] = bytes(RhA[k]) == bytes(SB[k])
    return ok
