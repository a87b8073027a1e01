"""Host-side mirror of the reference's ``pairing/bn254`` suite (Ethereum's alt_bn128; pairing.Suite,
pairing/bn254/suite.go:35-140; point.go) for the hot path, backed by the HIP engine through the C ABI.

Wire formats (pairing/bn254/point.go): scalars 32-byte big-endian (mod.Int), G1 64 bytes x || y, G2 128 bytes
x.x || x.y || y.x || y.y, GT 384 bytes; infinity is all-zero bytes.  Unlike bn256, UnmarshalBinary rejects coordinates
>= p (gfp.go:101-118) and G2 points outside the order-n subgroup (twist.go:47-66); F_TRUSTED(i) on a G2 operand that was
unmarshalled before skips the subgroup re-check.
"""
import ctypes

import numpy as np

from ._engine import F_SCALAR_BITS, F_TRUSTED, F_TRUSTED_ALL, F_UNCOMPRESSED, F_UNCOMPRESSED_OUT, Engine  # noqa: F401 (re-exported flags)

# constants.go:23, 27
ORDER = 21888242871839275222246405745257275088548364400416034343698204186575808495617
_P = 21888242871839275222246405745257275088696311157297823662689037894645226208583
G1_LEN, G2_LEN, GT_LEN, SCALAR_LEN = 64, 128, 384, 32
G1_BASE = (1).to_bytes(32, "big") + (2).to_bytes(32, "big")  # curve.go:19-23
G2_BASE = bytes.fromhex(  # twist.go:21-33 (de-Montgomerised) in wire order x.x x.y y.x y.y
    "198e9393920d483a7260bfb731fb5d25f1aa493335a9e71297e485b7aef312c2"
    "1800deef121f1e76426a00665e5c4479674322d4f75edadd46debd5cd992f6ed"
    "090689d0585ff075ec9e99ad690c3395bc4b313370b38ef355acdadcd122975b"
    "12c85ea5db8c6deb4aab71808dcb408fe3d1e7690c43d37b4ce6cc0166fa7daa")
G1_NULL, G2_NULL = bytes(64), bytes(128)
DOMAIN_G1 = b"BN254G1_XMD:KECCAK-256_SVDW_RO_"  # suite.go:42-44


def _neg(group: int, enc: bytes) -> bytes:
    """-P on the wire format: y -> p - y per coordinate (curvePoint.Neg / twistPoint.Neg act on y only)."""
    half = len(enc) // 2
    out = bytearray(enc[:half])
    for i in range(half, len(enc), 32):
        y = int.from_bytes(enc[i:i + 32], "big")
        out += ((_P - y) % _P).to_bytes(32, "big")
    return bytes(out)


ENGINE = Engine("bn254", "bn254", ORDER, G1_LEN, G2_LEN, GT_LEN, G1_BASE, G2_BASE, G1_NULL, G2_NULL, _neg)
g1_batch_mul, g2_batch_mul = ENGINE.g1_batch_mul, ENGINE.g2_batch_mul
g1_commit, g2_commit = ENGINE.g1_commit, ENGINE.g2_commit
batch_pair, batch_validate_pairing = ENGINE.batch_pair, ENGINE.batch_validate_pairing
_mul = ENGINE.mul
g1_msm, g2_msm = ENGINE.g1_msm, ENGINE.g2_msm
gt_batch_mul = ENGINE.gt_batch_mul
g1_batch_add = lambda a, b: ENGINE.add(1, a, b)
g2_batch_add = lambda a, b: ENGINE.add(2, a, b)
g1_batch_unmarshal = lambda pts, flags=0: ENGINE.batch_unmarshal(1, pts, flags)
g2_batch_unmarshal = lambda pts, flags=0: ENGINE.batch_unmarshal(2, pts, flags)
Scalar, G1Elt, G2Elt, GTElt, Suite = ENGINE.make_types()


def NewSuite() -> Suite:
    return Suite()


def batch_hash_g1(msgs, dst: bytes = DOMAIN_G1):
    """(out, status): out[i] = pointG1.Hash(msgs[i]) (pairing/bn254/point.go:207-285: expand_message_xmd over legacy
    Keccak-256, Shallue-van de Woestijne map) for n equal-length messages under the domain separation tag `dst`
    (Suite.SetDomainG1).  `msgs` is a list of equal-length bytes objects, or a packed (n, msg_len) uint8 array /
    CUDA tensor."""
    from .._lib import check, load
    from ._engine import _is_torch, _stream

    lib = load()
    dbuf = ctypes.create_string_buffer(bytes(dst), len(dst)) if dst else None
    dptr = ctypes.cast(dbuf, ctypes.c_void_p) if dst else None
    if _is_torch(msgs):
        import torch

        m = msgs.contiguous()
        n, ln = m.shape[0], m.shape[1]
        out = torch.empty((n, 64), dtype=torch.uint8, device=m.device)
        st = torch.empty(n, dtype=torch.uint8, device=m.device)
        check(lib.kyb_bn254_hash_g1_dev(n, m.data_ptr(), ln, dptr, len(dst), out.data_ptr(), st.data_ptr(), _stream()),
              "kyb_bn254_hash_g1_dev")
        return out, st
    if isinstance(msgs, (list, tuple)):
        ln = len(msgs[0]) if msgs else 0
        if any(len(x) != ln for x in msgs):
            raise ValueError("batch_hash_g1: messages must have equal length")
        n = len(msgs)
        buf = np.frombuffer(b"".join(msgs), dtype=np.uint8)
    else:
        a = np.ascontiguousarray(msgs, dtype=np.uint8)
        n, ln = a.shape[0], a.shape[1]
        buf = a.reshape(-1)
    buf = np.ascontiguousarray(buf) if buf.size else np.zeros(1, dtype=np.uint8)
    out = np.empty((n, 64), dtype=np.uint8)
    st = np.empty(n, dtype=np.uint8)
    check(lib.kyb_bn254_hash_g1(n, buf.ctypes.data, ln, dptr, len(dst), out.ctypes.data, st.ctypes.data), "kyb_bn254_hash_g1")
    return out, st
