"""Host-side mirror of the reference's ``pairing/bn256`` suite for the hot path (pairing.Suite,
pairing/bn256/suite.go:22-107; point.go), backed by the HIP engine through the C ABI.

Wire formats (pairing/bn256/point.go): scalars 32-byte big-endian (mod.Int), G1 64 bytes x || y,
G2 128 bytes x.x || x.y || y.x || y.y, GT 384 bytes; infinity is all-zero bytes.
"""
from ._engine import F_SCALAR_BITS, F_TRUSTED, F_TRUSTED_ALL, F_UNCOMPRESSED, F_UNCOMPRESSED_OUT, Engine  # noqa: F401 (re-exported flags)

# constants.go:25
ORDER = 65000549695646603732796438742359905742570406053903786389881062969044166799969
_P = 65000549695646603732796438742359905742825358107623003571877145026864184071783
G1_LEN, G2_LEN, GT_LEN, SCALAR_LEN = 64, 128, 384, 32
G1_BASE = (1).to_bytes(32, "big") + (_P - 2).to_bytes(32, "big")  # curve.go:19-24
G2_BASE = bytes.fromhex(  # twist.go:21-33 in wire order x.x x.y y.x y.y
    "2ecca446ff6f3d4d03c76e9b5c752f28bc37b364cb05ac4a37eb32e1c3245970"
    "8f25386f72c9462b81597d65ae2092c4b97792155dcdaad32b8a6dd41792534c"
    "2db10ef5233b0fe3962b9ee6a4bbc2b5bde01a54f3513d42df972e128f31bf12"
    "274e5747e8cafacc3716cc8699db79b22f0e4ff3c23e898f694420a3be3087a5")
G1_NULL, G2_NULL = bytes(64), bytes(128)



def _neg(group: int, enc: bytes) -> bytes:
    """-P on the wire format: y -> p - y per coordinate (curvePoint.Neg / twistPoint.Neg act on y only)."""
    half = len(enc) // 2
    out = bytearray(enc[:half])
    for i in range(half, len(enc), 32):
        y = int.from_bytes(enc[i:i + 32], "big") % _P
        out += ((_P - y) % _P).to_bytes(32, "big")
    return bytes(out)


ENGINE = Engine("bn256", "bn256", ORDER, G1_LEN, G2_LEN, GT_LEN, G1_BASE, G2_BASE, G1_NULL, G2_NULL, _neg)
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


def batch_hash_g1(msgs, msg_len: int | None = None):
    """(out, status): out[i] = pointG1.Hash(msgs[i]) (pairing/bn256/point.go:261-313) for n equal-length
    messages.  `msgs` is a list of equal-length bytes objects, or a packed (n, msg_len) uint8 array /
    CUDA tensor."""
    import numpy as np

    from .._lib import check, load
    from ._engine import _is_torch, _stream

    lib = load()
    if _is_torch(msgs):
        import torch

        m = msgs.contiguous()
        n, ln = m.shape[0], (m.shape[1] if m.dim() > 1 else msg_len)
        out = torch.empty((n, 64), dtype=torch.uint8, device=m.device)
        st = torch.empty(n, dtype=torch.uint8, device=m.device)
        check(lib.kyb_bn256_hash_g1_dev(n, m.data_ptr(), ln, out.data_ptr(), st.data_ptr(), _stream()), "kyb_bn256_hash_g1_dev")
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
    check(lib.kyb_bn256_hash_g1(n, buf.ctypes.data, ln, out.ctypes.data, st.ctypes.data), "kyb_bn256_hash_g1")
    return out, st


def batch_hash_g1_svdw(msgs, dst: bytes = b""):
    """(points, status): HashG1(msg, dst) for a batch -- the package-level hash of pairing/bn256/hash.go:10-110 (HKDF-SHA-256
    to the base field, gfp.go:46-68, then the Shallue-van de Woestijne map).  `msgs` is a list of equal-length bytes
    objects, or a packed (n, msg_len) uint8 array / CUDA tensor; `dst` the domain separation tag (hash_test.go passes nil)."""
    import ctypes

    import numpy as np

    from .._lib import check, load
    from ._engine import _is_torch, _stream

    lib = load()
    dst = bytes(dst or b"")
    dbuf = ctypes.create_string_buffer(dst, len(dst)) if dst else None
    dptr = ctypes.cast(dbuf, ctypes.c_void_p) if dst else None
    if _is_torch(msgs):
        import torch

        m = msgs.contiguous()
        n, ln = m.shape[0], m.shape[1]
        out = torch.empty((n, 64), dtype=torch.uint8, device=m.device)
        st = torch.empty(n, dtype=torch.uint8, device=m.device)
        check(lib.kyb_bn256_hash_g1_svdw_dev(n, m.data_ptr(), ln, dptr, len(dst), out.data_ptr(), st.data_ptr(), _stream()),
              "kyb_bn256_hash_g1_svdw_dev")
        return out, st
    if isinstance(msgs, (list, tuple)):
        ln = len(msgs[0]) if msgs else 0
        if any(len(x) != ln for x in msgs):
            raise ValueError("batch_hash_g1_svdw: messages must have equal length")
        n = len(msgs)
        buf = np.frombuffer(b"".join(msgs), dtype=np.uint8)
    else:
        a = np.ascontiguousarray(msgs, dtype=np.uint8)
        n, ln = a.shape[0], a.shape[1]
        buf = a.reshape(-1)
    buf = np.ascontiguousarray(buf) if buf.size else np.zeros(1, dtype=np.uint8)
    out = np.empty((n, 64), dtype=np.uint8)
    st = np.empty(n, dtype=np.uint8)
    check(lib.kyb_bn256_hash_g1_svdw(n, buf.ctypes.data, ln, dptr, len(dst), out.ctypes.data, st.ctypes.data), "kyb_bn256_hash_g1_svdw")
    return out, st


def HashG1(msg: bytes, dst: bytes = b""):
    """bn256.HashG1 (hash.go:10-12): one message -> a G1 point of this suite"""
    out, st = batch_hash_g1_svdw([bytes(msg)], dst)
    return G1Elt(bytes(out[0]))
