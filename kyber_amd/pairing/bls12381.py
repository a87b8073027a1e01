"""Host-side mirror of the reference's ``pairing/bls12381`` suites for the hot path
(pairing.Suite, pairing/pairing.go:8-20; kilic adapter kilic/{g1,g2,gt,suite}.go), backed by the
HIP engine through the C ABI.  No curve or field arithmetic happens in Python.

Wire formats are the adapters' MarshalBinary encodings: scalars 32-byte big-endian (mod.Int),
G1 48-byte / G2 96-byte ZCash compressed, GT 576 bytes.

Batch functions accept host data (bytes / numpy uint8) or device-resident ``torch.uint8`` CUDA
tensors; device inputs are processed on the current stream and results stay on the device.
"""
from ._engine import F_SCALAR_BITS, F_TRUSTED, F_TRUSTED_ALL, F_UNCOMPRESSED, F_UNCOMPRESSED_OUT, Engine  # noqa: F401 (re-exported flags)

ORDER = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001  # kilic/scalar.go:11-12
G1_LEN, G2_LEN, GT_LEN, SCALAR_LEN = 48, 96, 576, 32
G1_BASE = bytes.fromhex(
    "97f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb")
G2_BASE = bytes.fromhex(
    "93e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e"
    "024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8")
G1_NULL = bytes([0xC0]) + bytes(47)
G2_NULL = bytes([0xC0]) + bytes(95)



def _neg(group: int, enc: bytes) -> bytes:
    """-P on the ZCash compressed encoding: flip the y-sign flag (no point of the r-torsion has y = 0)."""
    if enc[0] & 0x40:  # infinity
        return enc
    return bytes([enc[0] ^ 0x20]) + enc[1:]


ENGINE = Engine("bls12381", "bls12-381", ORDER, G1_LEN, G2_LEN, GT_LEN, G1_BASE, G2_BASE, G1_NULL, G2_NULL, _neg)
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

DOMAIN_G1 = b"BLS_SIG_BLS12381G1_XMD:SHA-256_SSWU_RO_NUL_"  # kilic/g1.go:17
DOMAIN_G2 = b"BLS_SIG_BLS12381G2_XMD:SHA-256_SSWU_RO_NUL_"  # kilic/g2.go:18


def _batch_hash(group: int, msgs, dst: bytes):
    import ctypes

    import numpy as np

    from .._lib import check, load
    from ._engine import _is_torch, _stream

    lib = load()
    w = G1_LEN if group == 1 else G2_LEN
    dbuf = ctypes.create_string_buffer(bytes(dst), len(dst)) if dst else None
    dptr = ctypes.cast(dbuf, ctypes.c_void_p) if dst else None
    if _is_torch(msgs):
        import torch

        m = msgs.contiguous()
        n, ln = m.shape[0], m.shape[1]
        out = torch.empty((n, w), dtype=torch.uint8, device=m.device)
        st = torch.empty(n, dtype=torch.uint8, device=m.device)
        fn = getattr(lib, f"kyb_bls12381_hash_g{group}_dev")
        check(fn(n, m.data_ptr(), ln, dptr, len(dst), out.data_ptr(), st.data_ptr(), _stream()), "hash_dev")
        return out, st
    if isinstance(msgs, (list, tuple)):
        ln = len(msgs[0]) if msgs else 0
        if any(len(x) != ln for x in msgs):
            raise ValueError("batch hash: messages must have equal length")
        n = len(msgs)
        buf = np.frombuffer(b"".join(msgs), dtype=np.uint8)
    else:
        a = np.ascontiguousarray(msgs, dtype=np.uint8)
        n, ln = a.shape[0], a.shape[1]
        buf = a.reshape(-1)
    buf = np.ascontiguousarray(buf) if buf.size else np.zeros(1, dtype=np.uint8)
    out = np.empty((n, w), dtype=np.uint8)
    st = np.empty(n, dtype=np.uint8)
    fn = getattr(lib, f"kyb_bls12381_hash_g{group}")
    check(fn(n, buf.ctypes.data, ln, dptr, len(dst), out.ctypes.data, st.ctypes.data), "hash")
    return out, st


def batch_hash_g1(msgs, dst: bytes = DOMAIN_G1):
    """(out, status): G1Elt.Hash for n equal-length messages (kilic/g1.go:161-170, RFC 9380 hash_to_curve)."""
    return _batch_hash(1, msgs, dst)


def batch_hash_g2(msgs, dst: bytes = DOMAIN_G2):
    return _batch_hash(2, msgs, dst)


def batch_verify_g1(pubkeys, msgs, sigs, dst: bytes = DOMAIN_G1, flags: int = 0):
    return _batch_verify(1, pubkeys, msgs, sigs, dst, flags)


def batch_verify_g2(pubkeys, msgs, sigs, dst: bytes = DOMAIN_G2, flags: int = 0):
    """As batch_verify_g1 for the scheme with signatures on G2 and keys on G1 (NewSchemeOnG2, bls.go:48-58)."""
    return _batch_verify(2, pubkeys, msgs, sigs, dst, flags)


def batch_verify_g1_same_key(pubkey, msgs, sigs, dst: bytes = DOMAIN_G1, flags: int = 0):
    """(ok, status): N x bls.Verify under ONE public key (sign/bls/bls.go:82-96 in a loop with the same X: a drand
    chain, sign/tbls/tbls.go:100-107) -- both Miller loops from line tables (kyb_bls12381_verify_g1_same_key).
    pubkey: 96 bytes (192 with F_UNCOMPRESSED) or a CUDA tensor of that size when msgs / sigs are CUDA tensors."""
    import ctypes

    import numpy as np

    from .._lib import check, load
    from ._engine import F_UNCOMPRESSED, _host, _is_torch, _stream, pack_fixed

    wk, wsig = (192, 96) if flags & F_UNCOMPRESSED else (96, 48)
    lib = load()
    dbuf = ctypes.create_string_buffer(bytes(dst), len(dst)) if dst else None
    dptr = ctypes.cast(dbuf, ctypes.c_void_p) if dst else None
    if _is_torch(msgs):
        import torch

        m, s = msgs.contiguous(), sigs.contiguous().view(-1, wsig)
        k = pubkey if _is_torch(pubkey) else torch.from_numpy(np.frombuffer(bytes(pubkey), dtype=np.uint8).copy())
        k = k.to(m.device).contiguous().view(-1)
        n, ln = m.shape[0], m.shape[1]
        if k.numel() != wk or s.shape[0] != n:
            raise ValueError(f"batch_verify_same_key: key of {k.numel()} bytes, {n} messages, {s.shape[0]} signatures")
        ok = torch.empty(n, dtype=torch.uint8, device=m.device)
        st = torch.empty(n, dtype=torch.uint8, device=m.device)
        check(lib.kyb_bls12381_verify_g1_same_key_dev(n, k.data_ptr(), m.data_ptr(), ln, dptr, len(dst), s.data_ptr(), ok.data_ptr(),
                                                      st.data_ptr(), flags, _stream()), "kyb_bls12381_verify_g1_same_key_dev")
        return ok, st
    if isinstance(msgs, (list, tuple)):
        ln = len(msgs[0]) if msgs else 0
        if any(len(x) != ln for x in msgs):
            raise ValueError("batch_verify_same_key: messages must have equal length")
        mb = np.frombuffer(b"".join(msgs), dtype=np.uint8)
        n = len(msgs)
    else:
        a = np.ascontiguousarray(msgs, dtype=np.uint8)
        n, ln = a.shape[0], a.shape[1]
        mb = a.reshape(-1)
    mb = np.ascontiguousarray(mb) if mb.size else np.zeros(1, dtype=np.uint8)
    s, bad_s = pack_fixed(sigs, wsig) if isinstance(sigs, (list, tuple)) else (_host(sigs, wsig), [])
    kb = bytes(pubkey)
    if s.shape[0] != n:
        raise ValueError(f"batch_verify_same_key: {n} messages, {s.shape[0]} signatures")
    ok = np.empty(n, dtype=np.uint8)
    st = np.empty(n, dtype=np.uint8)
    if len(kb) != wk:  # a key of the wrong length fails every element, as UnmarshalBinary would fail the one key
        ok[:], st[:] = 0, 1
        return ok, st
    check(lib.kyb_bls12381_verify_g1_same_key(n, kb, mb.ctypes.data, ln, dptr, len(dst), s.ctypes.data, ok.ctypes.data,
                                              st.ctypes.data, flags), "kyb_bls12381_verify_g1_same_key")
    for i in bad_s:  # precedence as the header documents it: the key's verdict first, then the signature's
        ok[i] = 0
        if st[i] == 0:
            st[i] = 1  # KYB_ST_BAD_POINT
    return ok, st


def batch_verify_g1_same_msg(pubkeys, msg, sigs, dst: bytes = DOMAIN_G1, flags: int = 0):
    """(ok, status): ok[i] = bls.Verify(pubkeys[i], msg, sigs[i]) for ONE message -- the verification loop of
    tbls.Recover (sign/tbls/tbls.go:118-131: every partial signature signs the same msg under its own public share):
    H(msg) is hashed once per call (kyb_bls12381_verify_g1_same_msg), the rest is batch_verify_g1.
    msg: bytes, or a 1-D CUDA uint8 tensor when pubkeys / sigs are CUDA tensors."""
    import ctypes

    import numpy as np

    from .._lib import check, load
    from ._engine import F_UNCOMPRESSED, _host, _is_torch, _stream, pack_fixed

    wk, wsig = (192, 96) if flags & F_UNCOMPRESSED else (96, 48)
    lib = load()
    dbuf = ctypes.create_string_buffer(bytes(dst), len(dst)) if dst else None
    dptr = ctypes.cast(dbuf, ctypes.c_void_p) if dst else None
    if _is_torch(pubkeys):
        import torch

        p, s = pubkeys.contiguous().view(-1, wk), sigs.contiguous().view(-1, wsig)
        m = msg if _is_torch(msg) else torch.from_numpy(np.frombuffer(bytes(msg), dtype=np.uint8).copy())
        m = m.to(p.device).contiguous().view(-1)
        n, ln = p.shape[0], int(m.numel())
        if s.shape[0] != n:
            raise ValueError(f"batch_verify_same_msg: {n} public keys, {s.shape[0]} signatures")
        if ln == 0:
            m = torch.zeros(1, dtype=torch.uint8, device=p.device)
        ok = torch.empty(n, dtype=torch.uint8, device=p.device)
        st = torch.empty(n, dtype=torch.uint8, device=p.device)
        check(lib.kyb_bls12381_verify_g1_same_msg_dev(n, p.data_ptr(), m.data_ptr(), ln, dptr, len(dst), s.data_ptr(), ok.data_ptr(),
                                                      st.data_ptr(), flags, _stream()), "kyb_bls12381_verify_g1_same_msg_dev")
        return ok, st
    mb = bytes(msg)
    p, bad_p = pack_fixed(pubkeys, wk) if isinstance(pubkeys, (list, tuple)) else (_host(pubkeys, wk), [])
    s, bad_s = pack_fixed(sigs, wsig) if isinstance(sigs, (list, tuple)) else (_host(sigs, wsig), [])
    n = p.shape[0]
    if s.shape[0] != n:
        raise ValueError(f"batch_verify_same_msg: {n} public keys, {s.shape[0]} signatures")
    ok = np.empty(n, dtype=np.uint8)
    st = np.empty(n, dtype=np.uint8)
    check(lib.kyb_bls12381_verify_g1_same_msg(n, p.ctypes.data, mb if mb else None, len(mb), dptr, len(dst), s.ctypes.data, ok.ctypes.data,
                                              st.ctypes.data, flags), "kyb_bls12381_verify_g1_same_msg")
    for i in bad_p + bad_s:  # a wrong-length key / signature fails alone (status 1 unless the native call already said more)
        ok[i] = 0
        if st[i] == 0:
            st[i] = 1
    return ok, st


def _batch_verify(sig_group: int, pubkeys, msgs, sigs, dst: bytes, flags: int):
    """(ok, status): N x bls.Verify (sign/bls/bls.go:82-96; signatures on G1, keys on G2) fused in ONE kernel:
    hash_to_curve, both unmarshal checks, two Miller loops sharing their squarings and one final exponentiation
    per lane.  msgs: (n, msg_len) uint8 array / CUDA tensor or list of equal-length bytes.  flags: F_TRUSTED(0) for
    public keys validated before (the usual case: keys are unmarshalled once), F_TRUSTED(1) for the signatures,
    F_UNCOMPRESSED for uncompressed-affine keys and signatures."""
    import ctypes

    import numpy as np

    from .._lib import check, load
    from ._engine import F_UNCOMPRESSED, _host, _is_torch, _stream, pack_fixed

    wk, wsig = (96, 48) if sig_group == 1 else (48, 96)
    if flags & F_UNCOMPRESSED:
        wk, wsig = 2 * wk, 2 * wsig
    name = f"kyb_bls12381_verify_g{sig_group}"

    lib = load()
    dbuf = ctypes.create_string_buffer(bytes(dst), len(dst)) if dst else None
    dptr = ctypes.cast(dbuf, ctypes.c_void_p) if dst else None
    if _is_torch(msgs):
        import torch

        m, p, s = msgs.contiguous(), pubkeys.contiguous().view(-1, wk), sigs.contiguous().view(-1, wsig)
        n, ln = m.shape[0], m.shape[1]
        if p.shape[0] != n or s.shape[0] != n:
            raise ValueError(f"batch_verify: {n} messages, {p.shape[0]} public keys, {s.shape[0]} signatures")
        ok = torch.empty(n, dtype=torch.uint8, device=m.device)
        st = torch.empty(n, dtype=torch.uint8, device=m.device)
        check(getattr(lib, name + "_dev")(n, p.data_ptr(), m.data_ptr(), ln, dptr, len(dst), s.data_ptr(), ok.data_ptr(),
                                          st.data_ptr(), flags, _stream()), name + "_dev")
        return ok, st
    if isinstance(msgs, (list, tuple)):
        ln = len(msgs[0]) if msgs else 0
        if any(len(x) != ln for x in msgs):
            raise ValueError("batch_verify: messages must have equal length")
        mb = np.frombuffer(b"".join(msgs), dtype=np.uint8)
        n = len(msgs)
    else:
        a = np.ascontiguousarray(msgs, dtype=np.uint8)
        n, ln = a.shape[0], a.shape[1]
        mb = a.reshape(-1)
    mb = np.ascontiguousarray(mb) if mb.size else np.zeros(1, dtype=np.uint8)
    # per-element lists are packed element by element: a wrong-length key / signature (attacker-supplied) fails alone
    p, bad_p = pack_fixed(pubkeys, wk) if isinstance(pubkeys, (list, tuple)) else (_host(pubkeys, wk), [])
    s, bad_s = pack_fixed(sigs, wsig) if isinstance(sigs, (list, tuple)) else (_host(sigs, wsig), [])
    if p.shape[0] != n or s.shape[0] != n:
        raise ValueError(f"batch_verify: {n} messages, {p.shape[0]} public keys, {s.shape[0]} signatures")
    ok = np.empty(n, dtype=np.uint8)
    st = np.empty(n, dtype=np.uint8)
    check(getattr(lib, name)(n, p.ctypes.data, mb.ctypes.data, ln, dptr, len(dst), s.ctypes.data, ok.ctypes.data,
                             st.ctypes.data, flags), name)
    for i in bad_p + bad_s:
        ok[i], st[i] = 0, 1  # KYB_ST_BAD_POINT
    return ok, st
