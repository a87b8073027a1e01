"""ctypes access to oracle/libkyber_oracle.so (test infrastructure only)."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATH = os.path.join(ROOT, "oracle", "libkyber_oracle.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(PATH):
            import subprocess

            subprocess.check_call(["make"], cwd=os.path.join(ROOT, "oracle"))
        l = C.CDLL(PATH)
        vp, sz, i = C.c_void_p, C.c_size_t, C.c_int
        l.ora_ed25519_mul_base.argtypes = [sz, vp, vp, i]
        l.ora_ed25519_mul_base.restype = None
        l.ora_ed25519_mul.argtypes = [sz, vp, vp, vp, vp, i, i]
        l.ora_ed25519_mul.restype = None
        l.ora_ed25519_msm.argtypes = [sz, vp, vp, vp]
        l.ora_ed25519_msm.restype = C.c_long
        for name in ("ora_bn256_pair", "ora_bn256_g1_mul_sum", "ora_bn256_g1_mul", "ora_bn256_g2_mul", "ora_bls12381_g1_mul_sum",
                     "ora_bls12381_pair", "ora_bls12381_g1_mul", "ora_bls12381_g2_mul", "ora_bls12381_g1_mul_sum_compressed",
                     "ora_bls12381_pair_compressed"):
            f = getattr(l, name)
            f.argtypes = [sz, vp, vp, vp, vp, i]
            f.restype = None
        _lib = l
    return _lib


def ed_mul_base(scalars: np.ndarray, threads: int = 0) -> np.ndarray:
    s = np.ascontiguousarray(scalars, dtype=np.uint8).reshape(-1, 32)
    out = np.empty_like(s)
    lib().ora_ed25519_mul_base(len(s), s.ctypes.data, out.ctypes.data, threads or (os.cpu_count() or 1))
    return out


def ed_mul(scalars, points, vartime=False, threads: int = 0):
    s = np.ascontiguousarray(scalars, dtype=np.uint8).reshape(-1, 32)
    p = np.ascontiguousarray(points, dtype=np.uint8).reshape(-1, 32)
    out = np.empty_like(s)
    st = np.empty(len(s), dtype=np.uint8)
    lib().ora_ed25519_mul(len(s), s.ctypes.data, p.ctypes.data, out.ctypes.data, st.ctypes.data, int(vartime),
                          threads or (os.cpu_count() or 1))
    return out, st


def ed_msm(scalars, points):
    s = np.ascontiguousarray(scalars, dtype=np.uint8).reshape(-1, 32)
    p = np.ascontiguousarray(points, dtype=np.uint8).reshape(-1, 32)
    out = np.empty(32, dtype=np.uint8)
    rc = lib().ora_ed25519_msm(len(s), s.ctypes.data, p.ctypes.data, out.ctypes.data)
    return out, rc


def bn256_pair(g1, g2, threads: int = 0):
    """(gt, status): oracle/bn256_ref.c, the reference's optimalAte restated in C (64 + 128 bytes -> 384 bytes)."""
    a = np.ascontiguousarray(np.frombuffer(g1, dtype=np.uint8) if isinstance(g1, (bytes, bytearray)) else g1, dtype=np.uint8).reshape(-1, 64)
    b = np.ascontiguousarray(np.frombuffer(g2, dtype=np.uint8) if isinstance(g2, (bytes, bytearray)) else g2, dtype=np.uint8).reshape(-1, 128)
    out = np.empty((len(a), 384), dtype=np.uint8)
    st = np.empty(len(a), dtype=np.uint8)
    lib().ora_bn256_pair(len(a), a.ctypes.data, b.ctypes.data, out.ctypes.data, st.ctypes.data, threads or (os.cpu_count() or 1))
    return out, st


def bn256_g1_mul(scalars, points, threads: int = 0):
    s = np.ascontiguousarray(scalars, dtype=np.uint8).reshape(-1, 32)
    p = np.ascontiguousarray(points, dtype=np.uint8).reshape(-1, 64)
    out = np.empty((len(s), 64), dtype=np.uint8)
    st = np.empty(len(s), dtype=np.uint8)
    lib().ora_bn256_g1_mul(len(s), s.ctypes.data, p.ctypes.data, out.ctypes.data, st.ctypes.data, threads or (os.cpu_count() or 1))
    return out, st


def bn256_g1_mul_sum(scalars, points, threads: int = 0):
    """sum_i k_i P_i the reference's way: N x (Mul + Add)"""
    s = np.ascontiguousarray(scalars, dtype=np.uint8).reshape(-1, 32)
    p = np.ascontiguousarray(points, dtype=np.uint8).reshape(-1, 64)
    out = np.empty(64, dtype=np.uint8)
    st = np.zeros(len(s), dtype=np.uint8)
    lib().ora_bn256_g1_mul_sum(len(s), s.ctypes.data, p.ctypes.data, out.ctypes.data, st.ctypes.data, threads or (os.cpu_count() or 1))
    return out, st


def bls12381_g1_mul_sum(scalars, points_unc, threads: int = 0):
    """sum_i k_i P_i, N x (Mul + Add) on BLS12-381 G1; ZCash uncompressed points in, uncompressed point out (96 bytes)"""
    s = np.ascontiguousarray(scalars, dtype=np.uint8).reshape(-1, 32)
    p = np.ascontiguousarray(points_unc, dtype=np.uint8).reshape(-1, 96)
    out = np.empty(96, dtype=np.uint8)
    st = np.zeros(len(s), dtype=np.uint8)
    lib().ora_bls12381_g1_mul_sum(len(s), s.ctypes.data, p.ctypes.data, out.ctypes.data, st.ctypes.data, threads or (os.cpu_count() or 1))
    return out, st


def bls12381_pair(g1_unc, g2_unc, threads: int = 0):
    """(gt, status): oracle/bls12381_pair_ref.c, n x Suite.Pair on BLS12-381; ZCash uncompressed points in (96 / 192
    bytes), kilic's 576-byte GT encoding out"""
    a = np.ascontiguousarray(np.frombuffer(g1_unc, dtype=np.uint8) if isinstance(g1_unc, (bytes, bytearray)) else g1_unc, dtype=np.uint8).reshape(-1, 96)
    b = np.ascontiguousarray(np.frombuffer(g2_unc, dtype=np.uint8) if isinstance(g2_unc, (bytes, bytearray)) else g2_unc, dtype=np.uint8).reshape(-1, 192)
    out = np.empty((len(a), 576), dtype=np.uint8)
    st = np.empty(len(a), dtype=np.uint8)
    lib().ora_bls12381_pair(len(a), a.ctypes.data, b.ctypes.data, out.ctypes.data, st.ctypes.data, threads or (os.cpu_count() or 1))
    return out, st


def _mul(fn, scalars, points, plen, threads):
    s = np.ascontiguousarray(np.frombuffer(scalars, dtype=np.uint8) if isinstance(scalars, (bytes, bytearray)) else scalars, dtype=np.uint8).reshape(-1, 32)
    p = np.ascontiguousarray(np.frombuffer(points, dtype=np.uint8) if isinstance(points, (bytes, bytearray)) else points, dtype=np.uint8).reshape(-1, plen)
    assert len(s) == len(p)
    out = np.empty((len(s), plen), dtype=np.uint8)
    st = np.empty(len(s), dtype=np.uint8)
    fn(len(s), s.ctypes.data, p.ctypes.data, out.ctypes.data, st.ctypes.data, threads or (os.cpu_count() or 1))
    return out, st


def bn256_g2_mul(scalars, points, threads: int = 0):
    """(out, status): pointG2.Mul element-wise (twist.go:172-185 restated in oracle/bn256_ref.c), 128-byte points"""
    return _mul(lib().ora_bn256_g2_mul, scalars, points, 128, threads)


def bls12381_g1_mul(scalars, points, threads: int = 0):
    """(out, status): G1Elt.Mul element-wise, 48-byte compressed points in and out (oracle/bls12381_pair_ref.c)"""
    return _mul(lib().ora_bls12381_g1_mul, scalars, points, 48, threads)


def bls12381_g2_mul(scalars, points, threads: int = 0):
    """(out, status): G2Elt.Mul element-wise, 96-byte compressed points in and out"""
    return _mul(lib().ora_bls12381_g2_mul, scalars, points, 96, threads)


def bls12381_g1_mul_sum_compressed(scalars, points, threads: int = 0):
    """(sum, status): sum_i k_i P_i by N x (Mul + Add) over 48-byte compressed points; the 48-byte compressed sum"""
    s = np.ascontiguousarray(scalars, dtype=np.uint8).reshape(-1, 32)
    p = np.ascontiguousarray(points, dtype=np.uint8).reshape(-1, 48)
    out = np.empty(48, dtype=np.uint8)
    st = np.zeros(len(s), dtype=np.uint8)
    lib().ora_bls12381_g1_mul_sum_compressed(len(s), s.ctypes.data, p.ctypes.data, out.ctypes.data, st.ctypes.data, threads or host_threads())
    return out, st


def bls12381_pair_compressed(g1, g2, threads: int = 0):
    """(gt, status): n x Suite.Pair over compressed operands (48 + 96 bytes), kilic's 576-byte GT encoding out"""
    a = np.ascontiguousarray(np.frombuffer(g1, dtype=np.uint8) if isinstance(g1, (bytes, bytearray)) else g1, dtype=np.uint8).reshape(-1, 48)
    b = np.ascontiguousarray(np.frombuffer(g2, dtype=np.uint8) if isinstance(g2, (bytes, bytearray)) else g2, dtype=np.uint8).reshape(-1, 96)
    out = np.empty((len(a), 576), dtype=np.uint8)
    st = np.empty(len(a), dtype=np.uint8)
    lib().ora_bls12381_pair_compressed(len(a), a.ctypes.data, b.ctypes.data, out.ctypes.data, st.ctypes.data, threads or host_threads())
    return out, st


def host_threads() -> int:
    """threads worth starting on this host: the affinity mask capped by the cgroup CPU quota (a GPU lease shows every
    logical CPU of the machine to os.cpu_count() and grants a fraction of them)"""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = max(1, min(n, int(float(q) / float(per) + 0.5)))
    except (OSError, ValueError):
        pass
    return n
