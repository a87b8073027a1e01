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
