"""Build + load tests/host_harness.cpp (the device headers compiled for the CPU; test infrastructure)."""
import ctypes as C
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "host_harness.cpp")
OUT = os.path.join(ROOT, "tests", "_build", "libhostharness.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        deps = [SRC] + [os.path.join(ROOT, "kyber_amd", "csrc", f) for f in os.listdir(os.path.join(ROOT, "kyber_amd", "csrc"))
                        if f.endswith((".cuh", ".h"))]
        if not os.path.exists(OUT) or os.path.getmtime(OUT) < max(os.path.getmtime(d) for d in deps):
            os.makedirs(os.path.dirname(OUT), exist_ok=True)
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", OUT, SRC])
        _lib = C.CDLL(OUT)
    return _lib


def call(name, *bufs_and_ints, out_sizes=()):
    """Call harness function `name`; bytes args are inputs, ints pass through, outputs appended."""
    outs = [C.create_string_buffer(n) for n in out_sizes]
    args = []
    for a in bufs_and_ints:
        args.append(C.c_int(a) if isinstance(a, int) else C.c_char_p(bytes(a)))
    fn = getattr(lib(), name)
    fn.restype = C.c_int
    rc = fn(*args, *outs)
    return (rc, *[o.raw for o in outs])
