"""Build + load tests/host_harness.cpp (the device headers compiled for the CPU; test infrastructure)."""
import ctypes as C
import hashlib
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "host_harness.cpp")
OUT = os.path.join(ROOT, "tests", "_build", "libhostharness.so")
_lib = None


def _deps_digest() -> str:
    """Content hash of the harness source + every device header (mtimes do not survive a snapshot copy)."""
    csrc = os.path.join(ROOT, "kyber_amd", "csrc")
    deps = [SRC] + sorted(os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".cuh", ".h", ".inc")) and not f.startswith("tower_vm_"))
    h = hashlib.sha256()
    for d in deps:
        h.update(os.path.basename(d).encode())
        with open(d, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def lib():
    global _lib
    if _lib is None:
        stamp, digest = OUT + ".sha256", _deps_digest()
        fresh = os.path.exists(OUT) and os.path.exists(stamp) and open(stamp).read().strip() == digest
        if not fresh:
            os.makedirs(os.path.dirname(OUT), exist_ok=True)
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-shared", "-fPIC", "-o", OUT, SRC])
            with open(stamp, "w") as f:
                f.write(digest)
        _lib = C.CDLL(OUT)
    return _lib


_lib_audit = None


def lib_audit():
    """The same harness built with -DKYB_FE_AUDIT: Ed25519 field elements carry a magnitude bound that every
    multiplication checks in with (fe25519.cuh), and -DKYB_LZ_AUDIT: the lazy limb elements of fp_limbs.cuh
    carry their bound as a multiple of p and every product / subtraction checks its precondition.  Rebuilt on every first use (a few seconds)."""
    global _lib_audit
    if _lib_audit is None:
        out = OUT.replace("libhostharness.so", "libhostharness_audit.so")
        os.makedirs(os.path.dirname(out), exist_ok=True)
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-pthread", "-DKYB_FE_AUDIT", "-DKYB_LZ_AUDIT", "-shared", "-fPIC", "-o", out, SRC])
        _lib_audit = C.CDLL(out)
    return _lib_audit


def call(name, *bufs_and_ints, out_sizes=(), audit=False):
    """Call harness function `name`; bytes args are inputs, ints pass through, outputs appended."""
    outs = [C.create_string_buffer(n) for n in out_sizes]
    args = []
    for a in bufs_and_ints:
        args.append(C.c_int(a) if isinstance(a, int) else C.c_char_p(bytes(a)))
    fn = getattr(lib_audit() if audit else lib(), name)
    fn.restype = C.c_int
    rc = fn(*args, *outs)
    return (rc, *[o.raw for o in outs])
