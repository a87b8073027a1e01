"""share.PriPoly.Eval / Shares (share/poly.go:85-102) on the engine: kyb_<suite>_scalar_poly_eval against Python
big-integer Horner.  The CPU tests run the same header (scalar_field.cuh) compiled for the host."""
import random

import numpy as np
import pytest

from oracle import bls12381 as OB, bn254 as ON4, bn256 as ON, ed25519 as OE
from tests import _host_harness as hh

SUITES = {"ed25519": (0, OE.L, "little"), "bls12381": (1, OB.R, "big"), "bn256": (2, ON.ORDER, "big"), "bn254": (3, ON4.ORDER, "big")}


def _expect(q, order, coeffs, indices):
    out = []
    for i in indices:
        x, v = i + 1, 0
        for c in reversed(coeffs):  # poly.go:88-91: v = v * xi + coeffs[j]
            v = (v * x + c) % q
        out.append(v.to_bytes(32, order))
    return out


def _cases(q, rng):
    edge = [0, 1, q - 1, q, q + 1, 2**256 - 1, 2**255, (q - 1) // 2, 2**252]
    yield [c for c in edge], [0, 1, 2, 0xFFFFFFFF, 0xFFFFFFFE, 12345]
    yield [], [0, 5]
    yield [rng.randrange(q)], [0, 7, 0xFFFFFFFF]
    yield [rng.randrange(2**256) for _ in range(2)], [rng.randrange(2**32) for _ in range(9)]
    yield [rng.randrange(q) for _ in range(67)], list(range(40))


@pytest.mark.parametrize("suite", sorted(SUITES))
def test_scalar_horner_host_build_matches_bigint(suite):
    sid, q, order = SUITES[suite]
    rng = random.Random(11 + sid)
    for coeffs, indices in _cases(q, rng):
        cb = b"".join(c.to_bytes(32, order) for c in coeffs)
        ib = np.asarray(indices, dtype=np.uint32).tobytes()
        _, out = hh.call("hh_scalar_poly_eval", sid, len(indices), ib, len(coeffs), cb, out_sizes=(32 * len(indices),))
        got = [bytes(out[32 * i:32 * i + 32]) for i in range(len(indices))]
        assert got == _expect(q, order, coeffs, indices)


def test_group_orders_in_the_header_are_the_oracles():
    """scalar_field.cuh Q_* against the oracle constants (the reference's: const.go primeOrder, constants.go Order)."""
    import os
    import re

    src = open(os.path.join(os.path.dirname(__file__), "..", "kyber_amd", "csrc", "scalar_field.cuh")).read()
    for name, q in (("Q_ED25519", OE.L), ("Q_BLS12381", OB.R), ("Q_BN256", ON.ORDER), ("Q_BN254", ON4.ORDER)):
        words = re.search(name + r"\[8\] = \{([^}]*)\}", src).group(1).split(",")
        assert sum(int(w.strip().rstrip("u"), 0) << (32 * i) for i, w in enumerate(words)) == q


@pytest.mark.gpu
@pytest.mark.parametrize("suite", sorted(SUITES))
@pytest.mark.parametrize("t,n", [(1, 1), (2, 1000), (667, 1), (667, 1 << 16), (1, 1 << 16), (0, 3)])
def test_gpu_scalar_poly_eval(suite, t, n):
    """t in {1, 2, 667}, n in {1, 1000, 2^16}; sampled against the big-integer loop."""
    from kyber_amd.group import edwards25519 as ed
    from kyber_amd.pairing import bls12381, bn254, bn256

    sid, q, order = SUITES[suite]
    rng = random.Random(100 * sid + t + n)
    coeffs = [rng.randrange(q) for _ in range(t)]
    if t >= 2:
        coeffs[0], coeffs[-1] = q - 1, 2**256 - 1  # an unreduced string is taken modulo q
    indices = [rng.randrange(2**32) for _ in range(n)]
    if n >= 4:
        indices[:4] = [0, 1, 0xFFFFFFFF, 0xFFFFFFFE]
    cb = b"".join(c.to_bytes(32, order) for c in coeffs)
    fn = ed.scalar_poly_eval if suite == "ed25519" else {"bls12381": bls12381, "bn256": bn256, "bn254": bn254}[suite].ENGINE.scalar_poly_eval
    out = np.asarray(fn(cb, indices))
    assert out.shape == (n, 32)
    sample = sorted(set(list(range(min(n, 8))) + [n - 1] + [rng.randrange(n) for _ in range(24)]))
    exp = _expect(q, order, coeffs, [indices[i] for i in sample])
    assert [bytes(out[i]) for i in sample] == exp


@pytest.mark.gpu
def test_gpu_pripoly_shares_on_the_engine():
    """PriPoly.Shares(n >= DEVICE_MIN) is one launch and equals the host loop (share/poly.go:96-102)."""
    from kyber_amd.group import edwards25519 as ed
    from kyber_amd.pairing import bls12381
    from kyber_amd.share import poly

    rng = random.Random(3)
    rand = lambda k: bytes(rng.randrange(256) for _ in range(k))
    for g in (ed.NewSuite(), bls12381.NewSuite().G1()):
        pri = poly.PriPoly.new(g, 5, rand=rand)
        dev = pri.Shares(100)
        host = [pri.Eval(i) for i in range(100)]
        assert [(s.I, s.V.MarshalBinary()) for s in dev] == [(s.I, s.V.MarshalBinary()) for s in host]
