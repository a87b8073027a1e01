"""Device constants (kyber_amd/csrc/*.inc) re-derived independently."""
import os
import re

from oracle import ed25519 as O

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "kyber_amd", "csrc")


def _limbs_to_int(ls):
    x, off = 0, 0
    for i, l in enumerate(ls):
        x += l << off
        off += 25 if i & 1 else 26
    return x


def test_ed25519_consts():
    src = open(os.path.join(CSRC, "ed25519_consts.inc")).read()
    vals = {}
    for name, body in re.findall(r"fe fe_(\w+)\(\) \{ return fe\{\{([^}]*)\}\}", src):
        ls = [int(x) for x in body.split(",")]
        assert len(ls) == 10
        # limb bounds the field code relies on
        for i, l in enumerate(ls):
            assert 0 <= l < (1 << (25 if i & 1 else 26))
        vals[name] = _limbs_to_int(ls)
    assert vals["d"] == O.D
    assert vals["d2"] == 2 * O.D % O.P
    assert vals["sqrtm1"] == O.SQRT_M1
    assert (vals["bx"], vals["by"]) == O.B
    assert vals["bt"] == O.B[0] * O.B[1] % O.P
