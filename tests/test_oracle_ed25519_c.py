"""The C restatement (oracle/ed25519_ref.c) against the pinned Python oracle
and the reference's golden vectors.  CPU only."""
import os

import numpy as np

from oracle import ed25519 as O
from tests import _oracle_c as OC

G = os.path.join(os.path.dirname(__file__), "golden")
KAT = np.load(os.path.join(G, "ed25519_sign_input.npy"))


def test_fixed_base_all_1024_kats():
    assert (OC.ed_mul_base(KAT[:, 0]) == KAT[:, 1]).all()
    assert (OC.ed_mul_base(KAT[:, 2]) == KAT[:, 3]).all()


def test_var_base_matches_python_oracle_both_modes():
    out, st = OC.ed_mul(KAT[:, 4], KAT[:, 1])
    outv, _ = OC.ed_mul(KAT[:, 4], KAT[:, 1], vartime=True)
    assert not st.any()
    for i in range(0, len(KAT), 32):
        assert bytes(out[i]) == O.mul(bytes(KAT[i, 4]), bytes(KAT[i, 1]))
        assert bytes(outv[i]) == O.mul(bytes(KAT[i, 4]), bytes(KAT[i, 1]), vartime=True)


def test_edge_scalars_and_bad_points():
    rng = np.random.default_rng(7)
    scalars = rng.integers(0, 256, size=(64, 32), dtype=np.uint8)  # includes >= 2^255
    scalars[0] = 0
    scalars[1] = 0xFF
    scalars[2] = np.frombuffer(O.L.to_bytes(32, "little"), dtype=np.uint8)
    pts = np.repeat(KAT[:64, 1], 1, axis=0).copy()
    pts[5] = np.frombuffer((2).to_bytes(32, "little"), dtype=np.uint8)  # not on curve
    out, st = OC.ed_mul(scalars, pts)
    outv, stv = OC.ed_mul(scalars, pts, vartime=True)
    base = OC.ed_mul_base(scalars)
    for i in range(64):
        exp = O.mul(bytes(scalars[i]), bytes(pts[i]))
        if exp is None:
            assert st[i] == 1 and stv[i] == 1 and not out[i].any()
            continue
        assert st[i] == 0
        assert bytes(out[i]) == exp
        assert bytes(outv[i]) == O.mul(bytes(scalars[i]), bytes(pts[i]), vartime=True)
        assert bytes(base[i]) == O.mul_base(bytes(scalars[i]))


def test_msm_matches_python():
    s, p = KAT[:24, 4], KAT[:24, 1]
    out, rc = OC.ed_msm(s, p)
    assert rc == 0
    assert bytes(out) == O.msm([bytes(x) for x in s], [bytes(x) for x in p])
