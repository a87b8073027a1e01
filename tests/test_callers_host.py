"""Host logic of the caller-level mirrors (no GPU): BLAKE2Xs coefficients of sign/bdn against the reference's fixtures."""
import json
import os

from kyber_amd.sign import bdn
from oracle import bn256 as O


def test_bdn_coefficients_match_reference_fixture(golden_dir):
    G = json.load(open(os.path.join(golden_dir, "bn256.json")))
    pubs = [O.g2_marshal(O.g2_mul(i, O.G2_GEN)) for i in (1, 2, 3)]
    assert [f"{c:032x}" for c in bdn.hash_point_to_r(pubs)] == G["bdn_coefs"]
