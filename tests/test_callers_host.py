"""Host logic of the caller-level mirrors (no GPU): BLAKE2Xs coefficients of sign/bdn and the bn256
hash-to-point, against the reference's fixtures."""
import json
import os

from kyber_amd.sign import bdn, bls
from oracle import bn256 as O


def test_bdn_coefficients_match_reference_fixture(golden_dir):
    G = json.load(open(os.path.join(golden_dir, "bn256.json")))
    pubs = [O.g2_marshal(O.g2_mul(i, O.G2_GEN)) for i in (1, 2, 3)]
    assert [f"{c:032x}" for c in bdn.hash_point_to_r(pubs)] == G["bdn_coefs"]


def test_bn256_hash_to_g1_matches_reference_fixture(golden_dir):
    G = json.load(open(os.path.join(golden_dir, "bn256.json")))
    for h in G["hash_g1"]:
        assert bls.bn256_hash_to_g1(bytes.fromhex(h["msg_hex"])).hex() == h["point"]
