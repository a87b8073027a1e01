"""Whole-batch digests at BASELINE.json's config sizes: every output of the batch against the C oracle's, SHA-256 over
the concatenation (SURVEY.md section 8d "Correctness at scale") -- not the sampled KAT lanes of test_gpu_full_size.py.
Reference pattern: pairing/bn256/suite_test.go:231-259 (pairing results compared whole), twist.go:172-185 and
kilic/g1.go:110-116 / g2.go (the multiplications), share/poly.go:340-348 (the N x (Mul + Add) sum the MSM replaces).
The host side of this file is ~1 minute of oracle time on the GPU box's cores."""
import pytest

from tests import _full_digest as FD

pytestmark = pytest.mark.gpu


def _report(rec):
    return {k: v for k, v in rec.items() if k not in ("sha256", "what")}


def test_ed25519_config1_whole_batch_digest():
    """configs[1]: 2^20 variable-base + 2^20 fixed-base outputs (and the 2^20 input points, fixed-base too)"""
    r = FD.ed25519_config1(1 << 20)
    assert r["outputs_match"] and r["outputs_compared"] == 3 << 20, _report(r)


def test_bls12381_config2_msm_against_the_reference_shaped_sum():
    """configs[2]: 2^20-point G1 MSM == the oracle's 2^20 x (Mul + Add), every calling convention"""
    r = FD.bls12381_g1_msm(1 << 20)
    assert r["outputs_match"] and r["outputs_compared"] == 1 << 20, _report(r)


@pytest.mark.parametrize("leg", ["pair", "g1_mul", "g2_mul"])
def test_bls12381_config3_whole_batch_digest(leg):
    """configs[3]: all 2^16 Suite.Pair GT encodings / G1Elt.Mul / G2Elt.Mul outputs"""
    r = FD.pairing_suite("bls12381", 1 << 16, legs=(leg,))
    assert r[leg]["outputs_match"] and r[leg]["outputs_compared"] == 1 << 16, _report(r[leg])


@pytest.mark.parametrize("leg", ["pair", "g1_mul", "g2_mul"])
def test_bn256_config4_whole_batch_digest(leg):
    """configs[4]: all 2^18 Suite.Pair GT encodings / pointG1.Mul / pointG2.Mul outputs, bit-exact against the
    restatement of pairing/bn256"""
    r = FD.pairing_suite("bn256", 1 << 18, legs=(leg,))
    assert r[leg]["outputs_match"] and r[leg]["outputs_compared"] == 1 << 18, _report(r[leg])
