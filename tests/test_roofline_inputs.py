"""tools/roofline_inputs.py refuses a profile taken from other sources than the tree's: the
mechanism, on a temporary copy of the round's PMC summaries -- a profile whose recorded digest of one of the kernel's
sources differs is dropped with a REFUSED line; with matching digests it is kept."""
import glob
import json
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _run(d, *extra):
    return subprocess.run([sys.executable, os.path.join(ROOT, "tools", "roofline_inputs.py"), d, "r04_final", *extra],
                          capture_output=True, text=True)


def test_profile_of_other_sources_is_refused(tmp_path):
    import source_digest

    src = os.path.join(ROOT, "profiles")
    files = [f for p in ("ed", "fb") for f in glob.glob(os.path.join(src, f"r04_final_{p}_*"))]
    if not files:
        pytest.skip("no round-4 profiles in this tree")
    d = str(tmp_path)
    for f in files + [os.path.join(src, "roofline_inputs.json")]:
        shutil.copy(f, d)
    now = source_digest.digests()
    # digests as of NOW for both workloads: both kept
    for p in ("ed", "fb"):
        json.dump({"sources": now}, open(os.path.join(d, f"r04_final_{p}_meta.json"), "w"))
    r = _run(d)
    assert r.returncode == 0, r.stderr
    kept = json.load(open(os.path.join(d, "roofline_inputs.json")))["kernels"]
    assert "ed25519_mul" in kept and "bls12381_g1_commit" in kept
    assert kept["ed25519_mul"]["sources_unchanged_since_profile"] is True and kept["bls12381_g1_commit"]["valu_busy"] <= 1.0
    # the fixed-base header was different when "fb" was profiled: refused; the Ed25519 profile stays
    then = dict(now)
    then["kyber_amd/csrc/fixed_base.cuh"] = "0" * 64
    json.dump({"sources": then}, open(os.path.join(d, "r04_final_fb_meta.json"), "w"))
    r = _run(d)
    kept = json.load(open(os.path.join(d, "roofline_inputs.json")))["kernels"]
    assert "REFUSED fb" in r.stderr and "fixed_base.cuh" in r.stderr
    assert "bls12381_g1_commit" not in kept and "ed25519_mul" in kept
    # a profile without digests at all (rounds 1-3) is refused too, unless asked for
    os.remove(os.path.join(d, "r04_final_ed_meta.json"))
    r = _run(d)
    assert "REFUSED ed" in r.stderr
    r = _run(d, "--allow-stale")
    assert "ed25519_mul" in json.load(open(os.path.join(d, "roofline_inputs.json")))["kernels"]
