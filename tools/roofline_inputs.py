#!/usr/bin/env python3
"""profiles/roofline_inputs.json from the rocprofv3 PMC summaries of a round (tools/gpu/r02_final.sh writes them,
tools/rocpd_summary.py turns the rocpd databases into text): VALU-busy and HBM traffic per launch of the dominant
kernels, the numbers bench.py quotes next to its live timings.

usage: tools/roofline_inputs.py profiles r02_final [prefix=tag ...]   (e.g. msm_bls=r03_close2: that prefix's files
come from another evidence run -- kernels re-profiled after the main run)

Reading the counters (profiles/README.md): SQ_* are quad-cycles summed over all waves, GRBM_GUI_ACTIVE is cycles summed
over the 8 XCDs, FETCH_SIZE / WRITE_SIZE are KiB (FETCH_SIZE may under-count wide coalesced reads 2x on gfx950,
MI355X_MICROARCH.md; left uncorrected).  VALU-busy = 4 SQ_ACTIVE_INST_VALU / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)."""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import source_digest  # noqa: E402

ALLOW_STALE = "--allow-stale" in sys.argv
argv = [a for a in sys.argv if a != "--allow-stale"]
d, tag = argv[1], argv[2]
tag_of = dict(a.split("=", 1) for a in argv[3:])
NOW = source_digest.digests()
C = "kyber_amd/csrc/"
# the sources a kernel is compiled from (beyond hd.h / context.h / include/kyber_hip.h, which every kernel shares and
# whose edits are interface-level): a profile is stale once any of them differs from what the profiled binary was built from
COMMON_PAIRING = [C + "mont.cuh", C + "curve.cuh", C + "tower.cuh", C + "fp_limbs.cuh", C + "jac_lazy.cuh"]
SOURCES = {
    "ed": [C + "ed25519.hip", C + "fe25519.cuh", C + "ge25519.cuh"],
    "bls12381": [C + "bls12381_pair.hip", C + "tower_vm.cuh", C + "gen_tower_vm.py", C + "bls12381_tvm.h", C + "bls12381_prep.hip", C + "bls12381.cuh"] + COMMON_PAIRING,
    "verify": [C + "bls12381_pair.hip", C + "tower_vm.cuh", C + "gen_tower_vm.py", C + "bls12381_tvm.h", C + "bls12381_prep.hip", C + "bls12381.cuh", C + "bls12381_h2c.cuh",
               C + "bls12381_keylines.cuh", C + "rowfp.cuh"] + COMMON_PAIRING,
    "gtmul": [C + "bls12381_pair.hip", C + "bn256_pair.hip", C + "bn254_pair.hip", C + "bn_pair.inc", C + "tower_vm.cuh", C + "gen_tower_vm.py",
              C + "bls12381_tvm.h"],
    "bn256": [C + "bn256_pair.hip", C + "bn_pair.inc", C + "tower_vm.cuh", C + "gen_tower_vm.py", C + "bn256.cuh", C + "bn_suite.inc", C + "bn256.hip",
              C + "pairing_abi.cuh"] + COMMON_PAIRING,
    "bn254": [C + "bn254_pair.hip", C + "bn_pair.inc", C + "tower_vm.cuh", C + "gen_tower_vm.py", C + "bn254.cuh", C + "bn_suite.inc", C + "bn254.hip",
              C + "pairing_abi.cuh"] + COMMON_PAIRING,
    "mul": [C + "bls12381.hip", C + "bls12381_lvm.cuh", C + "lane_vm.cuh", C + "gen_lane_vm.py", C + "bls12381.cuh", C + "pairing_abi.cuh",
            C + "bls12381_unm2.hip", C + "bls12381_g1split.hip"] + COMMON_PAIRING,
    "mulperlane": [C + "bls12381.hip", C + "bls12381.cuh", C + "pairing_abi.cuh"] + COMMON_PAIRING,
    "msm_bls": [C + "bls12381_msm.hip", C + "bls12381_msm_codec.cuh", C + "msm_adapters.h", C + "msm.cuh", C + "msm_ws.cuh", C + "coop_slots.cuh", C + "bls12381.cuh", C + "rowfp.cuh"] + COMMON_PAIRING,
    "fb": [C + "fixed_base.cuh", C + "bls12381_fb.cuh", C + "pairing_abi.cuh", C + "bls12381_fb.hip", C + "bls12381.cuh", C + "coop_slots.cuh", C + "rowfp.cuh"] + COMMON_PAIRING,
}


def stale(prefix, tag):
    """names of the kernel's sources that changed since the profile `tag` was taken (None: the profile carries no
    source digests at all -- rounds 1-3)"""
    meta = os.path.join(d, f"{tag}_{prefix}_meta.json")
    if not os.path.exists(meta):
        return None
    then = json.load(open(meta))["sources"]
    return [f for f in SOURCES[prefix] if then.get(f) != NOW.get(f)]


def counters(path):
    out = {}
    if not os.path.exists(path):
        return out
    for line in open(path):
        f = [x.strip() for x in line.split("|")]
        if len(f) == 5 and re.fullmatch(r"[A-Z_0-9]+", f[1]):
            out.setdefault(f[0], {})[f[1]] = float(f[3])
    return out


def kernel(cs, sub):
    """counters of the first kernel whose name contains `sub`; `sub` may list alternatives with '|' (a kernel whose template
    gained a parameter: profiles of earlier rounds carry the shorter name)"""
    for alt in sub.split("|"):
        for k, v in cs.items():
            if alt in k:
                return v
    return {}


def entry(prefix, sub, units):
    tag = tag_of.get(prefix, argv[2])
    sq = kernel(counters(os.path.join(d, f"{tag}_{prefix}_sq.txt")), sub)
    fe = kernel(counters(os.path.join(d, f"{tag}_{prefix}_fetch.txt")), sub)
    wr = kernel(counters(os.path.join(d, f"{tag}_{prefix}_write.txt")), sub)
    if not sq:
        return None
    changed = stale(prefix, tag)
    if changed is None or changed:
        why = "no source digests recorded with it" if changed is None else "changed since: " + ", ".join(changed)
        if not ALLOW_STALE:
            print(f"REFUSED {prefix} ({sub}): profile {tag} is of another binary ({why})", file=sys.stderr)
            return None
    busy = 4 * sq["SQ_ACTIVE_INST_VALU"] / (1024 * sq["GRBM_GUI_ACTIVE"] / 8)
    e = {"units_per_launch": units,
         # GRBM_GUI_ACTIVE is summed over the 8 XCDs, whose busy intervals differ by a percent: a saturated kernel can read
         # 1.00x; the raw ratio is kept beside the clamped figure
         "valu_busy": min(1.0, busy), "valu_busy_raw": busy,
         "valu_insts_per_unit": sq["SQ_INSTS_VALU"] * 64 / units,
         "wait_share_of_wave_cycles": sq["SQ_WAIT_ANY"] / sq["SQ_WAVE_CYCLES"],
         "source": f"profiles/{tag}_{prefix}_{{sq,fetch,write}}.txt ({next((a for a in sub.split('|') if kernel(counters(os.path.join(d, f'{tag}_{prefix}_sq.txt')), a)), sub)})",
         "sources_unchanged_since_profile": changed == []}
    if fe and wr:
        e["hbm_bytes_per_launch"] = (fe["FETCH_SIZE"] + wr["WRITE_SIZE"]) * 1024
        e["fetch_bytes"], e["write_bytes"] = fe["FETCH_SIZE"] * 1024, wr["WRITE_SIZE"] * 1024
    return e


old = json.load(open(os.path.join(d, "roofline_inputs.json")))
res = {"_comment": "Inputs bench.py reads for its roofline objects; every number comes from a file in this directory "
                   "(tools/roofline_inputs.py).",
       "imad_peak_lane_ops_per_s": old["imad_peak_lane_ops_per_s"], "imad_peak_source": old["imad_peak_source"],
       "kernels": {}}
for key, prefix, sub, units in (("ed25519_mul", "ed", "ed25519_mul_kernel<true, false>|ed25519_mul_kernel<true>", 1 << 20),
                                ("bls12381_pair", "bls12381", "bls12381_tvm_kernel<0>", 1 << 16),
                                ("bls12381_check", "bls12381", "bls12381_tvm_kernel<1>", 1 << 16),
                                ("bls12381_verify", "verify", "bls12381_tvm_kernel<2>", 1 << 16),
                                ("bn256_pair", "bn256", "bn256_tvm_kernel<0>", 1 << 18),
                                # (bn256 ValidatePairing: the product-form program is kernel <2>; <1> only decides
                                # lanes whose joint Miller value was zero and returns at once otherwise)
                                ("bn256_check", "bn256", "bn256_tvm_kernel<2>", 1 << 18),
                                ("bn256_g1_mul", "bn256", "kyb::bn256_g1_mul_kernel", 1 << 18),
                                ("bn256_g2_mul", "bn256", "kyb::bn256_g2_mul_kernel", 1 << 18),
                                ("bn254_g1_mul", "bn254", "kyb::bn254_g1_mul_kernel", 1 << 18),
                                ("bn254_g2_mul", "bn254", "kyb::bn254_g2_mul_kernel", 1 << 18),
                                ("bn254_pair", "bn254", "bn254_tvm_kernel<0>", 1 << 18),
                                ("bn254_check", "bn254", "bn254_tvm_kernel<1>", 1 << 18),
                                # round 3: the lane machine's ladders (a G2 element is two lanes), the per-lane kernels of the
                                # same probe with KYB_LVM_MIN huge, the MSM's accumulate stage
                                # (2^16 G1 elements are one wave per SIMD: the dispatch rule keeps them on the per-lane kernel)
                                ("bls12381_g1_mul", "mul", "kyb::bls12381_g1_mul_kernel", 1 << 16),
                                ("bls12381_g2_mul", "mul", "bls12381_lvm_mul_kernel<true", 1 << 16),
                                ("bls12381_verifyk", "verify", "bls12381_tvm_kernel<3>", 1 << 16),
                                ("bls12381_gt_mul", "gtmul", "bls12381_gtmul_kernel", 1 << 16),
                                ("bn256_gt_mul", "gtmul", "bn256_gtmul_kernel", 1 << 16),
                                ("bn254_gt_mul", "gtmul", "bn254_gtmul_kernel", 1 << 16),
                                ("bls12381_g1_mul_perlane", "mulperlane", "bls12381_g1_mul_kernel", 1 << 16),
                                ("bls12381_g2_mul_perlane", "mulperlane", "bls12381_g2_mul_kernel", 1 << 16),
                                ("bls12381_g1_msm", "msm_bls", "accumulate_kernel", 1 << 20),
                                # same-base batches through the fixed-base table (fixed_base.cuh)
                                ("bls12381_g1_commit", "fb", "mul_kernel<kyb::bls12381_FbG1>", 1 << 20)):
    e = entry(prefix, sub, units)
    if e:
        res["kernels"][key] = e
json.dump(res, open(os.path.join(d, "roofline_inputs.json"), "w"), indent=1)
print(json.dumps(res["kernels"], indent=1))
