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

d, tag = sys.argv[1], sys.argv[2]
tag_of = dict(a.split("=", 1) for a in sys.argv[3:])


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
    for k, v in cs.items():
        if sub in k:
            return v
    return {}


def entry(prefix, sub, units):
    tag = tag_of.get(prefix, sys.argv[2])
    sq = kernel(counters(os.path.join(d, f"{tag}_{prefix}_sq.txt")), sub)
    fe = kernel(counters(os.path.join(d, f"{tag}_{prefix}_fetch.txt")), sub)
    wr = kernel(counters(os.path.join(d, f"{tag}_{prefix}_write.txt")), sub)
    if not sq:
        return None
    e = {"units_per_launch": units,
         "valu_busy": 4 * sq["SQ_ACTIVE_INST_VALU"] / (1024 * sq["GRBM_GUI_ACTIVE"] / 8),
         "valu_insts_per_unit": sq["SQ_INSTS_VALU"] * 64 / units,
         "wait_share_of_wave_cycles": sq["SQ_WAIT_ANY"] / sq["SQ_WAVE_CYCLES"],
         "source": f"profiles/{tag}_{prefix}_{{sq,fetch,write}}.txt ({sub})"}
    if fe and wr:
        e["hbm_bytes_per_launch"] = (fe["FETCH_SIZE"] + wr["WRITE_SIZE"]) * 1024
        e["fetch_bytes"], e["write_bytes"] = fe["FETCH_SIZE"] * 1024, wr["WRITE_SIZE"] * 1024
    return e


old = json.load(open(os.path.join(d, "roofline_inputs.json")))
res = {"_comment": "Inputs bench.py reads for its roofline objects; every number comes from a file in this directory "
                   "(tools/roofline_inputs.py).",
       "imad_peak_lane_ops_per_s": old["imad_peak_lane_ops_per_s"], "imad_peak_source": old["imad_peak_source"],
       "kernels": {}}
for key, prefix, sub, units in (("ed25519_mul", "ed", "ed25519_mul_kernel<true>", 1 << 20),
                                ("bls12381_pair", "bls12381", "bls12381_tvm_kernel<0>", 1 << 16),
                                ("bls12381_check", "bls12381", "bls12381_tvm_kernel<1>", 1 << 16),
                                ("bls12381_verify", "verify", "bls12381_tvm_kernel<2>", 1 << 16),
                                ("bn256_pair", "bn256", "bn256_tvm_kernel<0>", 1 << 18),
                                ("bn256_check", "bn256", "bn256_tvm_kernel<1>", 1 << 18),
                                ("bn254_pair", "bn254", "bn254_tvm_kernel<0>", 1 << 18),
                                ("bn254_check", "bn254", "bn254_tvm_kernel<1>", 1 << 18),
                                # round 3: the lane machine's ladders (a G2 element is two lanes), the per-lane kernels of the
                                # same probe with KYB_LVM_MIN huge, the MSM's accumulate stage
                                ("bls12381_g1_mul", "mul", "bls12381_lvm_mul_kernel<false", 1 << 16),
                                ("bls12381_g2_mul", "mul", "bls12381_lvm_mul_kernel<true", 1 << 16),
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
