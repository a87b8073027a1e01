"""Register budgets of the kernels whose occupancy was tuned, read from the code-object
metadata of the built objects -- no GPU needed.  The budget of a kernel is silently lost when one of its out-of-line
callees becomes reachable from another kernel that does not carry it (this happened to the MSM decode kernel when
poly_decode_kernel was added), so the numbers are pinned here."""
import glob
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "kyber_amd", "csrc")
LLVM = "/opt/rocm/lib/llvm/bin"


def _kernel_regs(obj):
    """{demangled-ish kernel name: total VGPR allocation (arch + acc)} of one fat object file."""
    if not os.path.exists(obj):
        pytest.skip(os.path.basename(obj) + " not built (python -c 'import __graft_entry__ as g; g.build()')")
    with tempfile.TemporaryDirectory() as tmp:
        local = os.path.join(tmp, os.path.basename(obj))
        shutil.copy(obj, local)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], cwd=tmp, check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        cos = glob.glob(os.path.join(tmp, "*gfx950*"))
        assert cos, "no gfx950 code object in " + obj
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", cos[0]], check=True, capture_output=True,
                               text=True).stdout
    out = {}
    for blk in notes.split("- .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        out[name] = int(re.search(r"\.vgpr_count:\s+(\d+)", blk).group(1))
    return out


@pytest.mark.skipif(not os.path.exists(os.path.join(LLVM, "llvm-readelf")), reason="no llvm-readelf")
def test_tuned_kernels_keep_their_register_budget():
    ed = _kernel_regs(os.path.join(CSRC, "ed25519.o"))
    msm = _kernel_regs(os.path.join(CSRC, "bls12381_msm.o"))

    def find(table, *parts):
        hits = [v for k, v in table.items() if all(p in k for p in parts)]
        assert len(hits) == 1, (parts, [k for k in table if parts[0] in k])
        return hits[0]

    # three waves per SIMD: <= 170 registers
    assert find(ed, "ed25519_mul_kernelILb1ELb0E") <= 170
    assert find(ed, "ed25519_mul_kernelILb1ELb1E") <= 170  # KYB_F_UNIFORM: the scanned table, same budget
    assert find(ed, "ed25519_mul_base_uniform_kernel") <= 170
    assert find(ed, "13decode_kernel", "EdMsm") <= 170
    # two waves per SIMD: <= 256
    assert find(msm, "13decode_kernel", "_8BlsG1MsmELb0E") <= 256
    # the light decode (vouched-for uncompressed points): four waves per SIMD
    assert find(msm, "13decode_kernel", "_8BlsG1MsmELb1E") <= 128
    assert find(msm, "26tree_fold_bits_coop_kernel", "_8BlsG1MsmE") <= 256
    assert find(msm, "18bucket_coop_kernel", "_8BlsG1MsmE") <= 256
    assert find(msm, "17accumulate_kernel", "_8BlsG1MsmE") <= 256
    # the cooperative tail (four lanes per point): twice the lanes of the one-lane kernels, so two waves per SIMD
    assert find(msm, "18reduce_coop_kernel", "_8BlsG1MsmE") <= 256
    assert find(msm, "21tree_fold_coop_kernel", "_8BlsG1MsmE") <= 256


@pytest.mark.skipif(not os.path.exists(os.path.join(LLVM, "llvm-readelf")), reason="no llvm-readelf")
def test_tower_machine_kernels_fit_three_waves_per_simd():
    """The machine is one 12-wave workgroup per CU (it owns the CU's LDS): three waves per SIMD, i.e. at most 168
    registers per lane.  The BLS12-381 interpreter sits right at that line (167): one register more and the workgroup
    no longer launches with its 12 waves resident."""
    for obj, name in (("bls12381_pair.o", "bls12381_tvm_kernel"), ("bn256_pair.o", "bn256_tvm_kernel"), ("bn254_pair.o", "bn254_tvm_kernel")):
        regs = _kernel_regs(os.path.join(CSRC, obj))
        tvm = {k: v for k, v in regs.items() if name in k}
        assert len(tvm) >= 2, (obj, list(regs))  # Pair, ValidatePairing (bn256: + its product form)
        assert all(v <= 168 for v in tvm.values()), tvm


@pytest.mark.skipif(not os.path.exists(os.path.join(LLVM, "llvm-readelf")), reason="no llvm-readelf")
def test_lane_machine_kernels_fit_two_waves_per_simd_without_a_scratch_working_set():
    """the lane machine's point is two waves per SIMD (65 536 G2 elements ARE two waves per SIMD: a
    wave that does not fit waits for the whole first round) out of 17.9 KB of LDS per wave (eight waves per CU), with no
    working set in scratch -- the per-lane kernels it replaces sat at 512 registers and 2-5 KB of scratch per lane."""
    obj = os.path.join(CSRC, "bls12381.o")
    regs = _kernel_regs(obj)
    lvm = {k: v for k, v in regs.items() if "bls12381_lvm_mul_kernel" in k}
    assert len(lvm) == 4, list(regs)                    # G1 / pair, product / micro-benchmark instantiations
    assert all(v <= 256 for v in lvm.values()), lvm
    with tempfile.TemporaryDirectory() as tmp:
        local = os.path.join(tmp, "bls12381.o")
        shutil.copy(obj, local)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], cwd=tmp, check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", glob.glob(os.path.join(tmp, "*gfx950*"))[0]],
                               check=True, capture_output=True, text=True).stdout
    seen = 0
    for blk in notes.split("- .agpr_count:")[1:]:
        if "bls12381_lvm_mul_kernel" not in blk:
            continue
        seen += 1
        lds = int(re.search(r"\.group_segment_fixed_size:\s+(\d+)", blk).group(1))
        scratch = int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", blk).group(1))
        assert lds * 8 <= 160 * 1024, lds               # eight waves per CU
        assert scratch <= 160, scratch                  # the out-of-line inversion's frame and a handful of spilled words
    assert seen == 4


@pytest.mark.skipif(not os.path.exists(os.path.join(LLVM, "llvm-readelf")), reason="no llvm-readelf")
def test_units_on_a_two_wave_budget_keep_it():
    """an out-of-line device function is compiled for the loosest register budget of the
    kernels that reach it, so ONE kernel without a budget in a translation unit lets every kernel that shares code with
    it grow past its own launch bounds (bn256_g2_mul_kernel: 374 registers under a three-wave bound).  The units that
    are meant to run two waves per SIMD -- every kernel of the BN scalar-multiplication units, the large-batch copies of
    the BLS12-381 unmarshal / hash kernels, the BLS12-381 fixed-base walk, Ed25519's element-wise kernels -- are pinned
    at 256 registers here; a new kernel with plain launch bounds in one of them fails this test, not a benchmark."""
    for obj in ("bn256.o", "bn254.o"):
        regs = _kernel_regs(os.path.join(CSRC, obj))
        assert len(regs) >= 15 and all(v <= 256 for v in regs.values()), {k: v for k, v in regs.items() if v > 256}
    w2 = _kernel_regs(os.path.join(CSRC, "bls12381_unm2.o"))
    assert len(w2) == 4 and all(v <= 256 for v in w2.values()), w2
    split = _kernel_regs(os.path.join(CSRC, "bls12381_g1split.o"))
    assert all(v <= 256 for v in split.values()), split
    fb = _kernel_regs(os.path.join(CSRC, "bls12381_fb.o"))
    walk = {k: v for k, v in fb.items() if "10mul_kernel" in k}
    assert len(walk) == 2 and all(v <= 256 for v in walk.values()), walk
    ed = _kernel_regs(os.path.join(CSRC, "ed25519.o"))
    for name in ("ed25519_add_kernel", "ed25519_unmarshal_kernel", "ed25519_hash_kernel", "ed25519_encode_kernel"):
        hits = [v for k, v in ed.items() if name in k]
        assert len(hits) == 1 and hits[0] <= 256, (name, hits)
    # ... and the unit that is meant to keep its registers (cooperating-lane kernels, ladders of a half-empty chip) has
    # no fixed-base kernels left in it
    bls = _kernel_regs(os.path.join(CSRC, "bls12381.o"))
    assert not [k for k in bls if "2fb" in k], [k for k in bls if "2fb" in k]
