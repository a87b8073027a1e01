#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path on MI355X.

Headline workload (BASELINE.json configs[1]): Ed25519 batched fixed-base + variable-base scalar
multiplication, 2^20 scalars per GPU.  One "step" = one pass of the hot path over one batch:
2^20 x Point.Mul(s, nil) (fixed-base) followed by 2^20 x Point.Mul(s, P) (variable-base; compressed
points in, compressed points out), inputs already resident in HBM.  value = scalar-muls/s over the
whole job.

N > 1 (launched by torch.distributed.run, one rank per GPU): the batch shards trivially -- every
rank owns its own 2^20-element batch (weak scaling); independent scalar multiplications have no
data-path collective, only the timing barrier and a MAX all-reduce of the elapsed time.

BASELINE.json's metric is composite ("scalar-muls/s + pairings/s per node; MSM sec at 2^20"), so the
run also measures, outside the headline's timed region, the BLS12-381 pairing rates at 2^16 pairs per
GPU (configs[3]), the node-wide BLS12-381 G1 MSM time at 2^20 points sharded over the ranks with
the RCCL all-gather of partial points (configs[2]) and the bn256 pairing rate (configs[4]).

Output (rank 0): the full record -- every per-kernel roofline object, the CPU legs' sample descriptions -- goes to
bench_detail.json on disk (and gpurun_out/bench_detail.json when that directory exists); stdout gets ONE compact
JSON line (compact_line(): < 4 KB, asserted by tests/test_bench_line.py) carrying the headline, ONE roofline
object, a scalar-only cpu_baseline and the composite metric's other figures as scalars with a fraction beside each
(one small object per run, like benchmark/benchmark.go:166-200).
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_PER_GPU = 1 << 20
# algorithmic bytes per unit: var-base 32+32 in, 32 out; fixed-base 32 in, 32 out
BYTES_VAR, BYTES_FIX = 96, 64
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
IMAD_4CYCLE_PEAK = 256 * 4 * 64 / 4 * 2.4e9  # a wave's 64 multiply-adds issue over 4 cycles on each of 1 024 SIMDs at 2.4 GHz: 3.93e13 lane-MAD/s
REF_SINGLE_CORE_PER_S = 2.0 / (349399e-9 + 60658e-9)
# 64-bit integer multiply-adds (v_mad_i64_i32) per scalar multiplication as built:
# a field multiplication is 100 MADs, a squaring 55.  With an in-kernel ToBytes, variable-base = 1366 M + 1517 S
# and fixed-base = 475 M + 270 S; batches >= 4096 defer the encoding to ed25519_encode_kernel, which replaces the
# per-element inversion (254 S + 11 M) by 3 M + 1/16 of an inversion (Montgomery's trick over 16 elements).
_INV_SAVED = (254 * 55 + 11 * 100) - (3 * 100 + (254 * 55 + 11 * 100) / 16)
IMADS_VAR = int(1366 * 100 + 1517 * 55 - _INV_SAVED)
# fixed-base: 32 mixed additions from the radix-256 table (3 M + 4 M for the completed -> extended conversion each)
# + the deferred encoding's share (3 M + 2 M + 1/16 inversion)
IMADS_FIX = int(32 * 7 * 100 + 5 * 100 + (254 * 55 + 11 * 100) / 16)


def shake(label: bytes, nbytes: int) -> np.ndarray:
    return np.frombuffer(hashlib.shake_256(label).digest(nbytes), dtype=np.uint8)


def make_inputs(n: int, rank: int):
    """Deterministic synthetic inputs: canonical scalars (< 2^252 <= l) from labelled SHAKE-256 streams;
    points P_i = h_i * B produced by the verified fixed-base path (prime-order subgroup, like
    util/key/key.go:41-49)."""
    s = shake(b"kyberhip/v1/ed25519/scalars/%d" % rank, n * 32).reshape(n, 32).copy()
    h = shake(b"kyberhip/v1/ed25519/point-seeds/%d" % rank, n * 32).reshape(n, 32).copy()
    s[:, 31] &= 0x0F
    h[:, 31] &= 0x0F
    return s, h


def be_scalars(label: bytes, n: int) -> np.ndarray:
    """n big-endian 32-byte scalars < 2^254 (valid for both pairing curves' plain-integer semantics)."""
    a = shake(label, n * 32).reshape(n, 32).copy()
    a[:, 0] &= 0x3F
    return a


def cpu_info():
    """cores this process may run on (the lease, not the machine: os.cpu_count() names every logical CPU of the host) and
    the CPU model"""
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    quota = None  # a cgroup CPU quota caps the lease below its affinity mask (cpu.max: "<quota> <period>" or "max")
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        pass
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return cores, os.cpu_count() or cores, model, quota


def usable_cores():
    """the cores the CPU legs may actually keep busy: the affinity mask capped by the cgroup CPU quota -- the figure
    `cpu_baseline.cores` reports and the number of oracle threads started (256 logical CPUs were reported for a
    16-core lease)"""
    cores, _, _, quota = cpu_info()
    return max(1, min(cores, int(quota + 0.5))) if quota else cores


def _sha(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def cpu_baseline(scalars: np.ndarray, points: np.ndarray, gpu_fix: np.ndarray, gpu_var: np.ndarray):
    """The oracle's C restatement (kind "port") on the host cores, on a bounded prefix of THE SAME batch the GPU
    just processed (the whole 2^20 when the host manages it in ~10 s), and the comparison SURVEY.md section 8d asks for:
    SHA-256 over the concatenated outputs, oracle against GPU."""
    from tests import _oracle_c as OC

    aff, logical, model, quota = cpu_info()
    cores = usable_cores()

    def rate(n, threads):
        t0 = time.perf_counter()
        OC.ed_mul(scalars[:n], points[:n], threads=threads)
        OC.ed_mul_base(scalars[:n], threads=threads)
        return 2 * n / (time.perf_counter() - t0)

    r1 = rate(2048, 1)
    rall = rate(min(len(scalars), 2048 * min(cores, 64)), cores) if cores > 1 else r1
    threads = cores if rall > r1 else 1
    per_s = max(r1, rall)
    n = int(min(len(scalars), max(4096, per_s * 6.0)))  # ~12 s of CPU work (two kernels x 6 s)
    t0 = time.perf_counter()
    var, st = OC.ed_mul(scalars[:n], points[:n], threads=threads)
    fix = OC.ed_mul_base(scalars[:n], threads=threads)
    dt = time.perf_counter() - t0
    h_cpu, h_gpu = _sha(fix, var), _sha(gpu_fix[:n], gpu_var[:n])
    return {"value": 2 * n / dt, "unit": "scalar-muls/s", "cores": threads, "kind": "port",
            "single_thread_value": r1, "affinity_cores": aff, "cgroup_cpu_quota_cores": quota,
            "logical_cpus_of_the_host": logical, "cpu_model": model,
            "outputs_match": bool(h_cpu == h_gpu and not st.any()), "outputs_compared": 2 * n, "outputs_sha256": h_cpu,
            "sample": f"{n} fixed-base + {n} variable-base Ed25519 scalar-muls = the first {n} elements of the GPU's batch, "
                      f"oracle/ed25519_ref.c (radix-2^51 C restatement of ge.go:373/443, gcc -O3), "
                      f"{threads} thread(s) on the {cores} core(s) this process may keep busy (affinity {aff}, cgroup quota {quota}, {logical} logical CPUs on the host, {model}); "
                      f"the Go reference itself cannot run here (no Go toolchain)"}


def cpu_baseline_pairing_and_msm(bn, bls):
    """CPU figures for the composite metric's other two thirds, each over the WHOLE config-size batch of the GPU's own
    inputs with every output compared (tests/_full_digest.py): Suite.Pair + G1 / G2 Mul on BLS12-381 (2^16) and bn256
    (2^18), the N x (Mul + Add) sum the reference runs where this engine runs an MSM (2^20 points), the 2^20 same-base
    commits."""
    import torch

    aff, logical, model, quota = cpu_info()
    cores = usable_cores()
    out = {"cores": cores, "affinity_cores": aff, "cgroup_cpu_quota_cores": quota, "cpu_model": model, "kind": "port"}
    # ---- whole-batch digests at the config sizes (SURVEY.md section 8d "Correctness at scale"): EVERY output of
    # configs[2] / [3] / [4] against the C oracle -- Suite.Pair, G1 Mul, G2 Mul over all 2^16 BLS12-381 and all 2^18 bn256
    # elements (oracle/bls12381_pair_ref.c: a port of the published algorithms, the backends being external modules;
    # oracle/bn256_ref.c: optate.go / curve.go / twist.go restated), the 2^20-point MSM against N x (Mul + Add) -- and the
    # oracle's time over the whole batch as that config's CPU figure.  Guarded: nothing here may cost the run its line.
    from tests import _full_digest as FD

    full = {}
    for key, fn in (("bls12381", lambda: FD.pairing_suite("bls12381", 1 << 16, threads=cores)),
                    ("bn256", lambda: FD.pairing_suite("bn256", 1 << 18, threads=cores)),
                    ("bls12381_g1_msm_2p20", lambda: FD.bls12381_g1_msm(1 << 20, threads=cores))):
        try:
            full[key] = fn()
        except Exception as e:  # noqa: BLE001 -- a reported baseline must never take the benchmark line down
            full[key] = {"error": repr(e)[:300]}
    out["full_batch_digests"] = full
    for suite in ("bls12381", "bn256"):
        rec = _g(full, suite, "pair") or {}
        out[suite + "_pairings"] = {"value": rec.get("cpu_per_s"), "unit": "pairings/s", "cores": cores,
                                    "outputs_match": rec.get("outputs_match"), "outputs_compared": rec.get("outputs_compared"),
                                    "sample": "the whole config-size batch: %s Suite.Pair calls, %s, GT bytes compared with the GPU's" % (
                                        rec.get("outputs_compared"), _g(full, suite, "oracle"))}
    mrec = full.get("bls12381_g1_msm_2p20") or {}
    out["bls12381_g1_mul_add"] = {"value": mrec.get("cpu_points_per_s"), "unit": "points/s", "cores": cores,
                                  "seconds_for_2p20_points": mrec.get("cpu_seconds"), "outputs_match": mrec.get("outputs_match"),
                                  "outputs_compared": mrec.get("outputs_compared"), "sample": mrec.get("what")}
    # ---- share.PriPoly.Commit through the fixed-base table (share/poly.go:143-149), an arbitrary base, ALL 2^20
    # coefficients of the measured configuration against the oracle's element-wise G1 Mul
    try:
        from tests import _oracle_c as OC

        n = 1 << 20
        base = np.asarray(bls.g1_commit((0x1234567).to_bytes(32, "big"))[0])[0]
        ks = be_scalars(b"kyberhip/v1/msm/k", n)
        outc, stc = bls.g1_commit(torch.from_numpy(ks).cuda(), torch.from_numpy(base.copy()).cuda())
        t0 = time.perf_counter()
        out_o, st_o = OC.bls12381_g1_mul(ks, np.tile(base, (n, 1)), threads=cores)
        dt = time.perf_counter() - t0
        okc = not bool(stc.any().item()) and not st_o.any() and _sha(outc.cpu().numpy()) == _sha(out_o)
        out["bls12381_g1_commit_oracle_sample"] = {"outputs_match": bool(okc), "outputs_compared": n, "batch": n, "cpu_seconds": dt,
                                                   "sample": "fixed-base table walk over an arbitrary base, all 2^20 coefficients against "
                                                             "oracle/bls12381_pair_ref.c ora_bls12381_g1_mul (SHA-256 over the outputs)"}
    except Exception as e:  # noqa: BLE001
        out["bls12381_g1_commit_oracle_sample"] = {"error": repr(e)[:300]}
    return out


REF_BLS_VERIFY_PER_S_SINGLE_CORE = 303  # BASELINE.md section 1 (sign/bls Verify, BLS12-381 circl backend, signatures on G1, one core)


def max_over_ranks(dist, values, device="cuda"):
    """the slowest rank's figure for each of `values` (one all-reduce MAX; the identity without a process group):
    every per-job rate of this file is units of ALL ranks / this time"""
    import torch

    if not dist:
        return [float(v) for v in values]
    t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return [float(x) for x in t]


def timed(fn, reps=20, warm=5):
    """median of `reps` HIP-event timings (ms) after `warm` untimed calls"""
    import torch

    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2]


def _prof():
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "roofline_inputs.json")))
    except Exception:
        return {}


def _tvm_mads():
    """integer MADs per pairing / pairing check of the tower-machine programs, counted from the generated programs
    themselves (kyber_amd/csrc/gen_tower_vm.py is product tooling, not the oracle)"""
    sys.path.insert(0, os.path.join(ROOT, "kyber_amd", "csrc"))
    import gen_tower_vm as G

    return {"bls12381": (G.build_bls12381_pair().mads(), G.build_bls12381_check().mads(), G.build_bls12381_verify().mads()),
            # ValidatePairing runs the product-form program by default since round 4 (bn_pair.inc); the two-pairing
            # program only decides lanes whose joint Miller value is zero
            "bn256": (G.build_bn256_pair().mads(), G.build_bn256_check_product().mads()),
            "bn254": (G.build_bn254_pair().mads(), G.build_bn254_check().mads()),
            "verifyk": G.build_bls12381_verify_same_key().mads(),
            "gtmul": {"bls12381": G.build_bls12381_gtmul().mads(), "bn256": G.build_bn256_gtmul().mads(),
                      "bn254": G.build_bn254_gtmul().mads()}}


def _lvm_mads():
    """integer MADs per scalar multiplication of the lane-machine programs (kyber_amd/csrc/gen_lane_vm.py, product
    tooling): a G1 element is one lane, a G2 element two"""
    sys.path.insert(0, os.path.join(ROOT, "kyber_amd", "csrc"))
    import gen_lane_vm as G

    g1, g2 = G.build_bls12381_g1_mul(), G.build_bls12381_g2_mul()
    return {"g1": g1.mads(), "g2": 2 * g2.mads(), "g2_karatsuba": 2 * g2.mads(karatsuba=True)}


# UnmarshalBinary on BLS12-381 as the per-lane code does it (bls12381.cuh; a field multiplication is 13 x 13 products +
# 13 x 13 reduction multiply-adds = 338, a squaring 91 + 169 = 260): G1 = a 379-bit power for the square root (4-bit
# windows: 379 S + 109 M) + the endomorphism test phi(P) = [-z^2]P (2 x (63 doublings of 2M + 5S + 5 additions of 11M + 5S));
# G2 = two such powers for the Fp2 square root + psi(Q) = [z]Q (63 doublings + 5 additions over Fp2, Karatsuba: M2 = 3M, S2 = 2M)
_M, _S = 338, 260
# (round 5: both 63-bit multiplications of the r-torsion test on lazy limbs -- jac_lazy.cuh jaclz_dbl_t 3M + 4S,
# jaclz_madd_t 6M + 3S + one two-product multiplication of 3 x 169 -- the numerator FELL slightly with the change)
_DBL_T, _MADD_T = 3 * _M + 4 * _S, 6 * _M + 3 * _S + 3 * 169
MADS_G1_UNMARSHAL = (379 * _S + 109 * _M) + 126 * _DBL_T + 10 * _MADD_T
# G1Elt.Mul as the per-lane kernel does it since round 5 (the band the bench's 2^16 elements fall in): the GLV walk of 136
# doublings + 68 mixed additions on lazy limbs, the packed table of eight multiples (1 doubling + 6 additions + the shared
# inversion's ~8 M), 8 beta x products, 3 M to leave the form -- 540 566, against 606 046 for the lane machine's program
MADS_BLS_G1_LADDER = 136 * _DBL_T + 68 * _MADD_T + (2 * _M + 5 * _S) + 6 * (11 * _M + 5 * _S) + 8 * _M + 8 * _M + 3 * _M
MADS_G1_DECOMPRESS = 379 * _S + 109 * _M  # the square root alone (validated compressed points: no subgroup test)
MADS_G2_UNMARSHAL = 2 * (379 * _S + 109 * _M) + 12 * _M + (63 * (2 * 3 + 5 * 2) + 5 * (11 * 3 + 5 * 2)) * _M
# pointG1 / pointG2.Mul of the BN suites as the per-lane code does it (bn_suite.inc; nine 29-bit limbs: a field
# multiplication is 81 + 81 = 162 multiply-adds, a squaring 45 + 81 = 126; Fp2: M2 = 3M, S2 = 2M).  G1: GLV, 34 windows of
# (4 doublings of 2M + 5S + 2 mixed additions of 7M + 4S), the table (1 doubling + 6 additions of 11M + 5S + ~53 M for the
# shared inversion).  G2: the membership relation (63 doublings + 21 mixed additions for [u]Q, 3 more additions, a doubling)
# and the GLS walk (17 windows of (4 + 4), the same table) over Fp2.
_MB, _SB = 162, 126
# (round 5, lazy limbs: the mixed addition's Y3 is ONE two-product multiplication on G1 -- 5M + 3 x 81 + 4S; over Fp2 the two
# products stay two multiplications of two two-product halves each: the same 4 698 as 7 M2 + 4 S2)
_BN_DBL, _BN_MADD, _BN_ADD = 2 * _MB + 5 * _SB, 5 * _MB + 3 * 81 + 4 * _SB, 11 * _MB + 5 * _SB
_BN_DBL2, _BN_MADD2, _BN_ADD2 = (2 * 3 + 5 * 2) * _MB, (7 * 3 + 4 * 2) * _MB, (11 * 3 + 5 * 2) * _MB
MADS_BN_G1_MUL = 136 * _BN_DBL + 68 * _BN_MADD + _BN_DBL + 6 * _BN_ADD + 53 * _MB
MADS_BN_G2_GLS = 64 * _BN_DBL2 + 68 * _BN_MADD2 + _BN_DBL2 + 6 * _BN_ADD2 + 53 * 3 * _MB
MADS_BN_G2_MEMBER = 63 * _BN_DBL2 + 22 * _BN_MADD2 + 2 * _BN_ADD2 + _BN_DBL2 + 6 * 3 * _MB
# Pippenger on BLS12-381 G1 at 2^20 points (msm.cuh): 2n half-scalars x 8 windows of 16 bits, one mixed addition per
# (point, window) -- XYZZ form since round 3, 8M + 2S (3 224 multiply-adds; madd-2007-bl, 7M + 4S = 3 406, until then:
# the numerator FELL with the change) -- + the running-sum reduction of 8 x 2^15 buckets (2 full additions each, 11M + 5S)
# Round 5: the addition runs in limb form with Y3 = R (Q - X3) - Y1 PPP as ONE two-product multiplication (3 x 169
# multiply-adds instead of 2 x 338): 6M + 2S + 507 = 3 055 -- the numerator fell again with the change.
_XYZZ_MADD = 6 * _M + 2 * _S + 3 * 169
MADS_MSM_BLS_G1_2P20_PER_POINT = 2 * 8 * _XYZZ_MADD + (8 * (1 << 15) * 2 * (11 * _M + 5 * _S)) / (1 << 20)
# share.PriPoly.Commit through the fixed-base table (fixed_base.cuh): 26 XYZZ additions (limb form, as above), leaving the form
# (2M), to affine (1S + 3M; the division-step inversion is ~25 batches of ~130 multiply-adds)
MADS_G1_COMMIT = 26 * _XYZZ_MADD + 5 * _M + _S + 25 * 130
# the best known count for the BLS12-381 pairing on this limb arithmetic: the Karatsuba tower of round 1 (5.4e6 per
# Pair) against the machine's schoolbook-with-lazy-reduction program; checks / verifies scaled alike
BLS_PAIR_BEST_KNOWN = 5.4e6
# pairing/bn256's own formulas cost 31 595 gfpMul per pairing (oracle/bn256_ref.c built with -DORA_COUNT): at one
# reduction per multiplication that is 31 595 x 200 multiply-adds on 10 limbs -- MORE than the machine's program,
# which reduces once per output coefficient
BN256_PAIR_REFERENCE_FORMULA = 31595 * 200


def _roof(units_per_s, mads_per_unit, alg_bytes_per_unit, prof, key, best_known=None):
    """roofline object of a side workload: integer-MAD issue is the binding resource, HBM figures alongside.
    best_known: multiply-adds of the cheapest formula known for the unit -- the fraction against it cannot be raised
    by doing more work"""
    peak = prof.get("imad_peak_lane_ops_per_s")
    k = prof.get("kernels", {}).get(key, {})
    r = {"bound": "valu-imad", "mads_per_unit": mads_per_unit, "achieved": units_per_s * mads_per_unit, "peak": peak,
         "unit": "lane-MAD/s", "frac": (units_per_s * mads_per_unit / peak) if peak else None,
         "hbm": {"algorithmic_bytes_per_unit": alg_bytes_per_unit, "achieved_GBps": units_per_s * alg_bytes_per_unit / 1e9,
                 "peak_GBps": HBM_PEAK_GBS, "frac": units_per_s * alg_bytes_per_unit / 1e9 / HBM_PEAK_GBS},
         "traffic": k.get("hbm_bytes_per_launch"), "traffic_units_per_launch": k.get("units_per_launch"),
         # counter bytes over algorithmic bytes for the profiled launch: well above 1 = re-reads / spills / per-lane tables
         "traffic_ratio": (k["hbm_bytes_per_launch"] / (alg_bytes_per_unit * k["units_per_launch"]))
         if k.get("hbm_bytes_per_launch") and k.get("units_per_launch") and alg_bytes_per_unit else None,
         "valu_busy_profiled": k.get("valu_busy"), "profile": k.get("source")}
    if best_known is not None:
        m = min(best_known, mads_per_unit)
        r["mads_best_known_formula"] = m
        r["frac_of_best_known_formula"] = (units_per_s * m / peak) if peak else None
    return r


def other_workloads(rank, world, dist):
    """BLS12-381 / bn256 pairing rates and the node-wide G1 MSM (outside the headline timing)."""
    import torch

    from kyber_amd import dist as kd
    from kyber_amd.pairing import bls12381 as bls, bn254 as bn4, bn256 as bn

    out = {}
    prof = _prof()
    mads = _tvm_mads()
    # configs[3] / configs[4] sizes; bn254 (SURVEY section 8 f4's last item, no config of its own) at bn256's
    for name, m, npair in (("bls12381", bls, 1 << 16), ("bn256", bn, 1 << 18), ("bn254", bn4, 1 << 18)):
        k = torch.from_numpy(be_scalars(b"kyberhip/v1/%s/k/%d" % (name.encode(), rank), npair)).cuda()
        h = torch.from_numpy(be_scalars(b"kyberhip/v1/%s/h/%d" % (name.encode(), rank), npair)).cuda()
        g1b = torch.from_numpy(np.frombuffer(m.G1_BASE, dtype=np.uint8).copy()).cuda()
        g2b = torch.from_numpy(np.frombuffer(m.G2_BASE, dtype=np.uint8).copy()).cuda()
        P, st1 = m._mul(1, h, g1b, True)
        Q, st2 = m._mul(2, k, g2b, True)
        ms_pair = timed(lambda: m.batch_pair(P, Q))
        # valid BLS-verify shaped quadruples: e(H, X) == e(sig, G2) with sig = x H  (sign/bls/bls.go:36-38)
        sig, _ = m.g1_batch_mul(k, P)
        G2 = g2b.repeat(npair, 1)
        ok, st3 = m.batch_validate_pairing(P, Q, sig, G2)
        ms_chk = timed(lambda: m.batch_validate_pairing(P, Q, sig, G2))
        ms_g1 = timed(lambda: m.g1_batch_mul(k, P))
        ms_g2 = timed(lambda: m.g2_batch_mul(k, Q))
        # the same check as sign/bls Verify meets it: H(m) is the library's own output, the key X was unmarshalled
        # (validated) once when it was registered and G2.Base() is a constant -- only the signature is new input
        trust = m.F_TRUSTED(0) | m.F_TRUSTED(1) | m.F_TRUSTED(3)
        ok_t, st_t = m.batch_validate_pairing(P, Q, sig, G2, trust)
        ms_chk_t = timed(lambda: m.batch_validate_pairing(P, Q, sig, G2, trust))
        ms_pair_t = timed(lambda: m.batch_pair(P, Q, m.F_TRUSTED(0) | m.F_TRUSTED(1)))
        # GT exponentiation (pointGT.Mul / GTElt.Mul) of the pairing values just computed: e(P, Q)^k == e(kP, Q)
        gt, _ = m.batch_pair(P, Q, m.F_TRUSTED(0) | m.F_TRUSTED(1))
        gk, st_gt = m.gt_batch_mul(k, gt)
        ms_gt = timed(lambda: m.gt_batch_mul(k, gt))
        ns = 256
        kP, _ = m.g1_batch_mul(k[:ns].contiguous(), P[:ns].contiguous())
        e_kp, _ = m.batch_pair(kP, Q[:ns].contiguous())
        gt_ok = bool((gk[:ns] == e_kp).all().item()) and not bool(st_gt.any().item())
        del gt, gk
        ms_pair, ms_chk, ms_g1, ms_g2, ms_chk_t, ms_pair_t, ms_gt = max_over_ranks(
            dist, [ms_pair, ms_chk, ms_g1, ms_g2, ms_chk_t, ms_pair_t, ms_gt])
        good = bool(ok.all().item()) and bool(ok_t.all().item()) and not (
            st1.any().item() or st2.any().item() or st3.any().item() or st_t.any().item())
        out[name] = {"pairs_per_gpu": npair, "pairings_per_s": world * npair / ms_pair * 1e3,
                     "pairing_checks_per_s": world * npair / ms_chk * 1e3,
                     "pairings_per_s_validated_inputs": world * npair / ms_pair_t * 1e3,
                     "pairing_checks_per_s_only_sig_unvalidated": world * npair / ms_chk_t * 1e3,
                     "g1_muls_per_s": world * npair / ms_g1 * 1e3, "g2_muls_per_s": world * npair / ms_g2 * 1e3,
                     "gt_muls_per_s": world * npair / ms_gt * 1e3, "gt_mul_matches_pairing_of_multiple": gt_ok,
                     "all_checks_true": good and gt_ok, "timing": "median of 20 launches after 5 warm-ups, HIP events"}
        g1b_, g2b_ = m.G1_LEN, m.G2_LEN
        # roofline of the pairing entry points: the MADs of the tower-machine program (operand unmarshalling and
        # its subgroup checks are extra work inside the measured time, so the fraction is a lower bound for the
        # machine itself; the validated-input figure is the machine alone plus a 5 % operand kernel)
        bk = {"bls12381": BLS_PAIR_BEST_KNOWN, "bn256": min(BN256_PAIR_REFERENCE_FORMULA, mads[name][0])}.get(name)
        bk_chk = None if bk is None else bk * mads[name][1] / mads[name][0]
        out[name]["roofline"] = {
            "pair": _roof(npair / ms_pair * 1e3, mads[name][0], g1b_ + g2b_ + m.GT_LEN, prof, name + "_pair", bk),
            "pair_validated_inputs": _roof(npair / ms_pair_t * 1e3, mads[name][0], g1b_ + g2b_ + m.GT_LEN, prof, name + "_pair", bk),
            "pair_check": _roof(npair / ms_chk * 1e3, mads[name][1], 2 * (g1b_ + g2b_) + 1, prof, name + "_check", bk_chk),
            "gt_mul": _roof(npair / ms_gt * 1e3, mads["gtmul"][name], 32 + 2 * m.GT_LEN, prof, name + "_gt_mul")}
        if name == "bls12381" and rank == 0:
            # Which kind of box is this?  The per-lane kernels stream 100-300 KB loop bodies through the 64 KB
            # instruction cache; the lane machine is a 27 KB interpreter.  On the boxes of rounds 1-3 the ratio below is
            # ~0.40; on one box (profiles/r03_codesize_ab.json) it was ~0.85 and every per-lane figure of this line 2-3x
            # lower, with the interpreter kernels and the Ed25519 headline unchanged.
            ms_um = timed(lambda: m.ENGINE.batch_unmarshal(2, Q))
            ms_lm = timed(lambda: m.g2_batch_mul(k, Q, m.F_TRUSTED(0)))
            out[name]["code_fetch_probe"] = {
                "g2_unmarshal_ms": ms_um, "g2_lane_machine_mul_validated_ms": ms_lm, "ratio": ms_um / ms_lm,
                "expected_ratio": 0.40, "slow_instruction_fetch_box": bool(ms_um / ms_lm > 0.6),
                "see": "profiles/r03_codesize_ab.json, DESIGN.md section 5 item 24"}
        if name == "bls12381":
            # G1Elt.Mul / G2Elt.Mul with everything UnmarshalBinary checks (flags = 0): the per-lane unmarshal kernel +
            # the lane machine's ladder; G2 also against the count with Karatsuba Fp2 products (3 instead of 4)
            lm = _lvm_mads()
            out[name]["roofline"]["g1_mul"] = _roof(npair / ms_g1 * 1e3, min(lm["g1"], MADS_BLS_G1_LADDER) + MADS_G1_UNMARSHAL, 32 + 2 * g1b_, prof, "bls12381_g1_mul")
            out[name]["roofline"]["g2_mul"] = _roof(npair / ms_g2 * 1e3, lm["g2"] + MADS_G2_UNMARSHAL, 32 + 2 * g2b_, prof, "bls12381_g2_mul",
                                                    lm["g2_karatsuba"] + MADS_G2_UNMARSHAL)
        if name in ("bn256", "bn254"):
            # pointG1 / pointG2.Mul with every operand re-validated (flags = 0): the per-lane kernels of bn_suite.inc
            out[name]["roofline"]["g1_mul"] = _roof(npair / ms_g1 * 1e3, MADS_BN_G1_MUL, 32 + 2 * g1b_, prof, name + "_g1_mul")
            out[name]["roofline"]["g2_mul"] = _roof(npair / ms_g2 * 1e3, MADS_BN_G2_GLS + MADS_BN_G2_MEMBER, 32 + 2 * g2b_, prof, name + "_g2_mul")
        if True:
            # the whole sign/bls Verify pipeline on the device: Hash(msg) (bn256: SHA-256 + try-and-increment; bn254:
            # Keccak-256 expand + Shallue-van de Woestijne; BLS12-381: RFC 9380 hash_to_curve) then the pairing check (sign/bls/bls.go:82-96), 32-byte messages
            msgs = torch.from_numpy(shake(b"kyberhip/v1/%s/msgs/%d" % (name.encode(), rank), npair * 32).reshape(npair, 32).copy()).cuda()

            def verify():
                if name == "bls12381":  # fused kernel: hash + unmarshal checks + 2 Miller loops + final exp
                    return m.batch_verify_g1(Q, msgs, sig)
                Hm, _ = m.batch_hash_g1(msgs)
                return m.batch_validate_pairing(Hm, Q, sig, G2)

            def verify_known_keys():  # keys validated when registered (KYB_F_TRUSTED on the key argument)
                if name == "bls12381":
                    return m.batch_verify_g1(Q, msgs, sig, flags=m.F_TRUSTED(0))
                Hm, _ = m.batch_hash_g1(msgs)
                return m.batch_validate_pairing(Hm, Q, sig, G2, trust)

            ms_v = timed(verify)
            ms_vk = timed(verify_known_keys)
            ms_v, ms_vk = max_over_ranks(dist, [ms_v, ms_vk])
            out[name]["bls_verify_pipeline_per_s"] = world * npair / ms_v * 1e3
            out[name]["bls_verify_pipeline_per_s_known_keys"] = world * npair / ms_vk * 1e3
            if name == "bls12381":  # the VERIFY program (generator lines from a table); hashing and unmarshalling are extra
                out[name]["roofline"]["verify"] = _roof(npair / ms_v * 1e3, mads[name][2], g1b_ + g2b_ + 32 + 1, prof, name + "_verify")
                # ONE signer for the whole batch (a drand chain; sign/bls/bls.go:82-96 in a loop with the same X): program
                # VERIFYK, both Miller loops from line tables -- valid signatures of the first key over the same messages
                Hm, _ = m.batch_hash_g1(msgs)
                x1 = k[:1].contiguous()
                X1 = m.g2_commit(x1)[0][0].contiguous()
                sig1, _ = m.g1_batch_mul(x1.repeat(npair, 1), Hm)
                ok1, st1 = m.batch_verify_g1_same_key(X1, msgs, sig1)
                ms_v1 = timed(lambda: m.batch_verify_g1_same_key(X1, msgs, sig1))
                ms_v1, = max_over_ranks(dist, [ms_v1])
                out[name]["bls_verify_same_key_per_s"] = world * npair / ms_v1 * 1e3
                out[name]["bls_verify_same_key_all_true"] = bool(ok1.all().item()) and not bool(st1.any().item())
                out[name]["roofline"]["verify_same_key"] = _roof(npair / ms_v1 * 1e3, mads["verifyk"], g1b_ + 32 + 1, prof, name + "_verifyk")
                # ONE message for the whole batch, a key per signature (tbls.Recover, sign/tbls/tbls.go:118-131): H(m)
                # hashed once per call; signatures k_i H(m) under the keys k_i G2 of this block
                m0 = msgs[0].contiguous()
                sigm, _ = m.g1_batch_mul(k, Hm[:1].repeat(npair, 1))
                okm, stm = m.batch_verify_g1_same_msg(Q, m0, sigm)
                ms_vm, = max_over_ranks(dist, [timed(lambda: m.batch_verify_g1_same_msg(Q, m0, sigm))])
                out[name]["bls_verify_same_msg_per_s"] = world * npair / ms_vm * 1e3
                out[name]["bls_verify_same_msg_all_true"] = bool(okm.all().item()) and not bool(stm.any().item())
                del Hm, sig1, sigm
        if name == "bls12381":
            # node-wide MSM at 2^20 points: points sharded over the ranks, all-gather of the partial points
            n = 1 << 20
            lo, hi = kd.shard_range(n, rank, world)
            ks = torch.from_numpy(be_scalars(b"kyberhip/v1/msm/k", n)[lo:hi].copy()).cuda()
            hs = torch.from_numpy(be_scalars(b"kyberhip/v1/msm/h", n)[lo:hi].copy()).cuda()
            pts, _ = m._mul(1, hs, g1b, True)
            if dist:
                fn = lambda: kd.bls12381_g1_msm(ks, pts)
            else:
                fn = lambda: m.g1_msm(ks, pts)
            ms = timed(fn)
            # the reference's MSM-shaped call sites (PubPoly.Eval, bdn aggregation) sum kyber.Points that were
            # validated when unmarshalled: same MSM with the per-point subgroup re-check off, and with the points
            # kept in the uncompressed form (no square root either)
            tr = m.F_TRUSTED(0)
            pts_u, _ = m._mul(1, hs, g1b, True, m.F_UNCOMPRESSED_OUT)
            if dist:
                fn_t = lambda: kd.bls12381_g1_msm(ks, pts, flags=tr)
                fn_u = lambda: kd.bls12381_g1_msm(ks, pts_u, flags=tr | m.F_UNCOMPRESSED)
            else:
                fn_t = lambda: m.g1_msm(ks, pts, tr)
                fn_u = lambda: m.g1_msm(ks, pts_u, tr | m.F_UNCOMPRESSED)
            # the expectation is independent of the bucket pipeline: P_i = h_i G, so sum k_i P_i = (sum k_i h_i mod r) G
            # (share/poly.go:340-348: the reference's N x (Mul + Add)), big-integer arithmetic on the host + ONE
            # fixed-base multiplication; all three calling conventions must give exactly these bytes
            k_all, h_all = be_scalars(b"kyberhip/v1/msm/k", n), be_scalars(b"kyberhip/v1/msm/h", n)
            tot = sum(int.from_bytes(bytes(a), "big") * int.from_bytes(bytes(b), "big") for a, b in zip(k_all, h_all)) % m.ORDER
            expect = bytes(np.asarray(m.g1_commit(tot.to_bytes(32, "big"))[0])[0])
            same = all(bytes(f()[0].cpu().numpy()) == expect for f in (fn, fn_t, fn_u))
            ms_t, ms_u = timed(fn_t), timed(fn_u)
            t = max_over_ranks(dist, [ms, ms_t, ms_u])
            out["bls12381_g1_msm_2p20"] = {"points": n, "seconds": t[0] * 1e-3,
                                           "seconds_validated_points": t[1] * 1e-3,
                                           "seconds_validated_uncompressed_points": t[2] * 1e-3,
                                           "matches_sum_ki_hi_times_G": same, "scaling": "strong",
                                           # SURVEY.md section 8d prices configs[2] at 80 B per point: 32-byte scalar +
                                           # 48-byte compressed point, so the decompression (a 379-bit power) is inside
                                           "roofline": _roof(n / t[1] * 1e3, MADS_MSM_BLS_G1_2P20_PER_POINT + MADS_G1_DECOMPRESS, 32 + 48, prof, "bls12381_g1_msm"),
                                           "roofline_uncompressed_points": _roof(n / t[2] * 1e3, MADS_MSM_BLS_G1_2P20_PER_POINT, 32 + 96, prof, "bls12381_g1_msm"),
                                           "exchange": "all-gather of %d encoded partial points" % world if dist else "none"}
            # share.PriPoly.Commit (share/poly.go:143-149): the same n coefficients times ONE base -- an arbitrary
            # point of the group, unmarshalled by the call like any base -- through the fixed-base table
            # (kyber_amd/csrc/fixed_base.cuh: 33 table additions per coefficient, no doublings); checked against the
            # variable-base kernels on a sample; weak scaling like every independent batch
            cb = torch.from_numpy(np.asarray(m.g1_commit((0x1234567).to_bytes(32, "big"))[0])[0].copy()).cuda()
            fn_c = lambda: m.g1_commit(ks, cb)
            ms_c = timed(fn_c)
            nchk = min(4096, int(ks.shape[0]))
            ok_c = bool(torch.equal(fn_c()[0][:nchk], m.g1_batch_mul(ks[:nchk], cb.repeat(nchk, 1))[0]))
            ms_c, = max_over_ranks(dist, [ms_c])
            out["bls12381_g1_commit_2p20"] = {"coefficients": n, "seconds": ms_c * 1e-3,
                                              "commits_per_s": n / ms_c * 1e3,
                                              "matches_variable_base_kernels": ok_c,
                                              "oracle_sample": "cpu_baseline.other_workloads.bls12381_g1_commit_oracle_sample",
                                              "scaling": "strong",
                                              "roofline": _roof(n / ms_c * 1e3, MADS_G1_COMMIT, 32 + 48, prof, "bls12381_g1_commit")}
            del ks, hs, pts, pts_u
    # Ed25519 MSM at 2^20 points (PubPoly.Eval / RecoverCommit shape), sharded like the BLS one
    from kyber_amd.group import edwards25519 as ed

    n = 1 << 20
    lo, hi = kd.shard_range(n, rank, world)
    s_all, h_all = make_inputs(n, 1000)
    ks = torch.from_numpy(s_all[lo:hi].copy()).cuda()
    pts = ed.batch_mul_base(torch.from_numpy(h_all[lo:hi].copy()).cuda())
    fn = (lambda: kd.ed25519_msm(ks, pts)) if dist else (lambda: ed.msm(ks, pts))
    ms, = max_over_ranks(dist, [timed(fn)])
    out["ed25519_msm_2p20"] = {"points": n, "seconds": ms * 1e-3, "scaling": "strong"}
    # KYB_F_UNIFORM (round 6): the same 2^20 multiplications with the window tables scanned, not indexed -- the access pattern
    # of the reference's default constant-time Mul (ge.go:352-371, 419-435), for secret scalars; same bytes as the default
    ms_uv, ms_uf = max_over_ranks(dist, [timed(lambda: ed.batch_mul(ks, pts, uniform=True), reps=10, warm=2),
                                         timed(lambda: ed.batch_mul_base(ks, uniform=True), reps=10, warm=2)])
    same = bool(torch.equal(ed.batch_mul(ks, pts, uniform=True)[0], ed.batch_mul(ks, pts)[0]) and
                torch.equal(ed.batch_mul_base(ks, uniform=True), ed.batch_mul_base(ks)))
    per = hi - lo
    out["ed25519_uniform"] = {"var_base_per_s": world * per / ms_uv * 1e3, "fixed_base_per_s": world * per / ms_uf * 1e3,
                              "outputs_equal_default": same, "elements_per_gpu": per}
    return out


LINE_LIMIT = 4096  # bytes of the one stdout line (a 20 KB line no longer parsed)


def _g(d, *path):
    for k in path:
        if not isinstance(d, dict):
            return None
        d = d.get(k)
    return d


def _r4(x):
    """4 significant digits for the stdout line (the detail file keeps full precision)"""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    return float("%.4g" % x)


def compact_line(res: dict) -> dict:
    """The one stdout line, built from the full record: headline keys of the bench contract, ONE roofline object
    (no sentences), a scalar-only cpu_baseline, and the composite metric's other two thirds as scalars, each with
    the fraction of the integer-MAD peak beside it.  Everything else lives in bench_detail.json."""
    ro, cb, ow = res.get("roofline") or {}, res.get("cpu_baseline") or {}, res.get("other_workloads") or {}
    line = {k: res.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                    "scaling", "vs_baseline", "dtype", "data")}
    line["config"] = {"workload": _g(res, "config", "workload"), "elements_per_gpu": _g(res, "config", "elements_per_gpu")}
    line["rccl_ranks_seen"] = res.get("rccl_ranks_seen")
    line["roofline"] = {"bound": ro.get("bound"), "kernel": ro.get("kernel_symbol") or "ed25519_mul_kernel<true, false>", "achieved": ro.get("achieved"),
                        "peak": ro.get("peak"), "unit": ro.get("unit"), "frac": _r4(ro.get("frac")),
                        "mads_per_op": ro.get("imads_per_op"), "kernel_ms": _r4(_g(res, "detail", "var_base_kernel_ms")),
                        "valu_busy_profiled": _r4(ro.get("valu_busy_profiled")), "traffic": ro.get("traffic"),
                        "hbm_frac": _r4(_g(ro, "hbm", "frac")), "traffic_ratio": _r4(ro.get("traffic_ratio")),
                        "hbm_frac_counters": _r4(ro.get("hbm_frac_counters")), "frac_vs_4cycle_issue": _r4(ro.get("frac_vs_4cycle_issue")),
                        "profile": ro.get("profile")}
    if cb:
        line["cpu_baseline"] = {"value": _r4(cb.get("value")), "unit": cb.get("unit"), "cores": cb.get("cores"), "kind": cb.get("kind"),
                                "cpu_model": cb.get("cpu_model"), "quota_cores": cb.get("cgroup_cpu_quota_cores"),
                                "outputs_match": cb.get("outputs_match"), "outputs_compared": cb.get("outputs_compared"),
                                "sample": "first %s elements of the GPU batch, fixed + var base, oracle/ed25519_ref.c" % (
                                    (cb.get("outputs_compared") or 0) // 2)}
    if ow:
        b, n6, mm, cm = ow.get("bls12381") or {}, ow.get("bn256") or {}, ow.get("bls12381_g1_msm_2p20") or {}, ow.get("bls12381_g1_commit_2p20") or {}
        cbo = cb.get("other_workloads") or {}
        sc = {  # name: (value, fraction of the integer-MAD peak)
            "bls12381_pairings_per_s": (b.get("pairings_per_s"), _g(b, "roofline", "pair", "frac")),
            "bls12381_pairings_per_s_trusted": (b.get("pairings_per_s_validated_inputs"), _g(b, "roofline", "pair_validated_inputs", "frac")),
            "bls12381_pair_checks_per_s": (b.get("pairing_checks_per_s"), _g(b, "roofline", "pair_check", "frac")),
            "bls12381_verifies_per_s": (b.get("bls_verify_pipeline_per_s"), _g(b, "roofline", "verify", "frac")),
            "bls12381_verifies_per_s_same_key": (b.get("bls_verify_same_key_per_s"), _g(b, "roofline", "verify_same_key", "frac")),
            "bls12381_verifies_per_s_same_msg": (b.get("bls_verify_same_msg_per_s"), None),
            "bls12381_g1_muls_per_s": (b.get("g1_muls_per_s"), _g(b, "roofline", "g1_mul", "frac")),
            "bls12381_g2_muls_per_s": (b.get("g2_muls_per_s"), _g(b, "roofline", "g2_mul", "frac")),
            # configs[2] at SURVEY 8d's 80 B per point (32-byte scalar + 48-byte compressed point, validated when it
            # was unmarshalled); _checked = the subgroup test repeated per point; _affine = 96-byte points, no square root
            "bls12381_g1_msm_2p20_s": (mm.get("seconds_validated_points"), _g(mm, "roofline", "frac")),
            "bls12381_g1_msm_2p20_s_checked": (mm.get("seconds"), None),
            "bls12381_g1_msm_2p20_s_affine": (mm.get("seconds_validated_uncompressed_points"), _g(mm, "roofline_uncompressed_points", "frac")),
            "bls12381_g1_commit_2p20_s": (cm.get("seconds"), _g(cm, "roofline", "frac")),
            "bn256_pairings_per_s": (n6.get("pairings_per_s"), _g(n6, "roofline", "pair", "frac")),
            "bn256_g1_muls_per_s": (n6.get("g1_muls_per_s"), _g(n6, "roofline", "g1_mul", "frac")),
            "bn256_g2_muls_per_s": (n6.get("g2_muls_per_s"), _g(n6, "roofline", "g2_mul", "frac")),
            "ed25519_msm_2p20_s": (_g(ow, "ed25519_msm_2p20", "seconds"), None),
            "ed25519_uniform_var_base_per_s": (_g(ow, "ed25519_uniform", "var_base_per_s"), None),
            "ed25519_uniform_fixed_base_per_s": (_g(ow, "ed25519_uniform", "fixed_base_per_s"), None),
        }
        for k, (v, f) in sc.items():
            line[k] = _r4(v)
            if f is not None:
                line[k + "_frac"] = _r4(f)
        line["checks"] = {"bls12381_all_true": b.get("all_checks_true"), "bn256_all_true": n6.get("all_checks_true"),
                          "msm_matches_expectation": mm.get("matches_sum_ki_hi_times_G"),
                          "commit_matches_var_base": cm.get("matches_variable_base_kernels"),
                          "uniform_equals_default": _g(ow, "ed25519_uniform", "outputs_equal_default"),
                          "commit_matches_oracle": _g(cbo, "bls12381_g1_commit_oracle_sample", "outputs_match"),
                          "bls12381_pair_matches_cpu_port": _g(cbo, "bls12381_pairings", "outputs_match"),
                          "bn256_pair_matches_cpu_port": _g(cbo, "bn256_pairings", "outputs_match")}
        fd = cbo.get("full_batch_digests") or {}
        for suite, short in (("bls12381", "bls12381"), ("bn256", "bn256")):
            for leg in ("pair", "g1_mul", "g2_mul"):
                rec = _g(fd, suite, leg)
                if rec is not None:  # true only when every output of the config-size batch equals the oracle's
                    line["checks"]["%s_%s_full" % (short, leg)] = bool(rec.get("outputs_match")) and rec.get("outputs_compared")
        if _g(fd, "bls12381_g1_msm_2p20") is not None:
            mrec = fd["bls12381_g1_msm_2p20"]
            line["checks"]["msm_2p20_full"] = bool(mrec.get("outputs_match")) and mrec.get("outputs_compared")
        for key in ("bls12381", "bn256", "bls12381_g1_msm_2p20"):
            if _g(fd, key, "error"):
                line["checks"]["full_error"] = (key + ": " + fd[key]["error"])[:120]
        line["cpu_bls12381_pairings_per_s"] = _r4(_g(cbo, "bls12381_pairings", "value"))
        line["cpu_bn256_pairings_per_s"] = _r4(_g(cbo, "bn256_pairings", "value"))
        line["cpu_bls12381_g1_mul_add_2p20_s"] = _r4(_g(cbo, "bls12381_g1_mul_add", "seconds_for_2p20_points") or
                                                     _g(cbo, "bls12381_g1_mul_add", "seconds_for_2p20_points_extrapolated"))
    if res.get("other_workloads_error"):
        line["other_workloads_error"] = res["other_workloads_error"][:200]
    line["detail_file"] = "bench_detail.json"
    return line


def emit(res: dict):
    """full record to bench_detail.json (+ gpurun_out/), the compact line to stdout"""
    blob = json.dumps(res, indent=1)
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        if os.path.isdir(d):
            try:
                with open(os.path.join(d, "bench_detail.json"), "w") as f:
                    f.write(blob)
            except OSError:
                pass
    print(fit_line(compact_line(res)), flush=True)


def fit_line(rec: dict) -> str:
    """the record as one JSON line under LINE_LIMIT: optional keys go first (the *_frac figures, the cpu_* side figures,
    the checks object, then every non-contract scalar) and `truncated` says so -- a long error string or a new key must
    never cost the run its headline line"""
    contract = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline", "rccl_ranks_seen", "detail_file", "truncated")
    rec = dict(rec)
    line = json.dumps(rec, separators=(",", ":"))
    stages = (lambda k: k.endswith("_frac"), lambda k: k.startswith("cpu_"), lambda k: k in ("checks", "other_workloads_error"),
              lambda k: k not in contract)
    for drop in stages:
        if len(line) < LINE_LIMIT:
            break
        rec = {k: v for k, v in rec.items() if k in contract or not drop(k)}
        rec["truncated"] = True
        line = json.dumps(rec, separators=(",", ":"))
    if len(line) >= LINE_LIMIT:  # a contract object itself is oversized (an error string inside cpu_baseline): cut strings
        def cut(o):
            if isinstance(o, dict):
                return {k: cut(v) for k, v in o.items()}
            return o[:60] if isinstance(o, str) and len(o) > 60 else o
        rec = cut(rec)
        rec["metric"] = "scalar-muls/s + pairings/s per node; MSM sec at 2^20 points"
        line = json.dumps(rec, separators=(",", ":"))
    return line


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--n", type=int, default=N_PER_GPU, help="elements per GPU (default 2^20)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other", action="store_true", help="skip the pairing / MSM side measurements")
    ap.add_argument("--no-host-path", action="store_true",
                    help="skip the host-buffer (PCIe-inclusive) measurement, whose chunked launches of the same "
                         "kernels would blur per-kernel averages in a rocprofv3 trace")
    args = ap.parse_args()

    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    torch.cuda.set_device(local_rank)
    dist = None
    rccl_ranks_seen = None
    # KYB_BENCH_FORCE_DIST=1 exercises the RCCL path (init, barrier, all-gather) with a single rank
    if world > 1 or os.environ.get("KYB_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist

        # RCCL prints a version banner on stdout when its first communicator comes up: keep stdout for the one JSON
        # line by pointing fd 1 at stderr until a first collective has run
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            warm = torch.zeros(1, device="cuda")
            dist.all_reduce(warm)
            # every rank of the job really is on the communicator: gather the rank ids over RCCL and count them
            ids = [torch.zeros(1, dtype=torch.int64, device="cuda") for _ in range(dist.get_world_size())]
            dist.all_gather(ids, torch.tensor([rank], dtype=torch.int64, device="cuda"))
            rccl_ranks_seen = len({int(x.item()) for x in ids})
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved_fd, 1)
            os.close(saved_fd)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    from kyber_amd.group import edwards25519 as ed

    n = args.n
    s_h, h_h = make_inputs(n, rank)
    d_s = torch.from_numpy(s_h).cuda()
    d_h = torch.from_numpy(h_h).cuda()
    d_pts = ed.batch_mul_base(d_h)  # input points (compressed), resident in HBM
    torch.cuda.synchronize()

    def step(ev=None):
        out_fix = ed.batch_mul_base(d_s)
        if ev:
            ev[0].record()
        out_var, st = ed.batch_mul(d_s, d_pts)
        if ev:
            ev[1].record()
        return out_fix, out_var, st

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    # HIP events on the launch stream (torch's current stream is the one the C ABI is handed)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    ev0 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev0.record()
    for i in range(args.steps):
        outs = step(evs[i])
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    elapsed, = max_over_ranks(dist, [elapsed])

    var_ms = [a.elapsed_time(b) for a, b in evs]
    fix_ms = []
    prev = ev0
    for a, b in evs:
        fix_ms.append(prev.elapsed_time(a))
        prev = b
    var_ms_avg = sum(var_ms) / len(var_ms)
    fix_ms_avg = sum(fix_ms) / len(fix_ms)
    ok = int(outs[2].sum().item()) == 0

    # PCIe-inclusive rate of the host-buffer entry points (allocate, H2D, kernels, D2H, free): reported, never `value`
    # (median of 5 calls of the C entry points on buffers that are reused, as a service would: a freshly allocated
    # 32 MB result array costs another ~3 ms of first-touch page faults per call, which is the caller's allocator)
    host_rate = None
    if rank == 0 and not args.no_host_path:
        from kyber_amd import _lib

        lib = _lib.load()
        s_c = np.ascontiguousarray(s_h)
        p_c = np.ascontiguousarray(ed.batch_mul_base(s_h))
        o_c, st_c = np.zeros((n, 32), dtype=np.uint8), np.zeros(n, dtype=np.uint8)
        ts = []
        for _ in range(6):
            t0 = time.perf_counter()
            _lib.check(lib.kyb_ed25519_mul_base(n, s_c.ctypes.data, o_c.ctypes.data, 0), "kyb_ed25519_mul_base")
            _lib.check(lib.kyb_ed25519_mul(n, s_c.ctypes.data, p_c.ctypes.data, o_c.ctypes.data, st_c.ctypes.data, 0), "kyb_ed25519_mul")
            ts.append(time.perf_counter() - t0)
        host_rate = 2 * n / sorted(ts[1:])[2]

    other = None
    other_error = None
    if not args.no_other:
        # the side measurements must never cost the run its headline line: on a single rank an exception is recorded and
        # the line goes out without them (with several ranks it propagates -- a rank that skipped the collectives of the
        # side measurements would hang the others)
        try:
            other = other_workloads(rank, world, dist)
        except Exception as e:  # noqa: BLE001
            if dist:
                raise
            other_error = repr(e)[:500]

    if rank == 0:
        total_ops = 2 * n * args.steps * world
        var_s = var_ms_avg * 1e-3
        prof = _prof()
        imad_peak = prof.get("imad_peak_lane_ops_per_s")
        ked = prof.get("kernels", {}).get("ed25519_mul", {})
        res = {
            "metric": "scalar-muls/s + pairings/s per node; MSM sec at 2^20 points",
            "value": total_ops / elapsed,
            "unit": "scalar-muls/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "rccl_ranks_seen": rccl_ranks_seen,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak",
            # BASELINE.md section 1 (docs/benchmark-app data.json:38-40, 14-16): one variable-base + one fixed-base
            # Point.Mul cost the reference 349 399 + 60 658 ns on ONE core of unstated hardware = 4 877 scalar-muls/s
            "vs_baseline": total_ops / elapsed / REF_SINGLE_CORE_PER_S,
            "vs_baseline_note": "reference: 2 / (349399 ns + 60658 ns) = 4877 scalar-muls/s, single core, hardware "
                                "unstated (BASELINE.md section 1); per node the reference publishes nothing",
            "dtype": "int32 limbs (radix 2^25.5), int64 accumulate", "data": "synthetic",
            "config": {"workload": "Ed25519 batched fixed-base + var-base scalar-mul, 2^20 scalars per GPU "
                                   "(BASELINE.json configs[1])",
                       "elements_per_gpu": n, "sharding": f"independent batches x{world}, no collective"},
            "detail": {"var_base_per_s_per_gpu": n / var_s,
                       "fixed_base_per_s_per_gpu": n / (fix_ms_avg * 1e-3),
                       "var_base_kernel_ms": var_ms_avg, "fixed_base_kernel_ms": fix_ms_avg,
                       "all_status_ok": ok,
                       "host_buffer_path_scalar_muls_per_s": host_rate},
            "roofline": {"bound": "valu-imad",
                         "kernel": "ed25519_mul_kernel (variable-base, dominant: ~85% of a step)",
                         "kernel_symbol": "ed25519_mul_kernel<true, false>",
                         "binding_resource": "integer VALU issue (v_mad_i64_i32 at half rate): 96 algorithmic bytes per "
                                             "2.06e5 integer MADs, no dense contraction for MFMA",
                         "imads_per_op": IMADS_VAR, "achieved": IMADS_VAR * n / var_s, "peak": imad_peak,
                         "unit": "lane-MAD/s", "frac": (IMADS_VAR * n / var_s / imad_peak) if imad_peak else None,
                         "peak_source": prof.get("imad_peak_source"),
                         "valu_busy_profiled": ked.get("valu_busy"), "valu_insts_per_op_profiled": ked.get("valu_insts_per_unit"),
                         "traffic": ked.get("hbm_bytes_per_launch"), "profile": ked.get("source"),
                         # counter bytes per launch over the algorithmic 96 B per element, and what that traffic is of the
                         # HBM peak at this launch's duration (hbm.frac below is the ALGORITHMIC figure)
                         "traffic_ratio": (ked["hbm_bytes_per_launch"] / (BYTES_VAR * ked.get("units_per_launch", n))) if ked.get("hbm_bytes_per_launch") else None,
                         "hbm_frac_counters": (ked["hbm_bytes_per_launch"] * (n / ked.get("units_per_launch", n)) / var_s / 1e9 / HBM_PEAK_GBS) if ked.get("hbm_bytes_per_launch") else None,
                         # the denominator of `frac` is the MEASURED dependent v_mad_u64_u32 rate; against the paper figure
                         # (256 CUs x 4 SIMDs x 64 lanes / 4 cycles x 2.4 GHz = 3.93e13 lane-MAD/s) the fraction is:
                         "frac_vs_4cycle_issue": IMADS_VAR * n / var_s / IMAD_4CYCLE_PEAK,
                         "hbm": {"achieved": BYTES_VAR * n / var_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": BYTES_VAR * n / var_s / 1e9 / HBM_PEAK_GBS}},
        }
        if other is not None:
            res["other_workloads"] = other
        if other_error:
            res["other_workloads_error"] = other_error
        if world == 1 and not args.no_cpu_baseline:
            try:
                res["cpu_baseline"] = cpu_baseline(s_h, d_pts.cpu().numpy(), outs[0].cpu().numpy(), outs[1].cpu().numpy())
            except Exception as e:  # noqa: BLE001 -- e.g. the oracle's library missing on the box: say so, keep the line
                res["cpu_baseline"] = {"value": None, "unit": "scalar-muls/s", "cores": 0, "kind": "port", "outputs_match": None,
                                       "outputs_compared": 0, "error": repr(e)[:300]}
            res["outputs_match"] = res["cpu_baseline"]["outputs_match"]
            if not args.no_other:
                from kyber_amd.pairing import bls12381 as bls_, bn256 as bn_

                try:
                    res["cpu_baseline"]["other_workloads"] = cpu_baseline_pairing_and_msm(bn_, bls_)
                    res["cpu_baseline"]["other_workloads"]["reference_published_bls_verify_per_s_single_core"] = REF_BLS_VERIFY_PER_S_SINGLE_CORE
                except Exception as e:  # noqa: BLE001 -- a reported baseline must never take the benchmark line down
                    res["cpu_baseline"]["other_workloads"] = {"error": repr(e)[:500]}
        emit(res)
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
