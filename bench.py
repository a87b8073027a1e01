#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path on MI355X.

Workload (BASELINE.json configs[1]): Ed25519 batched fixed-base + variable-base
scalar multiplication, 2^20 scalars per GPU.  One "step" = one pass of the hot
path over one batch: 2^20 x Point.Mul(s, nil) (fixed-base) followed by
2^20 x Point.Mul(s, P) (variable-base, compressed points in, compressed out),
inputs already resident in HBM.  value = scalar-muls/s over the whole job.

N > 1 (launched by torch.distributed.run, one rank per GPU): the batch shards
trivially -- every rank owns its own 2^20-element batch (weak scaling); there
is no data-path collective for independent scalar multiplications, only the
timing barrier and a MAX all-reduce of the elapsed time.

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes
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
# field multiplications (M) / squarings (S) per variable-base op as built:
#   decode 14M+255S... counted exactly in DESIGN.md; products per M = 100, per S = 55 (v_mad_i64_i32)
IMADS_VAR = None  # filled from DESIGN.md numbers below
IMAD_PEAK = None  # lane-MADs/s measured by tools/valu_peak (profiles/valu_peak_r01.json)


def shake(label: bytes, nbytes: int) -> np.ndarray:
    return np.frombuffer(hashlib.shake_256(label).digest(nbytes), dtype=np.uint8)


def make_inputs(n: int, rank: int):
    """Deterministic synthetic inputs: canonical scalars (< 2^252 <= l) from
    labelled SHAKE-256 streams; points P_i = h_i * B produced by the verified
    fixed-base path (prime-order subgroup, like util/key/key.go:41-49)."""
    s = shake(b"kyberhip/v1/ed25519/scalars/%d" % rank, n * 32).reshape(n, 32).copy()
    h = shake(b"kyberhip/v1/ed25519/point-seeds/%d" % rank, n * 32).reshape(n, 32).copy()
    s[:, 31] &= 0x0F
    h[:, 31] &= 0x0F
    return s, h


def cpu_baseline(scalars: np.ndarray, points: np.ndarray):
    """The oracle's C restatement (kind "port") on all host cores, bounded sample."""
    from tests import _oracle_c as OC

    cores = os.cpu_count() or 1
    probe = 512
    t0 = time.perf_counter()
    OC.ed_mul(scalars[:probe], points[:probe], threads=cores)
    OC.ed_mul_base(scalars[:probe], threads=cores)
    dt = time.perf_counter() - t0
    target_s = 6.0
    n = int(min(len(scalars), max(probe, probe * target_s / max(dt, 1e-6))))
    t0 = time.perf_counter()
    OC.ed_mul(scalars[:n], points[:n], threads=cores)
    OC.ed_mul_base(scalars[:n], threads=cores)
    dt = time.perf_counter() - t0
    return {"value": 2 * n / dt, "unit": "scalar-muls/s", "cores": cores, "kind": "port",
            "sample": f"{n} fixed-base + {n} variable-base Ed25519 scalar-muls of the same batch, "
                      f"oracle/ed25519_ref.c (radix-2^51 C restatement of ge.go:373/443), {cores} threads"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--n", type=int, default=N_PER_GPU, help="elements per GPU (default 2^20)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
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
    if dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # per-kernel durations from HIP events on the launch stream
    var_ms = [a.elapsed_time(b) for a, b in evs]
    fix_ms = []
    prev = ev0
    for a, b in evs:
        fix_ms.append(prev.elapsed_time(a))
        prev = b
    var_ms_avg = sum(var_ms) / len(var_ms)
    fix_ms_avg = sum(fix_ms) / len(fix_ms)

    ok = int(outs[2].sum().item()) == 0
    if rank == 0:
        total_ops = 2 * n * args.steps * world
        res = {
            "metric": "scalar-muls/s + pairings/s per node; MSM sec at 2^20 points",
            "value": total_ops / elapsed,
            "unit": "scalar-muls/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int32 limbs (radix 2^25.5), int64 accumulate", "data": "synthetic",
            "config": {"workload": "Ed25519 batched fixed-base + var-base scalar-mul, 2^20 scalars per GPU "
                                   "(BASELINE.json configs[1])",
                       "elements_per_gpu": n, "sharding": f"independent batches x{world}, no collective"},
            "detail": {"var_base_per_s_per_gpu": n / (var_ms_avg * 1e-3),
                       "fixed_base_per_s_per_gpu": n / (fix_ms_avg * 1e-3),
                       "var_base_kernel_ms": var_ms_avg, "fixed_base_kernel_ms": fix_ms_avg,
                       "all_status_ok": ok},
            "roofline": {"bound": "hbm", "kernel": "ed25519_mul_kernel (variable-base)",
                         "achieved": BYTES_VAR * n / (var_ms_avg * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": BYTES_VAR * n / (var_ms_avg * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "traffic": None,
                         "note": "integer-VALU bound, not HBM bound: see valu sub-object and DESIGN.md"},
        }
        try:
            prof = json.load(open(os.path.join(ROOT, "profiles", "valu_model.json")))
            imads = prof["imads_per_var_base_op"]
            peak = prof["imad_peak_lane_ops_per_s"]
            ach = imads * n / (var_ms_avg * 1e-3)
            res["roofline"]["valu"] = {"achieved": ach, "peak": peak, "unit": "v_mad_i64_i32 lane-ops/s",
                                       "frac": ach / peak, "traffic_bytes_per_launch": prof.get("hbm_bytes_per_launch")}
            res["roofline"]["traffic"] = prof.get("hbm_bytes_per_launch")
        except Exception:
            pass
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(s_h, d_pts.cpu().numpy())
        print(json.dumps(res))
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
