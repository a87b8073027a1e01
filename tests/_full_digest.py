"""Whole-batch parity at BASELINE.json's config sizes (SURVEY.md section 8d "Correctness at scale"): SHA-256 over the
concatenated outputs of the engine against the same digest over the C oracle's outputs (oracle/*.c through
tests/_oracle_c.py), every element of the batch compared -- not a sample.  Test infrastructure: used by
tests/test_gpu_full_digest.py and by bench.py's cpu_baseline leg (which reports the oracle's timing over the same
batch as the CPU figure); never by the product.

The inputs are the labelled SHAKE-256 streams bench.py measures on, so the batch whose rate is reported is the batch
whose bytes are compared."""
import hashlib
import importlib
import time

import numpy as np


def shake(label: bytes, nbytes: int) -> np.ndarray:
    return np.frombuffer(hashlib.shake_256(label).digest(nbytes), dtype=np.uint8)


def be_scalars(label: bytes, n: int) -> np.ndarray:
    """n big-endian 32-byte scalars < 2^254 (bench.py's be_scalars)"""
    a = shake(label, n * 32).reshape(n, 32).copy()
    a[:, 0] &= 0x3F
    return a


def _sha(a) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _cmp(name, gpu, cpu, st_gpu, st_cpu, dt, n, what):
    gpu, cpu = np.asarray(gpu), np.asarray(cpu)
    h_g, h_c = _sha(gpu), _sha(cpu)
    rec = {"outputs_match": bool(h_g == h_c and not np.asarray(st_gpu).any() and not np.asarray(st_cpu).any()),
           "outputs_compared": int(n), "sha256": h_c, "cpu_seconds": dt, "cpu_per_s": n / dt if dt > 0 else None, "what": what}
    if h_g != h_c:
        bad = np.nonzero((gpu.reshape(n, -1) != cpu.reshape(n, -1)).any(axis=1))[0]
        rec["first_mismatches"] = [int(i) for i in bad[:8]]
        rec["mismatch_count"] = int(bad.size)
    return name, rec


def pairing_suite(name: str, n: int, rank: int = 0, threads: int = 0, legs=("pair", "g1_mul", "g2_mul")) -> dict:
    """configs[3] (bls12381, n = 2^16) / configs[4] (bn256, n = 2^18): Suite.Pair, G1 Mul and G2 Mul over the WHOLE batch,
    flags = 0 (every operand re-validated as UnmarshalBinary would), engine against the C oracle"""
    import torch

    from tests import _oracle_c as OC

    m = importlib.import_module("kyber_amd.pairing." + name)
    threads = threads or OC.host_threads()
    k = be_scalars(b"kyberhip/v1/%s/k/%d" % (name.encode(), rank), n)
    h = be_scalars(b"kyberhip/v1/%s/h/%d" % (name.encode(), rank), n)
    dk, dh = torch.from_numpy(k).cuda(), torch.from_numpy(h).cuda()
    g1b = torch.from_numpy(np.frombuffer(m.G1_BASE, dtype=np.uint8).copy()).cuda()
    g2b = torch.from_numpy(np.frombuffer(m.G2_BASE, dtype=np.uint8).copy()).cuda()
    P, st1 = m._mul(1, dh, g1b, True)
    Q, st2 = m._mul(2, dk, g2b, True)
    assert not st1.any().item() and not st2.any().item()
    Ph, Qh = P.cpu().numpy(), Q.cpu().numpy()
    if name == "bls12381":
        f_pair, f_g1, f_g2 = OC.bls12381_pair_compressed, OC.bls12381_g1_mul, OC.bls12381_g2_mul
        src = "oracle/bls12381_pair_ref.c"
    elif name == "bn256":
        f_pair, f_g1, f_g2 = OC.bn256_pair, OC.bn256_g1_mul, OC.bn256_g2_mul
        src = "oracle/bn256_ref.c"
    else:
        raise ValueError("no C oracle for " + name)
    out = {"n": n, "threads": threads, "oracle": src}
    if "pair" in legs:
        gt, st = m.batch_pair(P, Q)
        gt, st = gt.cpu().numpy(), st.cpu().numpy()
        t0 = time.perf_counter()
        gt_c, st_c = f_pair(Ph, Qh, threads=threads)
        dt = time.perf_counter() - t0
        key, rec = _cmp("pair", gt, gt_c, st, st_c, dt, n, f"{n} x Suite.Pair, GT bytes")
        out[key] = rec
        del gt, gt_c
    if "g1_mul" in legs:
        r, st = m.g1_batch_mul(dk, P)
        t0 = time.perf_counter()
        r_c, st_c = f_g1(k, Ph, threads=threads)
        dt = time.perf_counter() - t0
        key, rec = _cmp("g1_mul", r.cpu().numpy(), r_c, st.cpu().numpy(), st_c, dt, n, f"{n} x G1 Point.Mul, encoded outputs")
        out[key] = rec
    if "g2_mul" in legs:
        r, st = m.g2_batch_mul(dh, Q)
        t0 = time.perf_counter()
        r_c, st_c = f_g2(h, Qh, threads=threads)
        dt = time.perf_counter() - t0
        key, rec = _cmp("g2_mul", r.cpu().numpy(), r_c, st.cpu().numpy(), st_c, dt, n, f"{n} x G2 Point.Mul, encoded outputs")
        out[key] = rec
    out["all_match"] = all(out[x]["outputs_match"] for x in legs)
    return out


def bls12381_g1_msm(n: int = 1 << 20, threads: int = 0) -> dict:
    """configs[2]: the engine's Pippenger MSM over n compressed points (80 bytes per point, every calling convention)
    against the reference's own shape for that sum -- N x (Point.Mul + Point.Add), share/poly.go:340-348 -- run by the
    C oracle over the same n (scalar, point) pairs"""
    import torch

    from kyber_amd.pairing import bls12381 as m
    from tests import _oracle_c as OC

    threads = threads or OC.host_threads()
    k, h = be_scalars(b"kyberhip/v1/msm/k", n), be_scalars(b"kyberhip/v1/msm/h", n)
    dk, dh = torch.from_numpy(k).cuda(), torch.from_numpy(h).cuda()
    g1b = torch.from_numpy(np.frombuffer(m.G1_BASE, dtype=np.uint8).copy()).cuda()
    P, st = m._mul(1, dh, g1b, True)
    assert not st.any().item()
    Pu, st = m._mul(1, dh, g1b, True, m.F_UNCOMPRESSED_OUT)
    assert not st.any().item()
    got = {"checked": m.g1_msm(dk, P), "vouched": m.g1_msm(dk, P, m.F_TRUSTED(0)),
           "uncompressed": m.g1_msm(dk, Pu, m.F_TRUSTED(0) | m.F_UNCOMPRESSED)}
    t0 = time.perf_counter()
    sum_c, st_c = OC.bls12381_g1_mul_sum_compressed(k, P.cpu().numpy(), threads=threads)
    dt = time.perf_counter() - t0
    ok = {c: bool(bytes(o.cpu().numpy()) == bytes(sum_c) and not s.any().item()) for c, (o, s) in got.items()}
    return {"outputs_match": bool(all(ok.values()) and not st_c.any()), "per_convention": ok, "outputs_compared": n,
            "points_summed": n, "sum": bytes(sum_c).hex(), "cpu_seconds": dt, "cpu_points_per_s": n / dt, "threads": threads,
            "what": f"sum of {n} x (Point.Mul + Point.Add) over compressed points, oracle/bls12381_pair_ref.c, against the "
                    f"engine's MSM of the same {n} points (checked, vouched-for, uncompressed)"}


def ed25519_config1(n: int = 1 << 20, threads: int = 0) -> dict:
    """configs[1]: 2^20 fixed-base + 2^20 variable-base Ed25519 scalar-muls against oracle/ed25519_ref.c"""
    import torch

    from kyber_amd.group import edwards25519 as ed
    from tests import _oracle_c as OC

    threads = threads or OC.host_threads()
    s = shake(b"kyberhip/v1/ed25519/scalars/0", n * 32).reshape(n, 32).copy()
    h = shake(b"kyberhip/v1/ed25519/point-seeds/0", n * 32).reshape(n, 32).copy()
    s[:, 31] &= 0x0F
    h[:, 31] &= 0x0F
    d_s = torch.from_numpy(s).cuda()
    pts = ed.batch_mul_base(torch.from_numpy(h).cuda())
    fix = ed.batch_mul_base(d_s).cpu().numpy()
    var, st = ed.batch_mul(d_s, pts)
    ph = pts.cpu().numpy()
    t0 = time.perf_counter()
    var_c, st_c = OC.ed_mul(s, ph, threads=threads)
    fix_c = OC.ed_mul_base(s, threads=threads)
    pts_c = OC.ed_mul_base(h, threads=threads)
    dt = time.perf_counter() - t0
    ok = _sha(fix) == _sha(fix_c) and _sha(var.cpu().numpy()) == _sha(var_c) and _sha(ph) == _sha(pts_c)
    return {"outputs_match": bool(ok and not st.any().item() and not st_c.any()), "outputs_compared": 3 * n, "cpu_seconds": dt,
            "threads": threads, "what": f"{n} variable-base + 2 x {n} fixed-base scalar-muls, oracle/ed25519_ref.c"}
