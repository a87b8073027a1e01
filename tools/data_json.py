#!/usr/bin/env python3
"""The engine's batch figures in the schema of the reference's benchmark collection (benchmark/README.md:63-80,
benchmark/benchmark.go:22-90: results[module][instance] = {group, description, benchmarks: {type: {operation:
{N, T, ...}}}}, T in nanoseconds for N operations, as testing.BenchmarkResult marshals) -- measured through the Python
mirror on the GPU this runs on, host buffers in and out (what a cgo caller gets).  The Go program that emits the same
file from inside the reference (go/kyberhip/cmd/kyberhip-bench) cannot be compiled where this repository is built.

usage: python tools/data_json.py [out.json] [log2 batch]"""
import hashlib
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def rows(label, n, w=32):
    return np.frombuffer(hashlib.shake_256(label).digest(n * w), dtype=np.uint8).reshape(n, w).copy()


def bench(fn, n, reps=5):
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter_ns()
        fn()
        ts.append(time.perf_counter_ns() - t0)
    t = sorted(ts)[reps // 2]
    return {"N": n, "T": t, "Bytes": 0, "MemAllocs": 0, "MemBytes": 0, "Extra": {}}


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else "data.hip.json"
    n = 1 << (int(sys.argv[2]) if len(sys.argv) > 2 else 16)
    from kyber_amd.group import edwards25519 as ed
    from kyber_amd.pairing import bls12381, bn254, bn256

    results = {"groups": {}}
    desc = "MI355X batch engine (libkyberhip) behind the kyber interfaces; variable-time; one call over %d elements, host buffers" % n
    s = rows(b"dj/ed/s", n)
    s[:, 31] &= 0x0F
    P = np.ascontiguousarray(ed.batch_mul_base(s))
    results["groups"]["Ed25519.hip"] = {"group": "Ed25519.hip", "description": desc, "benchmarks": {"batch": {
        "mul": bench(lambda: ed.batch_mul(s, P), n), "baseMul": bench(lambda: ed.batch_mul_base(s), n),
        "msm": bench(lambda: ed.msm(s, P), n)}}}
    for name, m in (("bls12-381.hip", bls12381), ("bn256.hip", bn256), ("bn254.hip", bn254)):
        k = rows(b"dj/k/" + name.encode(), n)
        k[:, 0] &= 0x3F
        P1 = np.ascontiguousarray(m.g1_commit(k)[0])
        P2 = np.ascontiguousarray(m.g2_commit(k)[0])
        T = m.F_TRUSTED(0)  # kyber.Points were validated when they were unmarshalled
        sig = np.ascontiguousarray(m.g1_batch_mul(k, P1, T)[0])
        G2 = np.tile(np.frombuffer(m.G2_BASE, dtype=np.uint8), (n, 1))
        results["groups"][name] = {"group": name, "description": desc, "benchmarks": {"batch": {
            "g1Mul": bench(lambda: m.g1_batch_mul(k, P1, T), n), "g2Mul": bench(lambda: m.g2_batch_mul(k, P2, T), n),
            "g1BaseMul": bench(lambda: m.g1_commit(k), n), "g1Msm": bench(lambda: m.g1_msm(k, P1, T), n),
            "pair": bench(lambda: m.batch_pair(P1, P2, T | m.F_TRUSTED(1)), n),
            "validatePairing": bench(lambda: m.batch_validate_pairing(P1, P2, sig, G2, m.F_TRUSTED_ALL), n)}}}
    json.dump(results, open(out, "w"), indent=2)
    for g, v in results["groups"].items():
        print(g, {op: round(r["T"] / r["N"], 1) for op, r in v["benchmarks"]["batch"].items()}, "ns/op")


if __name__ == "__main__":
    main()
