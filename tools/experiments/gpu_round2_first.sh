#!/bin/bash
# First GPU call of round 2: the changes made after round 1's GPU budget ran out have only static evidence
# (instruction counts).  Parity first, then the timings that DESIGN.md section 5 items 3a / 3b predict.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r2first; export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/r2first/pytest_gpu.log 2>&1; tail -3 gpurun_out/r2first/pytest_gpu.log
timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-other --no-host-path > gpurun_out/r2first/bench_ed.json 2> gpurun_out/r2first/bench_ed.err; cat gpurun_out/r2first/bench_ed.json
for s in bls12381 bn256; do timeout 120 python tools/pair_probe.py $s 65536 2>/dev/null | tail -1 | tee gpurun_out/r2first/probe_$s.json; done
timeout 200 python tools/msm_probe.py 1048576 2>/dev/null | tail -1 | tee gpurun_out/r2first/msm_probe_2p20.json
# round 1 (profiles/r01_final_*): ed25519 var-base 12.8-13.3 ms / 2^20, fixed-base 1.66 ms; BLS12-381 pairings 1.72e6/s,
# checks 1.25-1.28e6/s, G1 / G2 mul 1.10e7 / 4.8e6 /s; bn256 pairings 3.23e6/s; MSM 2^20 BLS G1 34 ms checked
