#!/bin/bash
# A/B: library builds kyber_amd/lib/libkyberhip_<v>.so against each other on one box: per-suite probes + the bench
mkdir -p gpurun_out/r03_ab; rocm-smi --showclocks --showperflevel --showpower --showmemuse > gpurun_out/r03_ab/smi_$(date +%s).txt 2>&1
for v in ${VARIANTS:-a b c d}; do
  L=$PWD/kyber_amd/lib/libkyberhip_$v.so
  for s in bls12381 bn256 bn254; do
    KYBER_HIP_LIB=$L timeout 300 python tools/ab_probe.py $s 65536 5 > gpurun_out/r03_ab/probe_${s}_$v.json 2> gpurun_out/r03_ab/probe_${s}_$v.err
  done
  [ -n "$SKIP_BENCH" ] || KYBER_HIP_LIB=$L timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-path > gpurun_out/r03_ab/bench_$v.json 2> gpurun_out/r03_ab/bench_$v.err
done
cat gpurun_out/r03_ab/probe_*.json; tail -n 2 gpurun_out/r03_ab/*.err | grep -v amdgpu.ids | head -30
