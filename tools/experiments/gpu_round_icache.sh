#!/bin/bash
# instruction-cache behaviour of the pairing kernel: is the wait share instruction fetch (straight-line code >> 64 KB I-cache)?
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
mkdir -p gpurun_out/icache; export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_VMEM SQ_WAVE_CYCLES -d $R/gpurun_out/icache -o a -- python $R/tools/pair_probe.py ${1:-bls12381} 65536 > $R/gpurun_out/icache/a.log 2>&1
cd $R
f=$(ls gpurun_out/icache/a_*.db 2>/dev/null | head -1)
[ -n "$f" ] && python tools/rocpd_summary.py $f > gpurun_out/icache/${1:-bls12381}.txt 2>&1
grep -E "pair_kernel \|| g1_mul_kernel \|" gpurun_out/icache/${1:-bls12381}.txt | grep -v "kernel | [0-9]"
tail -3 gpurun_out/icache/a.log
