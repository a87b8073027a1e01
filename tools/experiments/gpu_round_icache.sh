#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
mkdir -p gpurun_out/icache; export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $R/gpurun_out/icache -o ic -- python $R/tools/pair_probe.py bls12381 65536 > $R/gpurun_out/icache/probe.log 2>&1
cd $R
tail -3 gpurun_out/icache/probe.log
for f in gpurun_out/icache/*.db; do python tools/rocpd_summary.py $f > gpurun_out/icache/ic.txt 2>&1; rm -f $f; done
grep "pair_kernel\|g1_mul_kernel" gpurun_out/icache/ic.txt | head -20
rocprofv3 --list-avail 2>/dev/null | grep -i "icache\|ifetch" | head -20
