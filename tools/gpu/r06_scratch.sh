#!/bin/bash
# round 6: do the MSM's small kernels pay for their scratch (ROCr's use-once scratch above HSA_SCRATCH_SINGLE_LIMIT)?
# the same trace with the limit raised / async reclaim off
set -u
O=gpurun_out/r06_scratch; mkdir -p $O; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O -o base_trace -- python tools/msm_bls_probe.py 1048576 20 affine > $O/base.log 2>&1
HSA_SCRATCH_SINGLE_LIMIT=8589934592 timeout 300 rocprofv3 --kernel-trace --stats -d $O -o limit_trace -- python tools/msm_bls_probe.py 1048576 20 affine > $O/limit.log 2>&1
HSA_ENABLE_SCRATCH_ASYNC_RECLAIM=0 timeout 300 rocprofv3 --kernel-trace --stats -d $O -o noreclaim_trace -- python tools/msm_bls_probe.py 1048576 20 affine > $O/noreclaim.log 2>&1
HSA_SCRATCH_SINGLE_LIMIT=8589934592 HSA_ENABLE_SCRATCH_ASYNC_RECLAIM=0 timeout 300 rocprofv3 --kernel-trace --stats -d $O -o both_trace -- python tools/msm_bls_probe.py 1048576 20 affine > $O/both.log 2>&1
for f in $O/*.db; do python tools/rocpd_summary.py $f > ${f%_results.db}.txt 2>&1; rm -f $f; done
for v in base limit noreclaim both; do echo "== $v"; tail -1 $O/$v.log; grep -E "bucket_kernel|final_kernel|tree_fold|decode_kernel|reduce_coop|bucket_long" $O/${v}_trace.txt | cut -d'|' -f1,4; done
