#!/bin/bash
# round 6: the split tail for every adapter with cooperative slots (BLS12-381 G2, bn256 / bn254 G1 and G2: chains by
# final_chains_kernel) -- parity, A/B by switch (KYB_MSM_REDUCE=mul), traces
set -u
O=gpurun_out/r06_split_all; mkdir -p $O; export TMPDIR=/tmp
timeout 2400 python -m pytest tests/test_gpu_switches.py -k msm tests/test_gpu_msm.py tests/test_gpu_callers.py tests/test_gpu_bn256.py tests/test_gpu_bn254.py tests/test_gpu_bls12381.py "tests/test_gpu_full_size.py::test_msm_at_config_size_against_an_independent_expectation" tests/test_gpu_soak.py -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
tag() { sed "s/^{/{\"run\": \"$1\", /"; }
for i in 1 2; do
  KYB_MSM_REDUCE=mul timeout 600 python tools/msm_probe.py 1048576 2>/dev/null | tail -1 | tag reduce_mul >> $O/ab.jsonl
  timeout 600 python tools/msm_probe.py 1048576 2>/dev/null | tail -1 | tag split >> $O/ab.jsonl
done
KYB_MSM_REDUCE=mul timeout 600 python tools/msm_probe.py 65536 2>/dev/null | tail -1 | tag reduce_mul >> $O/ab.jsonl
timeout 600 python tools/msm_probe.py 65536 2>/dev/null | tail -1 | tag split >> $O/ab.jsonl
cat $O/ab.jsonl
timeout 600 rocprofv3 --kernel-trace --stats -d $O -o split_trace -- python tools/msm_probe.py 1048576 > $O/split_trace.log 2>&1
for f in $O/*.db; do python tools/rocpd_summary.py $f > ${f%_results.db}.txt 2>&1; rm -f $f; done
grep -E "reduce_coop|final_chains|final_kernel|tree_fold" $O/split_trace.txt | cut -c1-100,140-180
