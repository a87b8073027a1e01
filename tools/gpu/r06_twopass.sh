#!/bin/bash
# round 6: the MSM's two-pass sort -- parity (the switch probes both ways, the MSM tests, the full-size expectation and
# digest), A/B (KYB_MSM_SORT=single), traces
set -u
O=gpurun_out/r06_twopass; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_switches.py -k "msm" tests/test_gpu_msm.py tests/test_gpu_callers.py "tests/test_gpu_full_size.py::test_msm_at_config_size_against_an_independent_expectation" tests/test_gpu_full_digest.py::test_bls12381_config2_msm_against_the_reference_shaped_sum tests/test_gpu_ed25519.py -x -q > $O/tests.log 2>&1; tail -4 $O/tests.log
tag() { sed "s/^{/{\"run\": \"$1\", /"; }
for i in 1 2 3; do
  KYB_MSM_SORT=single timeout 300 python tools/msm_bls_probe.py 1048576 20 affine | tag single >> $O/ab.jsonl 2>$O/err.log
  timeout 300 python tools/msm_bls_probe.py 1048576 20 affine | tag twopass >> $O/ab.jsonl 2>>$O/err.log
done
for n in 65536 131072 262144 524288 2097152 4194304; do
  KYB_MSM_SORT=single timeout 300 python tools/msm_bls_probe.py $n 20 affine | tag single >> $O/ab.jsonl 2>>$O/err.log
  timeout 300 python tools/msm_bls_probe.py $n 20 affine | tag twopass >> $O/ab.jsonl 2>>$O/err.log
done
KYB_MSM_SORT=single timeout 300 python tools/msm_probe.py 1048576 2>/dev/null | tail -1 | tag single >> $O/ab.jsonl
timeout 300 python tools/msm_probe.py 1048576 2>/dev/null | tail -1 | tag twopass >> $O/ab.jsonl
cat $O/ab.jsonl
timeout 300 rocprofv3 --kernel-trace --stats -d $O -o twopass_trace -- python tools/msm_bls_probe.py 1048576 20 affine > $O/twopass_trace.log 2>&1
for f in $O/*.db; do python tools/rocpd_summary.py $f > ${f%_results.db}.txt 2>&1; rm -f $f; done
head -24 $O/twopass_trace.txt
