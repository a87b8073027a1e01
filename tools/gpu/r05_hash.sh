#!/bin/bash
# Round 5: G1 hash_to_curve with the one-power SSWU (sqrt_ratio): tests, then hash / verify timings
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05_hash; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_bls12381.py tests/test_gpu_verify_same_msg.py tests/test_gpu_verify_same_key.py tests/test_gpu_callers.py -m gpu -q -x > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -2 $O/pytest.log
timeout 300 python tools/verify_probe.py 65536 2>/dev/null | tail -1 | tee $O/verify_probe.json | cut -c1-330
timeout 300 rocprofv3 --kernel-trace --stats -d $O -o t -- python tools/verify_probe.py 65536 > $O/t.log 2>&1
for f in $O/*.db; do python tools/rocpd_summary.py $f > ${f%_results.db}.txt 2>&1; rm -f $f; done
grep -h "operand_kernel\|hash_g1" $O/t.txt | cut -c1-140
