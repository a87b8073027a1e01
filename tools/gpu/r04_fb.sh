#!/bin/bash
# Round 4: the fixed-base path with endomorphism policies, table-based membership, cooperative chain only.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04_fb; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_fixed_base.py tests/test_gpu_switches.py tests/test_gpu_callers.py tests/test_gpu_lane_vm.py -m gpu -q -x > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
for s in bls12381 bn256 bn254; do timeout 300 python tools/fb_probe.py $s 1048576 2>/dev/null | tail -1 | tee -a $O/fb_probe.jsonl; done
timeout 200 rocprofv3 --kernel-trace --stats -d $O -o fb_trace -- python tools/fb_probe.py bls12381 1048576 > $O/fb_trace.log 2>&1
for f in $O/*.db; do python tools/rocpd_summary.py $f > ${f%_results.db}.txt 2>&1; rm -f $f; done
head -14 $O/fb_trace.txt | cut -c1-170
