#!/bin/bash
# Round 3: fixed-base table at radix 2^10 (26 additions per scalar, 512 entries per window): tests + timings.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r03_fixed_base2; mkdir -p $O; export TMPDIR=/tmp
timeout 150 python -m pytest tests/test_gpu_fixed_base.py tests/test_gpu_callers.py tests/test_gpu_bn256.py tests/test_gpu_devices.py -m gpu -q -x --timeout 60 > $O/pytest_fb.log 2>&1; echo "rc=$?" >> $O/pytest_fb.log; tail -3 $O/pytest_fb.log
for suite in bls12381 bn256; do
  timeout 60 python tools/fb_probe.py $suite 1048576 2>/dev/null | tail -1 | tee -a $O/fb_probe.jsonl
done
timeout 60 rocprofv3 --kernel-trace --stats -d $O -o fb_trace -- python tools/fb_probe.py bls12381 1048576 > $O/fb_trace.log 2>&1
for f in $O/*.db; do python tools/rocpd_summary.py $f > ${f%_results.db}.txt 2>&1; rm -f $f; done
grep -E "fb::" $O/fb_trace.txt | head -6
