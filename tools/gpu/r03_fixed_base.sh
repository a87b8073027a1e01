#!/bin/bash
# Round 3: fixed-base table for same-base batches (fixed_base.cuh): the whole GPU suite (Commit-shaped calls now take
# the table), then timings with and without it (KYB_FB_MIN=0), same box.  Every step under its own timeout.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r03_fixed_base; mkdir -p $O; export TMPDIR=/tmp
timeout 200 python -m pytest tests/test_gpu_fixed_base.py -m gpu -q -x --timeout 60 > $O/pytest_fb.log 2>&1; echo "rc=$?" >> $O/pytest_fb.log; tail -4 $O/pytest_fb.log
timeout 200 python -m pytest tests -m gpu -q --timeout 60 --deselect tests/test_gpu_fixed_base.py > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
for suite in bls12381 bn256; do
  for fbm in 0 default; do
    if [ $fbm = 0 ]; then export KYB_FB_MIN=0; else unset KYB_FB_MIN; fi
    timeout 60 python tools/fb_probe.py $suite 1048576 2>/dev/null | tail -1 | tee -a $O/fb_probe.jsonl
  done
done
unset KYB_FB_MIN
timeout 60 rocprofv3 --kernel-trace --stats -d $O -o fb_trace -- python tools/fb_probe.py bls12381 1048576 > $O/fb_trace.log 2>&1
for f in $O/*.db; do python tools/rocpd_summary.py $f > ${f%_results.db}.txt 2>&1; rm -f $f; done
grep -E "fb::" $O/fb_trace.txt | head -8
