#!/bin/bash
# MSM: parity tests, then timings of every group at 2^20 and the per-stage trace of the BLS12-381 G1 MSM
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r02_msm; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_callers.py tests/test_gpu_full_size.py tests/test_gpu_devices.py -m gpu -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
timeout 300 python tools/msm_probe.py 1048576 2>/dev/null | tail -1 | tee $O/msm_probe_2p20.json
timeout 300 python tools/msm_probe.py 65536 2>/dev/null | tail -1 | tee $O/msm_probe_2p16.json
timeout 300 rocprofv3 --kernel-trace --stats -d $O -o msm_bls -- python tools/msm_bls_probe.py > $O/msm_bls_probe.json 2> $O/msm_bls.log; cat $O/msm_bls_probe.json
timeout 300 rocprofv3 --kernel-trace --stats -d $O -o msm_ed -- python tools/msm_ed_probe.py > $O/msm_ed_probe.json 2> $O/msm_ed.log; cat $O/msm_ed_probe.json
for f in $O/*.db; do python tools/rocpd_summary.py $f > ${f%_results.db}.txt 2>&1; rm -f $f; done
grep "msm::" $O/msm_bls.txt | cut -c1-120
