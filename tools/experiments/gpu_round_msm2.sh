#!/bin/bash
set -x
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
mkdir -p gpurun_out/msmprof
timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_callers.py tests/test_gpu_full_size.py -m gpu -x -q 2>&1 | tail -5
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/msmprof -o msm -- python $R/tools/msm_bls_probe.py > $R/gpurun_out/msmprof/probe.json 2> $R/gpurun_out/msmprof/probe.err
cd $R
cat gpurun_out/msmprof/probe.json
DB=$(find gpurun_out/msmprof -name "*.db" | head -1)
python tools/rocpd_summary.py $DB > gpurun_out/msmprof/summary.txt 2>&1
head -30 gpurun_out/msmprof/summary.txt
rm -f $DB
timeout 300 python tools/msm_probe.py 1048576
timeout 300 python tools/msm_probe.py 65536
