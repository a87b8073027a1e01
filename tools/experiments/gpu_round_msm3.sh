#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
mkdir -p gpurun_out/msmprof_ed
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/msmprof_ed -o msm -- python $R/tools/msm_ed_probe.py > $R/gpurun_out/msmprof_ed/probe.json 2> $R/gpurun_out/msmprof_ed/probe.err
cd $R
cat gpurun_out/msmprof_ed/probe.json
DB=$(find gpurun_out/msmprof_ed -name "*.db" | head -1)
python tools/rocpd_summary.py $DB > gpurun_out/msmprof_ed/summary.txt 2>&1
head -24 gpurun_out/msmprof_ed/summary.txt
rm -f $DB
