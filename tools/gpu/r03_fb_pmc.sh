#!/bin/bash
# Round 3: trace + PMC passes of the fixed-base same-base path (tools/fb_probe.py bls12381 2^20) for roofline_inputs.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r03_fb_pmc; mkdir -p $O; export TMPDIR=/tmp
SQ="SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
timeout 60 rocprofv3 --kernel-trace --stats -d $O -o fb_trace -- python tools/fb_probe.py bls12381 1048576 > $O/fb_trace.log 2>&1
timeout 60 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O -o fb_fetch -- python tools/fb_probe.py bls12381 1048576 > $O/fb_fetch.log 2>&1
timeout 60 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O -o fb_write -- python tools/fb_probe.py bls12381 1048576 > $O/fb_write.log 2>&1
timeout 60 rocprofv3 --kernel-trace --pmc $SQ -d $O -o fb_sq -- python tools/fb_probe.py bls12381 1048576 > $O/fb_sq.log 2>&1
for f in $O/*.db; do python tools/rocpd_summary.py $f > ${f%_results.db}.txt 2>&1; rm -f $f; done
grep -E "fb::mul_kernel<kyb::bls12381_FbG1>" $O/fb_trace.txt $O/fb_sq.txt | head -12 | cut -c1-200
