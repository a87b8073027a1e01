#!/bin/bash
# tower machine: kernel trace + PMC passes of the BLS12-381 pairing probe
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r02_tvm_prof; mkdir -p $O; export TMPDIR=/tmp
P="python tools/pair_probe.py bls12381 65536"
timeout 300 rocprofv3 --kernel-trace --stats -d $O -o trace -- $P > $O/trace.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d $O -o sq -- $P > $O/sq.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS -d $O -o lds -- $P > $O/lds.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O -o fetch -- $P > $O/fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O -o write -- $P > $O/write.log 2>&1
for f in $O/*.db; do python tools/rocpd_summary.py $f > ${f%_results.db}.txt 2>&1; rm -f $f; done
grep -h "tvm\|prep" $O/*.txt | cut -c1-200
