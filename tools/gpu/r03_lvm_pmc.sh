#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r03_lvm_pmc; mkdir -p $O; export TMPDIR=/tmp
P="python tools/mul_probe.py bls12381 65536 2"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d $O -o sq -- $P > $O/sq.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS -d $O -o lds -- $P > $O/lds.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_IFETCH SQ_INSTS_FLAT -d $O -o br -- $P > $O/br.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O -o fetch -- $P > $O/fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O -o write -- $P > $O/write.log 2>&1
for f in $O/*.db; do python tools/rocpd_summary.py $f > ${f%_results.db}.txt 2>&1; rm -f $f; done
for f in sq lds br fetch write; do echo "== $f"; grep -i "lvm_mul\|^#" $O/$f.txt | head -8; done
