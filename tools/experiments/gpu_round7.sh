#!/bin/bash
set -x
mkdir -p gpurun_out/prof2; export TMPDIR=/tmp
for s in bls12381; do
P="python tools/pair_probe.py $s 65536"
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/prof2 -o ${s}_fetch -- $P > gpurun_out/prof2/${s}_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/prof2 -o ${s}_write -- $P > gpurun_out/prof2/${s}_write.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d gpurun_out/prof2 -o ${s}_sq -- $P > gpurun_out/prof2/${s}_sq.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_IFETCH SQ_INSTS_LDS TCP_TCC_READ_REQ_sum -d gpurun_out/prof2 -o ${s}_mem -- $P > gpurun_out/prof2/${s}_mem.log 2>&1
done
ls gpurun_out/prof2
