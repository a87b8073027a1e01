#!/bin/bash
# instruction-class counters of the pairing kernels: how much of the wait share is vector memory (scratch)?
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
mkdir -p gpurun_out/pairmem; export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_FLAT SQ_WAVE_CYCLES -d $R/gpurun_out/pairmem -o a -- python $R/tools/pair_probe.py bls12381 65536 > $R/gpurun_out/pairmem/a.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR -d $R/gpurun_out/pairmem -o b -- python $R/tools/pair_probe.py bls12381 65536 > $R/gpurun_out/pairmem/b.log 2>&1
cd $R
for t in a b; do f=$(ls gpurun_out/pairmem/${t}_*.db 2>/dev/null | head -1); [ -n "$f" ] && python tools/rocpd_summary.py $f > gpurun_out/pairmem/$t.txt 2>&1 && rm -f $f; grep "bls12381_pair_kernel |" gpurun_out/pairmem/$t.txt | grep -v "^kyb::bls12381_pair_kernel | [0-9]"; done
