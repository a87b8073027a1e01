#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
mkdir -p gpurun_out/occ; export TMPDIR=/tmp
cd /tmp
for n in 65536 131072; do
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_LEVEL_WAVES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU -d $R/gpurun_out/occ -o occ_$n -- python $R/tools/pair_probe.py bls12381 $n > $R/gpurun_out/occ/probe_$n.log 2>&1
done
cd $R
for n in 65536 131072; do
f=$(ls gpurun_out/occ/occ_${n}*.db | head -1)
python tools/rocpd_summary.py $f > gpurun_out/occ/occ_$n.txt 2>&1; rm -f $f
echo "== $n"; grep "bls12381_pair_kernel\|bls12381_g1_mul_kernel" gpurun_out/occ/occ_$n.txt
done
