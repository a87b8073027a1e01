#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r03_micro; mkdir -p $O; export TMPDIR=/tmp
for k in "ctradd(no slot traffic)" "fp_raw(one slot)" "fp_mul(slot,slot)" "fp_2mul" "pair_mul2" "pair_sqr2"; do
  n=$(echo "$k" | tr -c 'a-z0-9_' '_')
  LVM_MICRO_ONLY="$k" timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU -d $O -o $n -- python tools/lvm_microbench.py 2 > $O/$n.log 2>&1
done
for f in $O/*.db; do python tools/rocpd_summary.py $f > ${f%_results.db}.txt 2>&1; rm -f $f; done
for f in $O/*.txt; do echo "== $f"; grep "lvm_mul_kernel" $f | grep -v "calls" | awk -F'|' '{printf "%s %s\n", $2, $4}'; done
