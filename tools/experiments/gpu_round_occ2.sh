#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
mkdir -p gpurun_out/occ; export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -i -B1 -A3 "occupancy" | head -40
cd /tmp
for n in 65536 131072; do
timeout 600 rocprofv3 --kernel-trace --pmc MeanOccupancyPerCU -d $R/gpurun_out/occ -o occb_$n -- python $R/tools/pair_probe.py bls12381 $n > $R/gpurun_out/occ/probeb_$n.log 2>&1
done
cd $R
for n in 65536 131072; do
f=$(ls gpurun_out/occ/occb_${n}*.db 2>/dev/null | head -1)
[ -n "$f" ] && python tools/rocpd_summary.py $f > gpurun_out/occ/occb_$n.txt 2>&1 && rm -f $f
echo "== $n"; grep "Occupancy" gpurun_out/occ/occb_$n.txt | head; tail -2 gpurun_out/occ/probeb_$n.log
done
