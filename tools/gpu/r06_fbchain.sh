#!/bin/bash
# round 6: the fixed-base table's doubling chain on rowfp.cuh (fixed_base.cuh chain_rows_kernel) against the four-lane chain
# (KYB_FB_CHAIN=lanes) -- alternating bases at 2^20 coefficients (every call rebuilds the table) + a trace; then the
# regenerated integer-VALU peak file (tools/valu_peak.hip without the two folded kinds; loop ISA in profiles/r06_valu_peak_isa.txt)
set -u
O=gpurun_out/r06_fbchain; mkdir -p $O; export TMPDIR=/tmp
for i in 1 2; do
  KYB_FB_CHAIN=lanes timeout 300 python tools/fb_probe.py bls12381 1048576 2>/dev/null | tail -1 | sed 's/^{/{"chain": "four lanes", /' >> $O/ab.jsonl
  timeout 300 python tools/fb_probe.py bls12381 1048576 2>/dev/null | tail -1 | sed 's/^{/{"chain": "rows", /' >> $O/ab.jsonl
done
cat $O/ab.jsonl
timeout 300 rocprofv3 --kernel-trace --stats -d $O -o rows_trace -- python tools/fb_probe.py bls12381 1048576 > $O/rows_trace.log 2>&1
KYB_FB_CHAIN=lanes timeout 300 rocprofv3 --kernel-trace --stats -d $O -o lanes_trace -- python tools/fb_probe.py bls12381 1048576 > $O/lanes_trace.log 2>&1
for f in $O/*.db; do python tools/rocpd_summary.py $f > ${f%_results.db}.txt 2>&1; rm -f $f; done
grep -E "chain|table_kernel|member" $O/rows_trace.txt $O/lanes_trace.txt
timeout 120 tools/valu_peak.bin > $O/valu_peak.json 2>$O/valu_peak.err; cat $O/valu_peak.json
