#!/bin/bash
# round 6: everything that touches the staging pools / pool streams, the key-cache counters, the BN G2 table slab and the
# fixed-base chain on rowfp.cuh, then the new-base cost of a same-base batch with the row chain and with the four-lane one
set -u
O=gpurun_out/r06_pools; mkdir -p $O
timeout 1700 python -m pytest tests/test_gpu_concurrency.py tests/test_gpu_verify_same_key.py tests/test_gpu_devices.py "tests/test_gpu_soak.py::test_concurrent_host_threads_and_streams" "tests/test_gpu_soak.py::test_concurrent_round4_entry_points" tests/test_gpu_ed25519.py tests/test_gpu_fixed_base.py tests/test_gpu_callers.py tests/test_gpu_bls12381.py tests/test_gpu_bn256.py -q -s > $O/tests2.log 2>&1; tail -6 $O/tests2.log; grep "variable base" $O/tests2.log | head -2
for i in 1 2; do
  KYB_FB_CHAIN=lanes timeout 300 python tools/fb_probe.py bls12381 65536 2>/dev/null | tail -1 | sed 's/^{/{"chain": "four lanes", /' >> $O/fb_chain_ab.jsonl
  timeout 300 python tools/fb_probe.py bls12381 65536 2>/dev/null | tail -1 | sed 's/^{/{"chain": "rows", /' >> $O/fb_chain_ab.jsonl
done
cat $O/fb_chain_ab.jsonl
export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O -o fb_rows_trace -- python tools/fb_probe.py bls12381 65536 > $O/fb_rows_trace.log 2>&1
for f in $O/*.db; do python tools/rocpd_summary.py $f > ${f%_results.db}.txt 2>&1; rm -f $f; done
grep -E "chain|table_kernel|member" $O/fb_rows_trace.txt
