#!/bin/bash
# round 6: the BLS12-381 G1 per-lane ladder's window table in a lane-contiguous global slab (bls12381.cuh g1_mul_glv_lz<TAB>)
# against private scratch (libkyberhip_g1scratch.so = AB_TUS="bls12381 bls12381_g1split" tools/ab_build.sh g1scratch
# -DKYB_BLS_G1_TAB_SCRATCH): parity, then mul_probe at 2^16 (the per-lane band) and 2^15 (split roles), trace + counters
set -u
O=gpurun_out/r06_g1tab; mkdir -p $O; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_bls12381.py tests/test_gpu_switches.py "tests/test_gpu_full_digest.py::test_bls12381_config3_whole_batch_digest" "tests/test_gpu_full_size.py::test_pairing_known_answers_inside_config_size_batches" -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
L=$PWD/kyber_amd/lib/libkyberhip_g1scratch.so
for i in 1 2; do
  for n in 65536 32768; do
    KYBER_HIP_LIB=$L timeout 300 python tools/mul_probe.py bls12381 $n 7 2>/dev/null | tail -1 | sed 's/^{/{"lib": "scratch", /' >> $O/ab.jsonl
    timeout 300 python tools/mul_probe.py bls12381 $n 7 2>/dev/null | tail -1 | sed 's/^{/{"lib": "slab", /' >> $O/ab.jsonl
  done
done
python3 -c "
import json
for l in open('$O/ab.jsonl'):
    d = json.loads(l); print(d['lib'], d['n'], 'g1 checked %.3f trusted %.3f unc %.3f' % (d['g1_checked_ms'], d['g1_trusted_ms'], d['g1_trusted_unc_ms']))"
for v in slab scratch; do
  E=""; [ $v = scratch ] && E="KYBER_HIP_LIB=$L"
  env $E timeout 300 rocprofv3 --kernel-trace --stats -d $O -o ${v}_trace -- python tools/mul_probe.py bls12381 65536 3 > $O/${v}_trace.log 2>&1
  env $E timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O -o ${v}_fetch -- python tools/mul_probe.py bls12381 65536 3 > $O/${v}_fetch.log 2>&1
  env $E timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O -o ${v}_write -- python tools/mul_probe.py bls12381 65536 3 > $O/${v}_write.log 2>&1
done
for f in $O/*.db; do python tools/rocpd_summary.py $f > ${f%_results.db}.txt 2>&1; rm -f $f; done
grep g1_mul_kernel $O/*_trace.txt $O/*_fetch.txt $O/*_write.txt | head -12
