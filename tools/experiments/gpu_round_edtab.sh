#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
mkdir -p gpurun_out/edtab; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ed25519.py tests/test_gpu_full_size.py tests/test_gpu_callers.py tests/test_gpu_group_conformance.py -m gpu -x -q 2>&1 | tail -4
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other > gpurun_out/edtab/bench.json 2> gpurun_out/edtab/bench.err
python -c "
import json; d=json.load(open('gpurun_out/edtab/bench.json')); print(d['value'], d['detail'])"
B2="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other"
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
timeout 600 rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/edtab -o ed_$c -- $B2 > $R/gpurun_out/edtab/ed_$c.log 2>&1
done
cd $R
for c in FETCH_SIZE WRITE_SIZE; do
DB=$(find gpurun_out/edtab -name "ed_${c}*.db" | head -1)
python tools/rocpd_summary.py $DB > gpurun_out/edtab/ed_$c.txt 2>&1
grep -i "mul_kernel\|encode\|counter\|#" gpurun_out/edtab/ed_$c.txt | head -12
rm -f $DB
done
